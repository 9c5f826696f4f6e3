// rg_convp.hip -- transposed 4x4 stride-2 conv with 64 output channels and 128 input channels (the generator's last MFMA
// layer 128 -> 64 at 128 x 128 and the discriminator's data gradient of layer 1) with the INPUT PATCH RESIDENT IN LDS.
//
// The implicit-GEMM kernels (rg_mfma.hip / rg_conv8.hip) pull every input pixel of this layer through LDS-DMA 16 times
// (4 output-parity classes x 2x2 taps): with only 64 output columns the matrix cores need 64 B/clk/CU of A operand at
// peak and the layer runs at the L2 -> LDS rate instead (138 us, 0.21 MFMA-busy at batch 64).  The 16 (class, tap)
// pairs read 9 shifted copies of the same pixels, so here a workgroup loads its pixels ONCE:
//
//  * tile = 256 consecutive low-resolution pixels (256 / Ws whole image rows) x all 4 classes x 64 columns.  The patch
//    is those rows plus one halo row above and below, [rows + 2][pad, x = 0 .. Ws-1] pixels of 128 B (64 channels: the
//    K = 128 channels are walked as two halves X, Y held in two buffers); the pad pixel between rows and the rows outside
//    the image are zero-filled by the DMA's range check, so a tap shift (dh, dw) is a plain row offset dh*(Ws+1) + dw.
//  * 8 waves = 4 classes x 2 halves of the tile: a wave owns 128 pixels x 64 columns of ONE class (32 accumulators of
//    v_mfma_f32_16x16x32_bf16) and walks its class's k = (channel half, tap, 32-channel step): 16 steps of 32 MFMAs.
//    A fragments come from the patch at the tap's shift (XOR swizzle rho & 6 of the 16-byte segment: conflict-free for the
//    ds_read_b128 lane groups at ANY row shift, found by exhaustive search over the linear swizzles; the row tiles of one image
//    row share one address register + immediates), B fragments from a 3-slot ring of 16 KB steps ([class][64 columns]
//    [32 channels], each wave DMAs its own class's rows).
//  * persistent: a workgroup walks consecutive tiles; the next tile's X half is DMA'd during steps 8-11 (X is free once
//    step 7's reads are done), the Y half during steps 0-3, the B ring runs 3 steps ahead; one s_barrier per step, counted
//    vmcnt (never 0 inside the loop).  Fragment reads are software-pipelined across the barrier (A sub-block 0 / B of
//    step s+1 are read under the MFMAs of step s's sub-block 1).  The 16 steps are fully unrolled (tap, channel half and
//    DMA schedule are compile-time), the image width is a template parameter.
//  * the MFMAs compute the transposed tile (weights as the A operand), so a lane's four accumulator values are four
//    consecutive channels of one pixel: the epilogue stores 8-byte pieces straight from registers (no LDS transposition:
//    the fp32 staging of a 1024 x 64 tile cost 7 us per tile, a third of the kernel), with the fused LeakyReLU-backward
//    mask (PACKED sign bits, one 64-bit word per output pixel, DMA'd into LDS during step 8) / folded BatchNorm affine /
//    BatchNorm partial sums (DPP row reductions) of the other kernels.
// LDS-DMA traffic per launch at batch 64: 100 MB of input + 268 MB of (L2-resident) weights instead of 1.07 GB + 268 MB.
#include "rg_gather.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int V> using icp = std::integral_constant<int, V>;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int CP_PXMAX = 392;                    // patch pixels per buffer, rounded up to whole 8-pixel DMA instructions
constexpr int CP_PBUF = CP_PXMAX * 128;          // 50176 B
constexpr int CP_OFF_X = 0, CP_OFF_Y = CP_PBUF, CP_OFF_RING = 2 * CP_PBUF;
constexpr int CP_SLOT = 16384;
constexpr int CP_OFF_AFF = CP_OFF_RING + 3 * CP_SLOT;  // [scale 64][shift 64] fp32 of the folded BatchNorm affine
constexpr int CP_OFF_BITS = CP_OFF_AFF + 512;          // 8 x 1 KB packed LeakyReLU mask bits of the waves' pixels
constexpr int CP_OFF_DUMMY = CP_OFF_BITS + 8192;       // 1 KB target of the dead (all-out-of-range) DMAs
constexpr int CP_LDS = CP_OFF_DUMMY + 1024;            // 159232 B

// EPI: 0 plain, 1 BatchNorm partial sums, 2 packed LeakyReLU mask, 3 folded BatchNorm affine + LeakyReLU (compile-time: the
// epilogue walks 32 accumulator tiles, run-time flags would be tested in every one of them)
template <int LGW, int EPI>
__global__ __launch_bounds__(512, 2) void convp_kernel(G2Args a2) {
  constexpr bool HAS_STATS = EPI == 1, HAS_MASK = EPI == 2, HAS_AFFINE = EPI == 3;
  constexpr int Ws = 1 << LGW, Wp = Ws + 1;
  constexpr int R = 256 >> LGW;                            // image rows per tile
  constexpr int PATCH_PX = (R + 2) * Wp + 1;
  constexpr int NPI = (PATCH_PX + 7) / 8;                  // patch DMA instructions per buffer (<= 49)
  constexpr int PPW = (NPI + 7) / 8;                       // ... per wave (wave w issues ids w*PPW .. w*PPW + PPW-1 < NPI)
  static_assert(PATCH_PX <= CP_PXMAX && PPW <= 8, "patch does not fit");
  __shared__ __attribute__((aligned(16))) uint4 lds[CP_LDS / 16];
  const GArgs& g = a2.g;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int cls = wave >> 1, half = wave & 1;
  const int ph = cls >> 1, pw = cls & 1;
  const int Hs = g.Hs, lgH = g.lgH;
  const int tiles = g.M >> 8;
  const int per = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T0 = (int)blockIdx.x * per, T1 = min(tiles, T0 + per);
  if (T0 >= T1) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a2.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, a2.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  char* const ldsb = reinterpret_cast<char*>(lds);
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;
  // The lane index through an opaque copy: what is derived from it is recomputed at the use site (a few VALU operations)
  // instead of being hoisted out of the tile loop as loop-invariant per-lane values -- with 128 accumulator and 64 fragment
  // registers live there is no room for those, and a spilled one costs an s_waitcnt vmcnt(0) (scratch reload) that drains the
  // DMA queue.  Used for everything outside the 16 steps' own few persistent values.
  auto opq = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  if (HAS_AFFINE && t < 128)                                // (before any DMA is in flight; published by the prologue's barrier)
    reinterpret_cast<float*>(ldsb + CP_OFF_AFF)[t] = t < 64 ? g.scale[t] : g.shift[t - 64];

  // ---- tap tables of this wave's class (wave-uniform scalars): patch row shift and B tap offset (bytes) of tap (a, b)
  int dlt[4], bof[4];
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) {
    int kh, kw, dh, dw;
    up_tap_dev(ph, tp >> 1, kh, dh);
    up_tap_dev(pw, tp & 1, kw, dw);
    dlt[tp] = dh * Wp + dw;
    bof[tp] = (kh * 4 + kw) * g.b_tap * 2;
  }

  // ---- B ring DMA: step = (c2, tap, kc) -> [64 columns][32 channels] of this wave's class; the wave's two instructions
  // cover columns half*32 + e*16 + (lane >> 2), 16-byte segment (lane & 3) holding logical segment seg ^ (((col >> 4) & 1) << 1)
  int b_lane[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int col = half * 32 + e * 16 + (lane >> 2);
    const int lseg = (lane & 3) ^ (((col >> 4) & 1) << 1);
    b_lane[e] = (col * g.b_col + lseg * 8) * 2;
  }
  const int ring_w = CP_OFF_RING + cls * 4096 + half * 2048;       // + slot * CP_SLOT + e * 1024
  auto issue_b = [&](auto STEP, int slot) {               // step 0..15 (the B stream repeats per tile)
    constexpr int step = decltype(STEP)::value & 15;
    constexpr int c2 = (step >> 3) & 1, tp = (step >> 1) & 3, kc = step & 1;
    const int bo = bof[tp] + (c2 * 64 + kc * 32) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int bl = b_lane[e];
      asm volatile("" : "+v"(bl));                        // (b_lane + bo is tile-invariant for each of the 16 steps: see opq)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr_t)(ldsb + ring_w + slot * CP_SLOT + e * 1024), 16,
                                               (unsigned)(bl + bo), 0, 0, 0);
    }
  };
  // ---- patch DMA: instruction id (0 .. NPI-1) covers LDS pixel rows 8 id + (lane >> 3), physical segment lane & 7 holding
  // logical segment seg ^ (row & 6).  tbase = byte offset of pixel (n, y0 - 1, 0) of the tile (wave-uniform, may be
  // negative), y0 = first image row of the tile.
  // `live` false: an all-out-of-range DMA into the dummy KB (every wave issues the same number of DMAs per step, so the
  // counted waits are compile-time constants)
  auto issue_patch = [&](int tbase, int y0, int c2, int id, bool live) {
    const int ln = opq();
    const int rho = id * 8 + (ln >> 3);
    const int yy = rho / Wp, c = rho - yy * Wp;
    const int lseg = (ln & 7) ^ (rho & 6);
    const bool v = live && c >= 1 && (unsigned)(y0 - 1 + yy) < (unsigned)Hs && rho < PATCH_PX;
    const int off = tbase + ((yy * Ws + (c - 1)) * 128 + c2 * 64 + lseg * 8) * 2;
    const int dst = live ? (c2 ? CP_OFF_Y : CP_OFF_X) + id * 1024 : CP_OFF_DUMMY;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr_t)(ldsb + dst), 16, v ? (unsigned)off : OOB, 0, 0, 0);
  };
  auto tile_base = [&](int T, int& tbase, int& y0) {
    const int m0 = T << 8;
    const int n = m0 >> (LGW + lgH);
    y0 = (m0 >> LGW) & (Hs - 1);
    tbase = ((n * Hs + y0 - 1) * Ws) * 256;
  };

  // ---- fragments (16x16x32: lane = (row fr of 16, k-quarter fh))
  // A: pixel p = half*128 + rt*16 + fr of the tile -> patch row rho = p + (p >> LGW) + Ws + 2 + delta(tap) = fr + U(rt, tap)
  // with U wave-uniform (fr < 16 <= Ws never carries into the image row).  Row tiles of one image row are 16 patch rows =
  // 2048 B apart with the same swizzle (rho & 6 ignores multiples of 16): one address + immediates per image row.
  const int fr = lane & 15, fh = lane >> 4;
  // B (= MFMA A operand, see the epilogue): row i of column tile j is output channel 32 (j >> 1) + 8 (i >> 2) + 4 (j & 1) + (i & 3),
  // so that tiles 0, 1 give a lane (px, fq = i >> 2) the 8 consecutive channels 8 fq .. 8 fq + 7 and tiles 2, 3 the channels
  // 32 + 8 fq .. + 7: each of the lane's two 16-byte stores, taken over the four lanes of a pixel, covers one CONTIGUOUS
  // 64-byte half of the pixel's 128-byte line.  (Until round 5 a lane held 16 consecutive channels and each store instruction
  // wrote 16-byte pieces with 16-byte holes: half-written 32-byte sectors that the L2 evicted and wrote again -- the PMC
  // counted 168-209 MB written for 134 MB of output.)  The ring's swizzle is a function of the COLUMN, (col >> 4) & 1, as
  // before: with col = 8 (fr >> 2) + (fr & 3) (+ 0, 4, 32, 36 per tile) that is bit 3 of fr, and the four segment values of
  // every ds_read_b128 lane group stay distinct.
  const unsigned b_rd = lds_base + CP_OFF_RING + cls * 4096 +
                        (unsigned)((8 * (fr >> 2) + (fr & 3)) * 64 + ((fh ^ (((fr >> 3) & 1) << 1)) << 4));
  f32x4_t acc[8][4];
  u32x4_t aS0[4], aS1[4], bS[2][4];                       // B(s) lives in set s & 1

#define CP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
  // A fragments of row sub-tiles RT0 .. RT0+3 (RT0 = 0 or 4) at patch shift `delta`, buffer LDS address `base`, k-step KC
  auto read_a = [&](u32x4_t(&dst)[4], auto RT0, int delta, unsigned base, auto KC) {
    constexpr int rt0 = decltype(RT0)::value, kc = decltype(KC)::value;
    constexpr int TPR = Ws / 16 < 4 ? Ws / 16 : 4;         // row tiles of this call that share an image row
    const int ln = opq();                                  // (every address below is tile-invariant: see opq)
    const int kq = kc * 4 + (ln >> 4);                     // logical 16-byte segment of this k-step
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += TPR) {
      const int pu = half * 128 + (rt0 + i0) * 16;         // wave-uniform
      const int U = pu + (pu >> LGW) + Ws + 2 + delta;
      const int rho = (ln & 15) + U;
      const unsigned addr = base + (unsigned)((rho << 7) + ((kq ^ (rho & 6)) << 4));
#pragma unroll
      for (int i = 0; i < TPR; ++i) {
        if (i == 0) CP_DSR(dst[i0 + 0], addr, 0);
        else if (i == 1) CP_DSR(dst[i0 + 1], addr, 2048);
        else if (i == 2) CP_DSR(dst[i0 + 2], addr, 4096);
        else CP_DSR(dst[i0 + 3], addr, 6144);
      }
    }
  };
#define CP_READ_B(SET, SLOTOFF)                                                                   \
  do {                                                                                            \
    unsigned ba_ = b_rd;                                                                          \
    asm volatile("" : "+v"(ba_));                                                                 \
    ba_ += (SLOTOFF);                                                                             \
    CP_DSR(bS[SET][0], ba_, 0); CP_DSR(bS[SET][1], ba_, 256);                                     \
    CP_DSR(bS[SET][2], ba_, 2048); CP_DSR(bS[SET][3], ba_, 2304);                                 \
  } while (0)
#define CP_WAIT4(X) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3])::"memory")
// MFMAs as inline asm with the accumulator tied in place ("+v"): with the builtin hipcc lets D and C differ and walks the
// accumulators through the register file from step to step, which at 250 live registers ends in spills.  Hazards the
// compiler no longer sees: the epilogue's first VALU read of an accumulator (explicit s_nop before it); A / B fragments
// are produced by ds_reads behind explicit lgkmcnt waits.
#define CP_MFMAS(RT0, AS, SET)                                                                    \
  do {                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)   \
        asm volatile(RG_MFMA_H16_ASM_16x16x32 " %0, %1, %2, %0" : "+v"(acc[(RT0) + i_][j_]) : "v"(bS[SET][j_]), "v"(AS[i_])); \
    __builtin_amdgcn_s_setprio(0);                                                                \
  } while (0)
// wait states tied to every accumulator: hazards between the asm MFMAs and the VALU instructions around them (the clearing
// v_movs before a tile's first MFMAs, the epilogue's reads after its last ones) are invisible to the compiler
// An MFMA re-reads its A / B registers in every pass: a VALU write to one of them within the MFMA's 16 cycles corrupts the rows
// of the later passes (seen while moving the DMA issue between the MFMAs: channels 4 j + 2, 4 j + 3 of a tile wrong).  hipcc
// cannot know that about the asm MFMAs and may reuse a dead fragment register as a temporary of the address arithmetic that
// follows a group: keep the group's operands allocated (and two more wait states) behind the waits that end it.
#define CP_KEEP_ALL(AS, SET)                                                                                  \
  asm volatile("s_nop 1" ::"v"(bS[SET][0]), "v"(bS[SET][1]), "v"(bS[SET][2]), "v"(bS[SET][3]), "v"(AS[0]), "v"(AS[1]), \
               "v"(AS[2]), "v"(AS[3]))
#define CP_ACC_FENCE()                                                                                       \
  do {                                                                                                       \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 8; rt_ += 4)                                                   \
      asm volatile("s_nop 15\n\ts_nop 15"                                                                    \
                   : "+v"(acc[rt_][0]), "+v"(acc[rt_][1]), "+v"(acc[rt_][2]), "+v"(acc[rt_][3]), "+v"(acc[rt_ + 1][0]),      \
                     "+v"(acc[rt_ + 1][1]), "+v"(acc[rt_ + 1][2]), "+v"(acc[rt_ + 1][3]), "+v"(acc[rt_ + 2][0]),              \
                     "+v"(acc[rt_ + 2][1]), "+v"(acc[rt_ + 2][2]), "+v"(acc[rt_ + 2][3]), "+v"(acc[rt_ + 3][0]),              \
                     "+v"(acc[rt_ + 3][1]), "+v"(acc[rt_ + 3][2]), "+v"(acc[rt_ + 3][3]));                                    \
  } while (0)
#define CP_STORE_TAIL "\n\ts_nop 2"
#define CP_SYNC() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- epilogue.  The MFMAs compute the TRANSPOSED tile (weights as the A operand, pixels as the B operand): accumulator
  // lane (px = lane & 15, fq = lane >> 4) of tile (rt, j) holds channels CH(fq, j) .. + 3 = 32 (j >> 1) + 8 fq + 4 (j & 1) .. + 3
  // of pixel rt*16 + px -- tiles 0, 1 are 8 consecutive channels = one 16-byte store, tiles 2, 3 the second, stored straight
  // from registers (no LDS transposition); per store instruction the four lanes of a pixel write 64 contiguous bytes.
  constexpr int Wq2 = 2 * Ws;
  // byte offset of output pixel (class (ph, pw)) of low-resolution pixel m (wave-uniform m), 128 B per output pixel
  auto out_base = [&](int mu) {
    const int wq = mu & (Ws - 1), hq = (mu >> LGW) & (Hs - 1), n = mu >> (LGW + lgH);
    return (((long long)n * (2 * Hs) + 2 * hq + ph) * Wq2 + 2 * wq + pw) * 128;
  };
  // Fused LeakyReLU backward: g.mask holds PACKED sign bits of the consumer's activation, one 64-bit word per output pixel
  // (bit c set: activation of channel c > 0; rg_sign_pack / first_down write them).  A wave's 128 pixels are 1 KB of bits,
  // DMA'd into its LDS slice [128 pixels][8 B] during step 8 (4 instructions of 64 x 4 B, counted with that step's patch DMAs).
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)g.mask, 0, 0x7fffff00, 0x00020000);
  auto load_maskbits = [&](int T) {
    const int m_w = (T << 8) + half * 128;                // (a multiple of 128 >= Ws: the wave starts at x = 0 of an image row)
    const unsigned sbase = (unsigned)(out_base(m_w) >> 4);                // 8 B per output pixel: wave-uniform, in SOFFSET
    const int ln = opq();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = i * 32 + (ln >> 1);                   // pixel of the wave; lane parity = dword of its 64-bit word
      const int d = (2 * (p >> LGW) * Wq2 + 2 * (p & (Ws - 1))) * 8 + (ln & 1) * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (lds_vptr_t)(ldsb + CP_OFF_BITS + wave * 1024 + i * 256), 4, (unsigned)d,
                                               sbase, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto dpp_sum16 = [&](float x) {                         // sum over the 16 lanes of a DPP row (every lane gets it)
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));  // row_mirror
    return x;
  };

  auto epilogue = [&](int T) {
    CP_ACC_FENCE();                                       // MFMA write -> VALU read
    // ... and the operands of the tile's LAST MFMA group (CP_LAST: sub-block 1 with aS1 / bS[1]) stay allocated across that
    // fence: the group has no wait behind it, and hipcc hoists the epilogue's lane arithmetic above the fence into registers it
    // considers dead -- a VALU write into a fragment register while its MFMA still re-reads it corrupts rows 2, 3 of the later
    // passes (round 5: after a change that moved the allocation, channels 32 + 8 fq + 2, 3 of the last row tiles were garbage
    // in the mask variant; the tests caught it)
    CP_KEEP_ALL(aS1, 1);
    const int m_w = (T << 8) + half * 128;                // first pixel of this wave
    const int ln = opq();
    const int px = ln & 15, fq = ln >> 4;                 // the lane's pixel of a row tile; its channels are 32 (j >> 1) + 8 fq + 4 (j & 1) + r
    const unsigned e_off = (unsigned)(px * 256 + fq * 16);
    const unsigned bits_r = lds_base + CP_OFF_BITS + wave * 1024 + (unsigned)(px * 8);
    u32x4_t asc[4], ash[4];                               // scale / shift of channels CH(fq, j) .. + 3 (LDS table, asm reads)
    if constexpr (HAS_AFFINE) {
      const unsigned aff_r = lds_base + CP_OFF_AFF + (unsigned)(fq * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(asc[j]) : "v"(aff_r), "n"((j >> 1) * 128 + (j & 1) * 16) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ash[j]) : "v"(aff_r), "n"(256 + (j >> 1) * 128 + (j & 1) * 16) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(asc[0]), "+v"(asc[1]), "+v"(asc[2]), "+v"(asc[3]), "+v"(ash[0]), "+v"(ash[1]),
                   "+v"(ash[2]), "+v"(ash[3])::"memory");
    }
    f32x2_t s1[4][2], s2[4][2];                           // BatchNorm partial sums of the bf16-rounded values (packed fp32 math)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) { s1[j][h] = f32x2_t{0.f, 0.f}; s2[j][h] = f32x2_t{0.f, 0.f}; }
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
      unsigned b8lo = 0, b8hi = 0;                        // sign bits of channels 8 fq .. + 7 and 32 + 8 fq .. + 7
      if constexpr (HAS_MASK) {
        u32x2_t w;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w) : "v"(bits_r), "n"(rt * 128) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w)::"memory");
        b8lo = w.x >> (fq * 8);
        b8hi = w.y >> (fq * 8);
        // both words are formed HERE, right behind the wait (an opaque use pins the two shifts to this point).  Left to the
        // compiler, the second shift was sunk 60 instructions down to its first use and lanes 12..15 of every row tile read
        // garbage for channels 32 + 8 fq + 2, 3 (found by the bit-exact test against the implicit-GEMM kernel; a build with
        // more wait states in front of the shift failed on MORE elements, one with the shifts pinned passed.  The cause was NOT
        // isolated: the pin is what the measurements support, nothing more)
        asm volatile("" : "+v"(b8lo), "+v"(b8hi));
      }
      u32x4_t o[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t v = acc[rt][j];
        if constexpr (HAS_MASK) {
          const float sl = g.mslope;
          const unsigned b4 = ((j >> 1) ? b8hi : b8lo) >> (4 * (j & 1));
          v[0] *= (b4 & 1u) ? 1.f : sl; v[1] *= (b4 & 2u) ? 1.f : sl;
          v[2] *= (b4 & 4u) ? 1.f : sl; v[3] *= (b4 & 8u) ? 1.f : sl;
        }
        if constexpr (HAS_AFFINE) {
          const float4 sc = __builtin_bit_cast(float4, asc[j]), sh = __builtin_bit_cast(float4, ash[j]);
          v[0] = lrelu_f(v[0] * sc.x + sh.x, g.slope); v[1] = lrelu_f(v[1] * sc.y + sh.y, g.slope);
          v[2] = lrelu_f(v[2] * sc.z + sh.z, g.slope); v[3] = lrelu_f(v[3] * sc.w + sh.w, g.slope);
        }
        const uint32_t h0 = f32_to_h16(v[0]), h1 = f32_to_h16(v[1]), h2 = f32_to_h16(v[2]), h3 = f32_to_h16(v[3]);
        const uint32_t d0 = h0 | (h1 << 16), d1 = h2 | (h3 << 16);
        o[j >> 1][(j & 1) * 2] = d0;
        o[j >> 1][(j & 1) * 2 + 1] = d1;
        if constexpr (HAS_STATS) {
          const f32x2_t r01 = {h16lo_to_f32(d0), h16hi_to_f32(d0)};
          const f32x2_t r23 = {h16lo_to_f32(d1), h16hi_to_f32(d1)};
          s1[j][0] += r01; s2[j][0] += r01 * r01;
          s1[j][1] += r23; s2[j][1] += r23 * r23;
        }
      }
      char* cb = reinterpret_cast<char*>(g.C) + out_base(m_w + rt * 16);
      // (CP_STORE_TAIL: a VALU write into the data registers of a store of more than 8 bytes needs wait states behind the store (two on gfx940 and later; three are given) --
      // hipcc provides it for its own stores, not for these: the fp16 build's statistics variant reused o[0] in the very next
      // instruction and stored garbage in the first channel pair of every pixel)
      asm volatile("global_store_dwordx4 %0, %1, %2" CP_STORE_TAIL ::"v"(e_off), "v"(o[0]), "s"(cb) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, %2 offset:64" CP_STORE_TAIL ::"v"(e_off), "v"(o[1]), "s"(cb) : "memory");
    }
    if constexpr (HAS_STATS) {
      const size_t grow = ((size_t)cls * tiles + T) * 2 + half;
      const char* sb = reinterpret_cast<const char*>(g.stats + grow * 128);       // wave-uniform row [2][64]
      const unsigned so = (unsigned)(fq * 32);             // floats CH(fq, j) .. + 3 of the [2][64] row
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t q1 = {dpp_sum16(s1[j][0][0]), dpp_sum16(s1[j][0][1]), dpp_sum16(s1[j][1][0]), dpp_sum16(s1[j][1][1])};
        const f32x4_t q2 = {dpp_sum16(s2[j][0][0]), dpp_sum16(s2[j][0][1]), dpp_sum16(s2[j][1][0]), dpp_sum16(s2[j][1][1])};
        if (px == 0) {
          asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" CP_STORE_TAIL ::"v"(so), "v"(q1), "s"(sb), "n"((j >> 1) * 128 + (j & 1) * 16) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" CP_STORE_TAIL ::"v"(so), "v"(q2), "s"(sb), "n"(256 + (j >> 1) * 128 + (j & 1) * 16) : "memory");
        }
      }
    }
    // (clearing here, not accumulating the next tile's first MFMAs onto a zero constant: with that form hipcc's register
    // allocator renames accumulators across the loop and spills)
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][j][r] = 0.f;
    CP_ACC_FENCE();                                       // VALU write -> MFMA read
  };

  // ---- prologue: X half of the first tile, B steps 0-2, then the fragments of step 0 and its sub-block 0
  int tb_cur, y0_cur, tb_nxt = 0, y0_nxt = 0;
  tile_base(T0, tb_cur, y0_cur);
#pragma unroll
  for (int j = 0; j < PPW; ++j)
    if (wave * PPW + j < NPI) issue_patch(tb_cur, y0_cur, 0, wave * PPW + j, true);
  issue_b(icp<0>{}, 0);
  issue_b(icp<1>{}, 1);
  issue_b(icp<2>{}, 2);
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rt][j][r] = 0.f;
  CP_ACC_FENCE();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CP_SYNC();
  {
    CP_READ_B(0, 0);
    read_a(aS0, icp<0>{}, dlt[0], lds_base + CP_OFF_X, icp<0>{});
    read_a(aS1, icp<4>{}, dlt[0], lds_base + CP_OFF_X, icp<0>{});
    CP_WAIT4(bS[0]);
    CP_WAIT4(aS0);
    CP_WAIT4(aS1);
    CP_MFMAS(0, aS0, 0);
  }
  int rd_slot = 1, wr_slot = 0;                           // ring slot of step s+1 / of step s+3

  // One interval = barrier(s) .. barrier(s+1) of step S (compile-time, KC = S & 1):
  //   DMA issue (B step s+3, then the patch instructions on schedule) | reads B(s+1), A sub-block 0 (s+1) | MFMA sub-block 1 (s)
  //   | reads A sub-block 1 (s+1) | MFMA sub-block 0 (s+1) | counted waits.
  // Counted wait at the end of interval s: B(s+2) (issued in interval s-1) has to be there at barrier(s+1); still allowed in
  // flight are the patch loads issued behind it in interval s-1 and everything issued in interval s, so a patch load has
  // two intervals to arrive.  The last interval of a tile does the MFMAs of sub-block 1 first, then the epilogue with
  // nothing but the accumulators live, then the reads of the next tile's step 0.
  // Patch schedule: wave w issues its instructions j = 0 .. PPW-1 two per interval, Y of this tile in steps 0-3, X of the
  // next tile in steps 8-11.
  auto issue_step = [&](auto S, int T) {
    constexpr int s = decltype(S)::value;
    issue_b(icp<s + 3>{}, wr_slot);
    wr_slot = wr_slot == 2 ? 0 : wr_slot + 1;
    if constexpr ((s & 4) == 0) {                          // steps 0-3 and 8-11: two patch DMAs per wave (dead ones included)
      constexpr bool nextx = (s & 8) != 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        constexpr int jbase = (s & 3) * 2;
        const int j = jbase + e;
        const int id = wave * PPW + j;
        const bool live = j < PPW && id < NPI && (!nextx || T + 1 < T1);
        issue_patch(nextx ? tb_nxt : tb_cur, nextx ? y0_nxt : y0_cur, nextx ? 0 : 1, live ? id : 0, live);
        __builtin_amdgcn_sched_barrier(0);                 // (one address computation at a time: register pressure)
      }
      if constexpr (s == 8)
        if constexpr (HAS_MASK) load_maskbits(T);
    }
  };
  // Counted wait at the end of interval s.  Operations of interval s in issue order: B (2), patch (2 in steps 0-3 / 8-11),
  // mask bits (4 in step 8 with a mask), epilogue stores (16, + 8 with statistics, in step 15).  Allowed in flight: what
  // interval s-1 issued behind its B, plus all of interval s.
  auto wait_v = [&](auto S) {
    constexpr int s = decltype(S)::value, sp = (s + 15) & 15;
    constexpr int np_s = (s & 4) == 0 ? 2 : 0, np_p = (sp & 4) == 0 ? 2 : 0;
    constexpr int base = np_p + 2 + np_s;
    if constexpr (s == 8 || s == 9) {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(base + (HAS_MASK ? 4 : 0)));
    } else if constexpr (s == 15 || s == 0) {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(base + (HAS_STATS ? 24 : 16)));
    } else {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(base));
    }
  };
#define CP_INTERVAL(S)                                                                                      \
  do {                                                                                                      \
    constexpr int kc_ = (S) & 1, sn_ = ((S) + 1) & 15;                                                      \
    CP_SYNC();                                                                                              \
    issue_step(icp<(S)>{}, T);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    const int dn_ = dlt[(sn_ >> 1) & 3];                                                                    \
    const unsigned bufn_ = lds_base + ((sn_ & 8) ? CP_OFF_Y : CP_OFF_X);                                    \
    CP_READ_B(1 - kc_, rd_slot * CP_SLOT);                                                                  \
    rd_slot = rd_slot == 2 ? 0 : rd_slot + 1;                                                               \
    read_a(aS0, icp<0>{}, dn_, bufn_, icp<1 - kc_>{});                                                      \
    CP_MFMAS(4, aS1, kc_);                                                                                  \
    CP_WAIT4(bS[1 - kc_]);                                                                                  \
    CP_WAIT4(aS0);                                                                                          \
    CP_KEEP_ALL(aS1, kc_);                                                                                  \
    read_a(aS1, icp<4>{}, dn_, bufn_, icp<1 - kc_>{});                                                      \
    CP_MFMAS(0, aS0, 1 - kc_);                                                                              \
    CP_WAIT4(aS1);                                                                                          \
    CP_KEEP_ALL(aS0, 1 - kc_);                                                                              \
    wait_v(icp<(S)>{});                                                                                     \
  } while (0)
#define CP_LAST()                                                                                           \
  do {                                                                                                      \
    CP_SYNC();                                                                                              \
    issue_step(icp<15>{}, T);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    CP_MFMAS(4, aS1, 1);                                                                                    \
    epilogue(T);                                                                                            \
    CP_READ_B(0, rd_slot * CP_SLOT);                                                                        \
    rd_slot = rd_slot == 2 ? 0 : rd_slot + 1;                                                               \
    read_a(aS0, icp<0>{}, dlt[0], lds_base + CP_OFF_X, icp<0>{});                                           \
    read_a(aS1, icp<4>{}, dlt[0], lds_base + CP_OFF_X, icp<0>{});                                           \
    CP_WAIT4(bS[0]);                                                                                        \
    CP_WAIT4(aS0);                                                                                          \
    CP_MFMAS(0, aS0, 0);                                                                                    \
    CP_WAIT4(aS1);                                                                                          \
    CP_KEEP_ALL(aS0, 0);                                                                                    \
    wait_v(icp<15>{});                                                                                      \
  } while (0)

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    if (T + 1 < T1) tile_base(T + 1, tb_nxt, y0_nxt);
    CP_INTERVAL(0);  CP_INTERVAL(1);  CP_INTERVAL(2);  CP_INTERVAL(3);
    CP_INTERVAL(4);  CP_INTERVAL(5);  CP_INTERVAL(6);  CP_INTERVAL(7);
    CP_INTERVAL(8);  CP_INTERVAL(9);  CP_INTERVAL(10); CP_INTERVAL(11);
    CP_INTERVAL(12); CP_INTERVAL(13); CP_INTERVAL(14);
    CP_LAST();
    tb_cur = tb_nxt; y0_cur = y0_nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CP_LAST
#undef CP_INTERVAL
#undef CP_SYNC
#undef CP_ACC_FENCE
#undef CP_MFMAS
#undef CP_KEEP_ALL
#undef CP_WAIT4
#undef CP_READ_B
#undef CP_DSR
}

}  // namespace

// shapes convp_kernel takes: 128 input channels, 64 output channels, power-of-two maps of 16 .. 64 pixels width, whole
// 256-pixel tiles inside one image
bool rg_convp_supported(int M, int Ncols, int Cin, int Hs, int Ws) {
  return Ncols == 64 && Cin == 128 && Ws >= 16 && Ws <= 64 && rg_is_pow2(Ws) && rg_is_pow2(Hs) && (Hs * Ws) % 256 == 0 &&
         M % 256 == 0 && M >= 256;
}

int rg_convp_tiles(int M) { return M / 256; }

int rg_convp_launch(const void* args, hipStream_t st) {
  const G2Args& a2 = *reinterpret_cast<const G2Args*>(args);
  const int tiles = a2.g.M / 256;
  const int target = rg_option("convp_blocks", 256);
  int grid = tiles < target ? tiles : target;
  const int per = (tiles + grid - 1) / grid;             // equal tile counts per workgroup where the tile count allows it
  grid = (tiles + per - 1) / per;
  const int epi = a2.g.mask ? 2 : a2.g.affine ? 3 : a2.g.stats ? 1 : 0;
#define CP_GO(LGW_)                                                                                            \
  do {                                                                                                         \
    if (epi == 0) hipLaunchKernelGGL((convp_kernel<LGW_, 0>), dim3((unsigned)grid), dim3(512), 0, st, a2);      \
    else if (epi == 1) hipLaunchKernelGGL((convp_kernel<LGW_, 1>), dim3((unsigned)grid), dim3(512), 0, st, a2); \
    else if (epi == 2) hipLaunchKernelGGL((convp_kernel<LGW_, 2>), dim3((unsigned)grid), dim3(512), 0, st, a2); \
    else hipLaunchKernelGGL((convp_kernel<LGW_, 3>), dim3((unsigned)grid), dim3(512), 0, st, a2);               \
  } while (0)
  if (a2.g.Ws == 64) CP_GO(6);
  else if (a2.g.Ws == 32) CP_GO(5);
  else CP_GO(4);
#undef CP_GO
  return RG_OK;
}
