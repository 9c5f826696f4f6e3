// rg_conv8.hip -- 8-wave ping-pong implicit-GEMM conv kernel for gfx950 (bf16 MFMA 32x32x16, fp32 accumulate).
//
// Same GEMM view as gather_gemm_dma_kernel (rg_mfma.hip): C[M][Ncols] = sum_k A(m,k) * Bt[col][k], A rows gathered
// from an NHWC tensor (stride-2 4x4 conv, its transpose in 4 output-parity classes, or plain rows), operands DMA'd
// global -> LDS (buffer_load_dwordx4 ... lds), XOR-swizzled source chunk / linear destination / swizzled read.
// What differs is the pipeline (CDNA guide "256^2 8-phase template", re-derived for a gathered A operand):
//
//  * block tile BM x BN = (WM*128) x (WN*64) with 8 waves, every wave owns 128 x 64 of it as 2 x 2 QUADRANTS of
//    64 x 32; quadrant row i of every wave lives in A half-tile i, quadrant column j in B half-tile j, so a k-tile is
//    four half-tiles (A0, A1, B0, B1) that are staged, waited for, read and freed independently.
//  * a k-tile is 4 PHASES (one quadrant x BK = 64 each: 8 MFMAs per wave).  Phase = load section (fragment
//    ds_reads of the one half-tile this phase needs first, the DMA issue of ONE half-tile, a counted vmcnt) ->
//    s_barrier -> MFMA section -> s_barrier.  Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave
//    is in its MFMA section while its partner reads LDS / issues DMA (ping-pong).
//  * the DMA stream runs 5 half-tiles ahead of the half-tile being waited for (issue order B0 A0 B1 A1 per k-tile,
//    each issued 6 phases before its first read, into the slot freed 2 phases earlier); vmcnt is never 0 in the loop.
//      phase of k-tile u :  reads        issues        MFMA quadrant
//        P1                 A0[u]        A1[u+1]       (0,0)  with B0[u] read in P4 of k-tile u-1
//        P2                 B1[u]        B0[u+2]       (0,1)
//        P3                 A1[u]        A0[u+2]       (1,1)
//        P4                 B0[u+1]      B1[u+2]       (1,0)
//    k-tiles beyond the last one are issued as all-out-of-range DMAs (zero fill into free slots, no memory traffic):
//    the counted waits stay uniform, there is no tail variant of the loop.
//  * B fragments rotate through three register sets (B0[u], B1[u], B0[u+1]), period two k-tiles = the unroll.
#include "rg_gather.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int V> using ic = std::integral_constant<int, V>;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// MF: MFMA shape, 32 = v_mfma_f32_32x32x16_bf16 (8 per phase), 16 = v_mfma_f32_16x16x32_bf16 (16 per phase).
// EB: operand element bytes.  2 = bf16 (k-tile 64 deep).  1 = fp8 e4m3 (OCP) operands for generator-only inference
// (BASELINE configs[4]): the same 128-byte LDS rows hold a 128-deep k-tile, fragments are 64-bit
// (v_mfma_f32_16x16x32_fp8_fp8, 32 per phase), fp32 accumulate; the epilogue applies the folded BatchNorm affine and
// LeakyReLU (the affine also carries the per-column weight scale) and writes bf16 or fp8 (a2.g.out_fp8).
// MXF (fp8 only): the MX-format matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 -- one MFMA per 128-deep k-tile and
// accumulator tile at TWICE the per-clock rate of the non-scaled fp8 / bf16 forms (MI355X_MICROARCH, matrix cores) -- with
// UNIT block scales (E8M0 127 = 2^0 on both operands): the per-output-channel weight scales of the fp8 path stay in the
// epilogue, so the numbers are those of the non-scaled fp8 kernel up to the fp32 summation order.  The operand of a lane is
// 32 bytes: the two 16-byte chunks (fh, 4 + fh) of its row, i.e. exactly the two conflict-free ds_read_b128 of the bf16
// path; which k positions a lane's bytes stand for does not matter as long as A and B agree (both are read the same way).
typedef int c8_v8i_t __attribute__((ext_vector_type(8)));
// PROBE (rg_probe.hip only): the k-loop WITHOUT its LDS-DMA issue -- the prologue stages two k-tiles, the loop then runs
// a2.probe_iters k-tiles over those resident stages (fragment reads, counted waits, barriers and MFMAs unchanged): the rate the
// 8-wave schedule reaches when nothing has to arrive from L2 -- the measured ceiling bench.py quotes beside the nominal peak.
// NP (rg_conv8f.hip only; 0 = off): fp32 operands as K-concatenated bf16 PLANES.  An fp32 value is the exact sum of three bf16
// numbers v = h + m + l (rg_split_planes: h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)); A and B are plane-major buffers
// [3][...] of the layouts this kernel takes anyway, and the k loop runs NP times as many k-tiles: flat k-tile q -> plane pair
// q % NP, k-tile q / NP.  NP = 6: hh, hm, mh, hl, lh, mm -- every product down to 2^-24 |a b|, fp32-grade; NP = 3: hh, hm, mh --
// 2^-16 |a b|.  The matrix pipe, the schedule and the operand traffic per k-tile are the bf16 kernel's; accumulation is the same
// fp32 MFMA chain; the result (and the BatchNorm statistics, and split-K slabs) leave as fp32.
template <int MODE, int WM, int WN, int MF, int EB = 2, int MXF = 0, int PROBE = 0, int NP = 0>
__global__ __launch_bounds__(512, 2) void conv8_kernel(G2Args a2) {
  static_assert(WM * WN == 8, "8 waves");
  static_assert(EB == 2 || (EB == 1 && MF == 16), "fp8 operands use the 16x16x32 MFMA");
  static_assert(MXF == 0 || EB == 1, "the MX-format MFMA takes fp8 operands");
  constexpr bool Q8 = EB == 1 && MXF == 0;                   // non-scaled fp8: 64-bit fragments, 4 k-steps of 32
  constexpr int KD = 128 / EB;                               // k-tile depth in elements
  constexpr int LGKD = EB == 2 ? 6 : 7;
  constexpr int BM = WM * 128, BN = WN * 64;
  constexpr int AH_ROWS = BM / 2, BH_ROWS = BN / 2;          // rows per half-tile
  constexpr int AH = AH_ROWS * 128, BH = BH_ROWS * 128;      // bytes per half-tile (BK = 64 bf16 = 128 B per row)
  constexpr int STAGE = 2 * AH + 2 * BH;                     // bytes per stage, laid out [B0][B1][A0][A1]
  constexpr int OFF_B = 0, OFF_A = 2 * BH;
  constexpr int NA = AH_ROWS / 64, NB = BH_ROWS / 64;        // block-wide DMA instructions per half-tile (64 rows each)
  static_assert(NA >= 1 && NB >= 1, "half-tiles are whole 64-row DMA instructions");
  static_assert(AH + 4096 < 65536 && BH < 65536, "fragment offsets must fit the 16-bit ds_read offset");
  constexpr int LDS_BYTES = 2 * STAGE;
  constexpr int EP_COLS = BN / 2;                            // epilogue: one quadrant column per pass
  static_assert(BM * EP_COLS * 4 <= LDS_BYTES, "fp32 epilogue slice must fit the staging LDS");
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_BYTES / 16];
  const GArgs& g = a2.g;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  int bid = blockIdx.x;
  if (a2.xcd_swizzle) bid = (bid & 7) * ((int)gridDim.x >> 3) + (bid >> 3);
  int par = (MODE == MODE_UP) ? (int)blockIdx.y : 0;
  if (MODE == MODE_UP && a2.class_fast) { par = bid & 3; bid >>= 2; }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int ph = par >> 1, pw = par & 1;
  const int zs = blockIdx.z;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a2.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, a2.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- DMA lane assignment: physical chunk pc of row r0 + 64*j of a half-tile; logical (source) chunk lc
  const int pc = t & 7, r0 = t >> 3;
  const int lc = pc ^ ((r0 >> 1) & 7);
  const int Wq = 1 << g.lgW, Hq = 1 << g.lgH;
  int a_off[2 * NA];                 // byte offset of the row's base pixel + this lane's chunk
  unsigned a_mask[NA];               // tap-validity masks, two rows per register (h = 0 low half, h = 1 high half)
  int b_off[2 * NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) a_mask[j] = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int m = bm + h * AH_ROWS + j * 64 + r0;
      const bool ok = m < g.M;
      const int mm = ok ? m : 0;
      const int wq = mm & (Wq - 1), hq = (mm >> g.lgW) & (Hq - 1), n = mm >> (g.lgW + g.lgH);
      unsigned mask = 0;
      long long base;
      if (MODE == MODE_DOWN) {
        const int hs0 = 2 * hq - 1, ws0 = 2 * wq - 1;
        base = (((long long)n * g.Hs + hs0) * g.Ws + ws0) * g.Cin;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
          for (int kw = 0; kw < 4; ++kw) {
            const bool v = (unsigned)(hs0 + kh) < (unsigned)g.Hs && (unsigned)(ws0 + kw) < (unsigned)g.Ws;
            mask |= (v ? 1u : 0u) << (kh * 4 + kw);
          }
      } else if (MODE == MODE_UP) {
        base = (((long long)n * g.Hs + hq) * g.Ws + wq) * g.Cin;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            int kh, kw, dh, dw;
            up_tap_dev(ph, a, kh, dh);
            up_tap_dev(pw, b, kw, dw);
            const bool v = (unsigned)(hq + dh) < (unsigned)g.Hs && (unsigned)(wq + dw) < (unsigned)g.Ws;
            mask |= (v ? 1u : 0u) << (a * 2 + b);
          }
      } else {
        base = (long long)mm * g.Cin;
        mask = 1u;
      }
      a_off[h * NA + j] = (int)((base + lc * (16 / EB)) * EB);
      a_mask[j] |= (ok ? mask : 0u) << (16 * h);
    }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = bn + h * BH_ROWS + j * 64 + r0;
      b_off[h * NB + j] = col < g.Ncols ? (int)(((long long)col * g.b_col + lc * (16 / EB)) * EB) : -1;
    }

  const int cpt = g.Cin >> LGKD;
  const int nkt_all = g.taps * cpt * (NP > 0 ? NP : 1);
  const int per = (nkt_all + a2.nsplit - 1) / a2.nsplit;      // host guarantees an even count per split
  const int kt_begin = zs * per;
  const int kt_end = min(nkt_all, kt_begin + per);
  const int nkt = kt_end - kt_begin;
  const int lgcpt = a2.lgcpt, cmask = a2.cmask;
  char* const ldsb = reinterpret_cast<char*>(lds);

  // k-tile index (relative to kt_begin) -> byte offsets added to the A row bases / B column bases, and the tap
  auto decode = [&](int ktr, int& ao, int& bo, int& tap) {
    int kt = kt_begin + ktr;
    int pa = 0, pb = 0;                  // planes of A / B this k-tile multiplies (NP > 0)
    if constexpr (NP > 0) {
      const int pp = kt % NP;
      kt = kt / NP;
      pa = (0x120100 >> (4 * pp)) & 15;  // pairs (A plane, B plane): hh hm mh hl lh mm
      pb = (0x102010 >> (4 * pp)) & 15;
    }
    int cb;
    if (a2.korder && MODE != MODE_PLAIN) {
      const int ti = kt & (g.taps - 1);
      cb = kt >> (MODE == MODE_DOWN ? 4 : 2);
      tap = MODE == MODE_DOWN ? rg_down_tap(ti) : ti;
    } else {
      tap = kt >> lgcpt;
      cb = kt & cmask;
    }
    const int c0 = cb * KD;
    int a_delta, b_tap;
    if (MODE == MODE_DOWN) {
      a_delta = ((tap >> 2) * g.Ws + (tap & 3)) * g.Cin;
      b_tap = tap;
    } else if (MODE == MODE_UP) {
      int kh, kw, dh, dw;
      up_tap_dev(ph, tap >> 1, kh, dh);
      up_tap_dev(pw, tap & 1, kw, dw);
      a_delta = (dh * g.Ws + dw) * g.Cin;
      b_tap = kh * 4 + kw;
    } else {
      a_delta = 0;
      b_tap = 0;
    }
    ao = (a_delta + c0) * EB;
    bo = (b_tap * g.b_tap + c0) * EB;
    if constexpr (NP > 0) { ao += pa * (int)a2.a_plane; bo += pb * (int)a2.b_plane; }
  };
  // one half-tile of k-tile ktr into stage S: NA (A) or NB (B) block-wide DMA instructions
  auto issue_a = [&](auto S, auto H, int ktr) {
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    int ao, bo, tap;
    decode(ktr, ao, bo, tap);
    const unsigned dead = ktr < nkt ? 0u : OOB;         // k-tiles beyond the end: every lane out of range (branch-free)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool v = (a_mask[j] >> ((tap & 15) + 16 * h)) & 1u;
      const unsigned vo = (v ? (unsigned)(a_off[h * NA + j] + ao) : OOB) | dead;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsA, (lds_vptr_t)(ldsb + s * STAGE + OFF_A + h * AH + j * 8192 + wave * 1024), 16, vo, 0, 0, 0);
    }
  };
  auto issue_b = [&](auto S, auto H, int ktr) {
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    int ao, bo, tap;
    decode(ktr, ao, bo, tap);
    const unsigned dead = ktr < nkt ? 0u : OOB;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned vo = (b_off[h * NB + j] >= 0 ? (unsigned)(b_off[h * NB + j] + bo) : OOB) | dead;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsB, (lds_vptr_t)(ldsb + s * STAGE + OFF_B + h * BH + j * 8192 + wave * 1024), 16, vo, 0, 0, 0);
    }
  };

  // ---- MFMA / fragment-read assignment
  const int wm = wave / WN, wn = wave - wm * WN;
  // 32x32x16: lane = (row fr of 32, k-half fh), chunk(kk) = 2kk + fh, kk < 4.   16x16x32: lane = (row fr of 16, k-quarter
  // fh), chunk(kk) = 4kk + fh, kk < 2.  The XOR swizzle (row>>1)&7 is the same for every sub-tile of a wave (16 | 32 rows apart).
  constexpr int FRW = MF == 32 ? 32 : 16;            // rows per MFMA tile
  constexpr int NKK = Q8 ? 4 : (MF == 32 ? 4 : 2);   // k-steps per k-tile (fp8: 4 steps of 32; MX: two 16-byte reads, one MFMA)
  constexpr int NAT = 64 / FRW, NBT = 32 / FRW;      // A row sub-tiles / B column sub-tiles per quadrant
  const int fr = lane & (FRW - 1), fh = lane / FRW;
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;
  unsigned aB[2][NKK], bB[2][NKK];   // [stage][k-step] byte addresses of this lane's A / B fragment chunk
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    // bf16: lane's 16-byte chunk of the 128-byte row; fp8: 8 bytes (fh & 1) of chunk 2 kk + (fh >> 1)
    const unsigned ch = Q8 ? (unsigned)((2 * kk + (fh >> 1)) ^ ((fr >> 1) & 7))
                                : (unsigned)(((64 / FRW) * kk + fh) ^ ((fr >> 1) & 7));
    const unsigned sub = Q8 ? 8u * (unsigned)(fh & 1) : 0u;
    aB[0][kk] = lds_base + OFF_A + 16u * (unsigned)((wm * 64 + fr) * 8 + ch) + sub;
    bB[0][kk] = lds_base + OFF_B + 16u * (unsigned)((wn * 32 + fr) * 8 + ch) + sub;
    aB[1][kk] = aB[0][kk] + STAGE;
    bB[1][kk] = bB[0][kk] + STAGE;
  }
  using acc_t = std::conditional_t<MF == 32, f32x16_t, f32x4_t>;
  constexpr int NACC = NAT * NBT, ACC_R = MF == 32 ? 16 : 4;
  acc_t acc[2][2][NACC];             // [quadrant row i][quadrant column j][A sub-tile * NBT + B sub-tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int r = 0; r < ACC_R; ++r) acc[i][j][s][r] = 0.f;
  u32x4_t aR[8];                     // A fragments of the current quadrant row: [A sub-tile][k-step]
  u32x4_t bS[3][4];                  // three rotating B fragment sets: [set][B sub-tile][k-step]
  u32x2_t aQ[16];                    // fp8: 64-bit fragments, [A sub-tile (4)][k-step (4)]
  u32x2_t bQ[3][8];                  //      [set][B sub-tile (2)][k-step (4)]

#define C8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define C8_DSR8(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define C8_READ_A(S, H)                                                                     \
  do {                                                                                      \
    _Pragma("unroll") for (int t_ = 0; t_ < NAT; ++t_) _Pragma("unroll") for (int kk_ = 0; kk_ < NKK; ++kk_) { \
      if constexpr (Q8) C8_DSR8(aQ[t_ * NKK + kk_], aB[S][kk_], (H) * AH + t_ * FRW * 128); \
      else C8_DSR(aR[(t_ * NKK + kk_) & 7], aB[S][kk_], (H) * AH + t_ * FRW * 128);         \
    }                                                                                       \
  } while (0)
#define C8_READ_B(S, H, SET)                                                                \
  do {                                                                                      \
    _Pragma("unroll") for (int t_ = 0; t_ < NBT; ++t_) _Pragma("unroll") for (int kk_ = 0; kk_ < NKK; ++kk_) { \
      if constexpr (Q8) C8_DSR8(bQ[SET][t_ * NKK + kk_], bB[S][kk_], (H) * BH + t_ * FRW * 128); \
      else C8_DSR(bS[SET][(t_ * NKK + kk_) & 3], bB[S][kk_], (H) * BH + t_ * FRW * 128);    \
    }                                                                                       \
  } while (0)
#define C8_WAIT_A()                                                                                            \
  do {                                                                                                         \
    if constexpr (Q8)                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
                   : "+v"(aQ[0]), "+v"(aQ[1]), "+v"(aQ[2]), "+v"(aQ[3]), "+v"(aQ[4]), "+v"(aQ[5]), "+v"(aQ[6]), \
                     "+v"(aQ[7]), "+v"(aQ[8]), "+v"(aQ[9]), "+v"(aQ[10]), "+v"(aQ[11]), "+v"(aQ[12]),          \
                     "+v"(aQ[13]), "+v"(aQ[14]), "+v"(aQ[15])::"memory");                                      \
    else                                                                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
                   : "+v"(aR[0]), "+v"(aR[1]), "+v"(aR[2]), "+v"(aR[3]), "+v"(aR[4]), "+v"(aR[5]), "+v"(aR[6]), \
                     "+v"(aR[7])::"memory");                                                                   \
  } while (0)
#define C8_WAIT_B(SET)                                                                                         \
  do {                                                                                                         \
    if constexpr (Q8)                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
                   : "+v"(bQ[SET][0]), "+v"(bQ[SET][1]), "+v"(bQ[SET][2]), "+v"(bQ[SET][3]), "+v"(bQ[SET][4]), \
                     "+v"(bQ[SET][5]), "+v"(bQ[SET][6]), "+v"(bQ[SET][7])::"memory");                          \
    else                                                                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
                   : "+v"(bS[SET][0]), "+v"(bS[SET][1]), "+v"(bS[SET][2]), "+v"(bS[SET][3])::"memory");        \
  } while (0)
// C8_PRIO_MODE (build-time experiment knob): 0 = s_setprio(1) around every MFMA section (CDNA guide T5); 1 = static priority
// for the second-dispatched half (waves 4-7) set once before the loop, no per-section flips (MI355X_MICROARCH "two waves per
// SIMD", item 4); 2 = no priority changes
#ifndef C8_PRIO_MODE
#define C8_PRIO_MODE 0
#endif
#define C8_MFMAS(I, J, SET)                                                                                    \
  do {                                                                                                         \
    if (C8_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(1);                                                      \
    _Pragma("unroll") for (int kk_ = 0; kk_ < NKK; ++kk_) _Pragma("unroll") for (int t_ = 0; t_ < NAT; ++t_)   \
    _Pragma("unroll") for (int c_ = 0; c_ < NBT; ++c_) {                                                       \
      if constexpr (MXF == 1) {                                                                                \
        if (kk_ == 0) {                                                                                        \
          const u32x4_t a0_ = aR[(t_ * NKK) & 7], a1_ = aR[(t_ * NKK + 1) & 7];                                \
          const u32x4_t b0_ = bS[SET][(c_ * NKK) & 3], b1_ = bS[SET][(c_ * NKK + 1) & 3];                      \
          const c8_v8i_t av_ = {(int)a0_.x, (int)a0_.y, (int)a0_.z, (int)a0_.w, (int)a1_.x, (int)a1_.y, (int)a1_.z, (int)a1_.w}; \
          const c8_v8i_t bv_ = {(int)b0_.x, (int)b0_.y, (int)b0_.z, (int)b0_.w, (int)b1_.x, (int)b1_.y, (int)b1_.z, (int)b1_.w}; \
          acc[I][J][t_ * NBT + c_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                         \
              av_, bv_, acc[I][J][t_ * NBT + c_], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                         \
        }                                                                                                      \
      } else if constexpr (EB == 1)                                                                            \
        acc[I][J][t_ * NBT + c_] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(                                 \
            __builtin_bit_cast(long, aQ[t_ * NKK + kk_]), __builtin_bit_cast(long, bQ[SET][c_ * NKK + kk_]),   \
            acc[I][J][t_ * NBT + c_], 0, 0, 0);                                                                \
      else if constexpr (MF == 32)                                                                             \
        acc[I][J][t_ * NBT + c_] = rg_mfma_h16_32x32x16(                                    \
            __builtin_bit_cast(h16x8_t, aR[(t_ * NKK + kk_) & 7]), __builtin_bit_cast(h16x8_t, bS[SET][(c_ * NKK + kk_) & 3]), \
            acc[I][J][t_ * NBT + c_], 0, 0, 0);                                                                \
      else                                                                                                     \
        acc[I][J][t_ * NBT + c_] = rg_mfma_h16_16x16x32(                                    \
            __builtin_bit_cast(h16x8_t, aR[(t_ * NKK + kk_) & 7]), __builtin_bit_cast(h16x8_t, bS[SET][(c_ * NKK + kk_) & 3]), \
            acc[I][J][t_ * NBT + c_], 0, 0, 0);                                                                \
    }                                                                                                          \
    if (C8_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(0);                                                      \
  } while (0)
#define C8_SYNC() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // counted waits (see the table in the header): half-tiles allowed to stay in flight behind the one needed next phase
  constexpr int W_ODD = vmcnt_imm(3 * NA + 2 * NB);     // P1, P3
  constexpr int W_EVEN = vmcnt_imm(2 * NA + 3 * NB);    // P2, P4
  // one k-tile u living in stage S; B0[u] is in set SB0, B1[u] goes to set 1, B0[u+1] goes to set SNX
#define C8_TILE(S, SB0, SNX, U)                                                             \
  do {                                                                                      \
    /* P1 */                                                                                \
    C8_READ_A(S, 0);                                                                        \
    if constexpr (!PROBE) issue_a(ic<1 - (S)>{}, ic<1>{}, (U) + 1);                        \
    __builtin_amdgcn_s_waitcnt(W_ODD);                                                      \
    C8_SYNC();                                                                              \
    C8_WAIT_A();                                                                            \
    C8_MFMAS(0, 0, SB0);                                                                    \
    C8_SYNC();                                                                              \
    /* P2 */                                                                                \
    C8_READ_B(S, 1, 1);                                                                     \
    if constexpr (!PROBE) issue_b(ic<(S)>{}, ic<0>{}, (U) + 2);                            \
    __builtin_amdgcn_s_waitcnt(W_EVEN);                                                     \
    C8_SYNC();                                                                              \
    C8_WAIT_B(1);                                                                           \
    C8_MFMAS(0, 1, 1);                                                                      \
    C8_SYNC();                                                                              \
    /* P3 */                                                                                \
    C8_READ_A(S, 1);                                                                        \
    if constexpr (!PROBE) issue_a(ic<(S)>{}, ic<0>{}, (U) + 2);                            \
    __builtin_amdgcn_s_waitcnt(W_ODD);                                                      \
    C8_SYNC();                                                                              \
    C8_WAIT_A();                                                                            \
    C8_MFMAS(1, 1, 1);                                                                      \
    C8_SYNC();                                                                              \
    /* P4 */                                                                                \
    C8_READ_B(1 - (S), 0, SNX);                                                             \
    if constexpr (!PROBE) issue_b(ic<(S)>{}, ic<1>{}, (U) + 2);                            \
    __builtin_amdgcn_s_waitcnt(W_EVEN);                                                     \
    C8_SYNC();                                                                              \
    C8_WAIT_B(SNX);                                                                         \
    C8_MFMAS(1, 0, SB0);                                                                    \
    C8_SYNC();                                                                              \
  } while (0)

  // ---- prologue: k-tile 0 and the first three half-tiles of k-tile 1 in flight
  issue_b(ic<0>{}, ic<0>{}, 0);
  issue_a(ic<0>{}, ic<0>{}, 0);
  issue_b(ic<0>{}, ic<1>{}, 0);
  issue_a(ic<0>{}, ic<1>{}, 0);
  issue_b(ic<1>{}, ic<0>{}, 1);
  issue_a(ic<1>{}, ic<0>{}, 1);
  issue_b(ic<1>{}, ic<1>{}, 1);
  if constexpr (PROBE) {                                       // the loop issues nothing: both stages complete before it starts
    issue_a(ic<1>{}, ic<1>{}, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * NA + 3 * NB));      // B0[0] and A0[0] landed (this wave's part)
  C8_SYNC();
  C8_READ_B(0, 0, 0);                                          // "P4 of k-tile -1"
  C8_WAIT_B(0);
  if (wave >= 4) __builtin_amdgcn_s_barrier();                 // waves 4-7 run one barrier behind waves 0-3
  if (C8_PRIO_MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  __builtin_amdgcn_sched_barrier(0);
  const int nloop = PROBE ? a2.probe_iters : nkt;
  for (int u = 0; u < nloop; u += 2) {
    C8_TILE(0, 0, 2, u);
    C8_TILE(1, 2, 0, u + 1);
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#undef C8_TILE
#undef C8_SYNC
#undef C8_MFMAS
#undef C8_WAIT_A
#undef C8_WAIT_B
#undef C8_READ_A
#undef C8_READ_B
#undef C8_DSR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the zero-fill DMAs of the k-tiles beyond the end
  __syncthreads();

  // ---- BatchNorm statistics from the accumulators (see gather_gemm_dma_kernel): a lane's registers of quadrant
  // column j all belong to ONE output column; the wave's 128 rows are register adds + one cross-half shuffle.
  if (g.stats && a2.nsplit == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < NBT; ++c) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < NAT; ++s)
#pragma unroll
            for (int r = 0; r < ACC_R; ++r) {
              const float v = NP > 0 ? acc[i][j][s * NBT + c][r] : h16_to_f32(f32_to_h16(acc[i][j][s * NBT + c][r]));
              s1 += v; s2 += v * v;
            }
#pragma unroll
        for (int o = FRW; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const int gcol = bn + j * BH_ROWS + wn * 32 + c * FRW + fr;
        if (fh == 0 && gcol < g.Ncols) {
          const size_t grow = ((size_t)par * a2.tiles_m + tile_m) * WM + wm;
          g.stats[(grow * 2 + 0) * g.Ncols + gcol] = s1;
          g.stats[(grow * 2 + 1) * g.Ncols + gcol] = s2;
        }
      }
  }

  // ---- epilogue through LDS: fp32 [BM][EP_COLS] per quadrant column, 16-byte row-contiguous global stores
  float* cs = reinterpret_cast<float*>(lds);
  constexpr int CG = EP_COLS / 8, RPP = 512 / CG;      // column groups per row, rows per pass
  const int cg = t % CG, rr = t / CG;
#pragma unroll
  for (int ep = 0; ep < 2; ++ep) {
    if (ep) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < NAT; ++s)
#pragma unroll
        for (int c = 0; c < NBT; ++c)
#pragma unroll
          for (int r = 0; r < ACC_R; ++r) {
            // accumulator row maps: 32x32 -> (r&3) + 8*(r>>2) + 4*fh ; 16x16 -> 4*fh + r
            const int rloc = MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * fh : 4 * fh + r;
            const int row = i * AH_ROWS + wm * 64 + s * FRW + rloc;
            cs[row * EP_COLS + wn * 32 + c * FRW + fr] = acc[i][ep][s * NBT + c][r];
          }
    __syncthreads();
    const int col = bn + ep * EP_COLS + cg * 8;
    // consumer's BatchNorm-backward sums (see GArgs::bwd_z): this thread's 8 columns over its BM / RPP rows
    float bs1[8], bs2[8], bmu[8], brs[8], bga[8], bbe[8];
    const bool bwd = EB == 2 && g.bwd_z != nullptr && col < g.Ncols;
    if (EB == 2 && g.bwd_z != nullptr) {
      const int grp_off = (g.bwd_half_m > 0 && bm >= g.bwd_half_m) ? g.Ncols : 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bs1[i] = 0.f; bs2[i] = 0.f;
        const int c = col + i < g.Ncols ? col + i : 0;
        bmu[i] = g.bwd_mean[grp_off + c]; brs[i] = g.bwd_invstd[grp_off + c]; bga[i] = g.bwd_gamma[c]; bbe[i] = g.bwd_beta[c];
      }
    }
#pragma unroll 4
    for (int p = 0; p < BM / RPP; ++p) {
      const int row = rr + RPP * p;
      const int m = bm + row;
      if (m >= g.M || col >= g.Ncols) continue;
      float4 v0 = *reinterpret_cast<const float4*>(cs + row * EP_COLS + cg * 8);
      float4 v1 = *reinterpret_cast<const float4*>(cs + row * EP_COLS + cg * 8 + 4);
      long long orow;
      if (MODE == MODE_UP) {
        const int wq = m & (Wq - 1), hq = (m >> g.lgW) & (Hq - 1), n = m >> (g.lgW + g.lgH);
        orow = ((long long)n * (2 * Hq) + 2 * hq + ph) * (2 * Wq) + 2 * wq + pw;
      } else {
        orow = m;
      }
      if constexpr (NP > 0) {            // fp32 result (or fp32 partial tile): two 16-byte stores per thread and row
        float* so = a2.nsplit > 1 ? a2.slab + (long long)zs * a2.slab_stride + orow * g.ldc + col
                                  : reinterpret_cast<float*>(g.C) + orow * g.ldc + col;
        *reinterpret_cast<float4*>(so) = v0;
        *reinterpret_cast<float4*>(so + 4) = v1;
        continue;
      }
      if (a2.nsplit > 1) {
        if (a2.slab16) {                 // bf16 partial tile: 16 instead of 32 bytes per thread and row (rounded partial sums)
          uint4 o;
          o.x = (uint32_t)f32_to_h16(v0.x) | ((uint32_t)f32_to_h16(v0.y) << 16);
          o.y = (uint32_t)f32_to_h16(v0.z) | ((uint32_t)f32_to_h16(v0.w) << 16);
          o.z = (uint32_t)f32_to_h16(v1.x) | ((uint32_t)f32_to_h16(v1.y) << 16);
          o.w = (uint32_t)f32_to_h16(v1.z) | ((uint32_t)f32_to_h16(v1.w) << 16);
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a2.slab) + (long long)zs * a2.slab_stride + orow * g.ldc + col) = o;
          continue;
        }
        float* so = a2.slab + (long long)zs * a2.slab_stride + orow * g.ldc + col;
        *reinterpret_cast<float4*>(so) = v0;
        *reinterpret_cast<float4*>(so + 4) = v1;
      } else {
        if (g.mask) {
          // (round 5, measured and rejected: issuing a pass's 8 mask loads in front of the LDS staging, so that their waits do
          // not drain the output stores once per row pass -- the step got 0.1 ms SLOWER in a three-round build A/B)
          const uint4 a = *reinterpret_cast<const uint4*>(g.mask + orow * g.ldc + col);
          v0.x *= rg_lmask(a.x, g.mslope); v0.y *= rg_lmask(a.x >> 16, g.mslope);
          v0.z *= rg_lmask(a.y, g.mslope); v0.w *= rg_lmask(a.y >> 16, g.mslope);
          v1.x *= rg_lmask(a.z, g.mslope); v1.y *= rg_lmask(a.z >> 16, g.mslope);
          v1.z *= rg_lmask(a.w, g.mslope); v1.w *= rg_lmask(a.w >> 16, g.mslope);
        }
        if (g.affine) rg_affine8(v0, v1, g.scale + col, g.shift + col, g.slope);
        if (g.out_fp8) {                 // 8 OCP e4m3 values = 8 bytes (the next fp8 layer's A operand)
          int lo = 0, hi = 0;
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(v0.x, v0.y, lo, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(v0.z, v0.w, lo, true);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(v1.x, v1.y, hi, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(v1.z, v1.w, hi, true);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(g.C) + orow * g.ldc + col) = make_uint2((unsigned)lo, (unsigned)hi);
          continue;
        }
        uint4 o;
        o.x = (uint32_t)f32_to_h16(v0.x) | ((uint32_t)f32_to_h16(v0.y) << 16);
        o.y = (uint32_t)f32_to_h16(v0.z) | ((uint32_t)f32_to_h16(v0.w) << 16);
        o.z = (uint32_t)f32_to_h16(v1.x) | ((uint32_t)f32_to_h16(v1.y) << 16);
        o.w = (uint32_t)f32_to_h16(v1.z) | ((uint32_t)f32_to_h16(v1.w) << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.C) + orow * g.ldc + col) = o;
        if (bwd) {
          // BwdRedF's arithmetic on the STORED (bf16-rounded) gradient and the consumer's stored z
          const uint4 zz = *reinterpret_cast<const uint4*>(g.bwd_z + orow * g.ldc + col);
          const uint32_t od[4] = {o.x, o.y, o.z, o.w}, zd[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int i = 2 * e + hh;
              const float gaf = hh ? h16hi_to_f32(od[e]) : h16lo_to_f32(od[e]);
              const float zf = hh ? h16hi_to_f32(zd[e]) : h16lo_to_f32(zd[e]);
              const float xh = (zf - bmu[i]) * brs[i];
              const float gy = gaf * lrelu_mask(xh * bga[i] + bbe[i], g.bwd_slope);
              bs1[i] += gy; bs2[i] += gy * xh;
            }
          }
        }
      }
    }
    if (EB == 2 && g.bwd_z != nullptr) {
      // block-tile column sums: the RPP row lanes of a column group through LDS in fixed order -> one partial row per tile
      __syncthreads();                                    // every thread is done reading its rows of cs
      float* red = cs;                                    // [RPP][EP_COLS][2]
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red[(rr * EP_COLS + cg * 8 + i) * 2 + 0] = bs1[i];
        red[(rr * EP_COLS + cg * 8 + i) * 2 + 1] = bs2[i];
      }
      __syncthreads();
      if (t < 2 * EP_COLS) {
        const int q = t / EP_COLS, c = t - q * EP_COLS;
        float sum = 0.f;
#pragma unroll 8
        for (int k = 0; k < RPP; ++k) sum += red[(k * EP_COLS + c) * 2 + q];
        const int gcol = bn + ep * EP_COLS + c;
        if (gcol < g.Ncols) {
          const size_t prow = (size_t)par * a2.tiles_m + tile_m;
          g.bwd_sums[(prow * 2 + q) * g.Ncols + gcol] = sum;
        }
      }
    }
  }
}


#ifndef RG_CONV8_KERNEL_ONLY
// ================================================================================================================
// conv8n_kernel: the transposed conv with 64 output channels (the generator's last MFMA layer 128 -> 64 at 128 x 128 and
// the discriminator's data gradient of layer 1).  K per parity class is only 4 taps x Cin (8 k-tiles at Cin = 128), so
// a block that computes one class of one row tile spends more time filling and draining its pipeline than computing.
// Here a block keeps its 512 low-resolution pixels and walks ALL FOUR parity classes as one stream of 16 "virtual
// taps" (class-major): the DMA prefetch runs across the class boundaries, and at each boundary the waves flush their
// accumulators through a wave-private 2 KB LDS scratch (no block barrier) while the next class's tiles are already in
// flight.  Block tile 512 x 64, 8 waves of 64 x 64 (quadrants 32 x 32, v_mfma_f32_16x16x32_bf16), A half-tile = 256
// rows (4 block-wide DMA instructions), B half-tile = 32 rows (issued by waves 0-3 only: their counted waits differ).
// ================================================================================================================
__global__ __launch_bounds__(512, 2) void conv8n_kernel(G2Args a2) {
  constexpr int AH = 256 * 128, BH = 32 * 128;               // bytes per half-tile
  constexpr int STAGE = 2 * AH + 2 * BH;                     // 72 KB, [B0][B1][A0][A1]
  constexpr int OFF_B = 0, OFF_A = 2 * BH;
  constexpr int SCRATCH = 2 * STAGE;                         // 8 x 2 KB wave-private epilogue scratch behind the stages
  constexpr int LDS_BYTES = SCRATCH + 8 * 2048;              // 160 KB
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_BYTES / 16];
  const GArgs& g = a2.g;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  int bid = blockIdx.x;
  if (a2.xcd_swizzle) bid = (bid & 7) * ((int)gridDim.x >> 3) + (bid >> 3);
  const int tile_m = bid;
  const int bm = tile_m * 512;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a2.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, a2.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  const int pc = t & 7, r0 = t >> 3;
  const int lc = pc ^ ((r0 >> 1) & 7);
  const int Wq = 1 << g.lgW, Hq = 1 << g.lgH;
  int a_off[8];                      // [half][j]: byte offset of the row's own pixel + this lane's chunk
  unsigned a_mask[4];                // 16 virtual-tap bits per row, two rows (half 0 low / half 1 high) per register
#pragma unroll
  for (int j = 0; j < 4; ++j) a_mask[j] = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = bm + h * 256 + j * 64 + r0;
      const bool ok = m < g.M;
      const int mm = ok ? m : 0;
      const int wq = mm & (Wq - 1), hq = (mm >> g.lgW) & (Hq - 1), n = mm >> (g.lgW + g.lgH);
      unsigned mask = 0;
#pragma unroll
      for (int vt = 0; vt < 16; ++vt) {
        int kh, kw, dh, dw;
        up_tap_dev(vt >> 3, (vt >> 1) & 1, kh, dh);          // class = vt >> 2 = (ph, pw); tap = vt & 3 = (a, b)
        up_tap_dev((vt >> 2) & 1, vt & 1, kw, dw);
        const bool v = (unsigned)(hq + dh) < (unsigned)g.Hs && (unsigned)(wq + dw) < (unsigned)g.Ws;
        mask |= (v ? 1u : 0u) << vt;
      }
      a_off[h * 4 + j] = (int)(((((long long)n * g.Hs + hq) * g.Ws + wq) * g.Cin + lc * 8) * 2);
      a_mask[j] |= (ok ? mask : 0u) << (16 * h);
    }
  int b_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) b_off[h] = (int)(((long long)(h * 32 + (r0 & 31)) * g.b_col + lc * 8) * 2);

  const int cpt = g.Cin >> 6;
  const int nkc = 4 * cpt;                                    // k-tiles per class
  const int nkt = 16 * cpt;
  const int lgcpt = a2.lgcpt, cmask = a2.cmask;
  char* const ldsb = reinterpret_cast<char*>(lds);

  auto decode = [&](int kt, int& ao, int& bo, int& vt) {
    vt = (kt >> lgcpt) & 15;
    const int c0 = (kt & cmask) << 6;
    int kh, kw, dh, dw;
    up_tap_dev(vt >> 3, (vt >> 1) & 1, kh, dh);
    up_tap_dev((vt >> 2) & 1, vt & 1, kw, dw);
    ao = ((dh * g.Ws + dw) * g.Cin + c0) * 2;
    bo = ((kh * 4 + kw) * g.b_tap + c0) * 2;
  };
  auto issue_a = [&](auto S, auto H, int kt) {
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    int ao, bo, vt;
    decode(kt, ao, bo, vt);
    const unsigned dead = kt < nkt ? 0u : OOB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool v = (a_mask[j] >> (vt + 16 * h)) & 1u;
      const unsigned vo = (v ? (unsigned)(a_off[h * 4 + j] + ao) : OOB) | dead;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsA, (lds_vptr_t)(ldsb + s * STAGE + OFF_A + h * AH + j * 8192 + wave * 1024), 16, vo, 0, 0, 0);
    }
  };
  auto issue_b = [&](auto S, auto H, int kt) {                // waves 0-3 only (32 rows = half a block-wide instruction)
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    if (wave < 4) {
      int ao, bo, vt;
      decode(kt, ao, bo, vt);
      const unsigned dead = kt < nkt ? 0u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr_t)(ldsb + s * STAGE + OFF_B + h * BH + wave * 1024), 16,
                                               (unsigned)(b_off[h] + bo) | dead, 0, 0, 0);
    }
  };

  // ---- fragments (16x16x32: lane = (row fr of 16, k-quarter fh))
  const int fr = lane & 15, fh = lane >> 4;
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;
  unsigned aB[2][2], bB[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const unsigned ch = (unsigned)((4 * kk + fh) ^ ((fr >> 1) & 7));
    aB[0][kk] = lds_base + OFF_A + 16u * (unsigned)((wave * 32 + fr) * 8 + ch);
    bB[0][kk] = lds_base + OFF_B + 16u * (unsigned)(fr * 8 + ch);
    aB[1][kk] = aB[0][kk] + STAGE;
    bB[1][kk] = bB[0][kk] + STAGE;
  }
  f32x4_t acc[2][2][4];              // [quadrant row i][quadrant column j][row sub-tile * 2 + column sub-tile]
  u32x4_t aR[4];                     // [row sub-tile][k-step]
  u32x4_t bS[3][4];                  // [set][column sub-tile][k-step]

#define N8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define N8_READ_A(S, H)                                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)       \
        N8_DSR(aR[t_ * 2 + kk_], aB[S][kk_], (H) * AH + t_ * 2048);                                            \
  } while (0)
#define N8_READ_B(S, H, SET)                                                                                   \
  do {                                                                                                         \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)       \
        N8_DSR(bS[SET][t_ * 2 + kk_], bB[S][kk_], (H) * BH + t_ * 2048);                                       \
  } while (0)
#define N8_WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aR[0]), "+v"(aR[1]), "+v"(aR[2]), "+v"(aR[3])::"memory")
#define N8_WAIT_B(SET) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bS[SET][0]), "+v"(bS[SET][1]), "+v"(bS[SET][2]), "+v"(bS[SET][3])::"memory")
#define N8_MFMAS(I, J, SET)                                                                                    \
  do {                                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_) _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)       \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                                           \
        acc[I][J][t_ * 2 + c_] = rg_mfma_h16_16x16x32(                                      \
            __builtin_bit_cast(h16x8_t, aR[t_ * 2 + kk_]), __builtin_bit_cast(h16x8_t, bS[SET][c_ * 2 + kk_]), \
            acc[I][J][t_ * 2 + c_], 0, 0, 0);                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)
#define N8_SYNC() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // counted waits: NA = 4 DMA instructions per A half-tile, NB = 1 (waves 0-3) or 0 (waves 4-7) per B half-tile
#define N8_WAITV(ODD)                                                                        \
  do {                                                                                       \
    if (wave < 4) __builtin_amdgcn_s_waitcnt((ODD) ? vmcnt_imm(14) : vmcnt_imm(11));         \
    else __builtin_amdgcn_s_waitcnt((ODD) ? vmcnt_imm(12) : vmcnt_imm(8));                   \
  } while (0)
#define N8_TILE(S, SB0, SNX, U)                                                              \
  do {                                                                                       \
    N8_READ_A(S, 0);                                                                         \
    issue_a(ic<1 - (S)>{}, ic<1>{}, (U) + 1);                                                \
    N8_WAITV(1);                                                                             \
    N8_SYNC();                                                                               \
    N8_WAIT_A();                                                                             \
    N8_MFMAS(0, 0, SB0);                                                                     \
    N8_SYNC();                                                                               \
    N8_READ_B(S, 1, 1);                                                                      \
    issue_b(ic<(S)>{}, ic<0>{}, (U) + 2);                                                    \
    N8_WAITV(0);                                                                             \
    N8_SYNC();                                                                               \
    N8_WAIT_B(1);                                                                            \
    N8_MFMAS(0, 1, 1);                                                                       \
    N8_SYNC();                                                                               \
    N8_READ_A(S, 1);                                                                         \
    issue_a(ic<(S)>{}, ic<0>{}, (U) + 2);                                                    \
    N8_WAITV(1);                                                                             \
    N8_SYNC();                                                                               \
    N8_WAIT_A();                                                                             \
    N8_MFMAS(1, 1, 1);                                                                       \
    N8_SYNC();                                                                               \
    N8_READ_B(1 - (S), 0, SNX);                                                              \
    issue_b(ic<(S)>{}, ic<1>{}, (U) + 2);                                                    \
    N8_WAITV(0);                                                                             \
    N8_SYNC();                                                                               \
    N8_WAIT_B(SNX);                                                                          \
    N8_MFMAS(1, 0, SB0);                                                                     \
    N8_SYNC();                                                                               \
  } while (0)

  // ---- epilogue geometry: a flush walks the wave's 64 rows in pieces of 8 rows x 64 columns of fp32 (2 KB); in the
  // read-back lane L owns row L >> 3 of the piece and 8 consecutive channels (L & 7) * 8
  const int e_row = lane >> 3, e_c8 = (lane & 7) * 8;
  const unsigned scr_base = lds_base + SCRATCH + wave * 2048;
  const unsigned scr_w = scr_base + (unsigned)(((4 * (fh & 1)) * 64 + fr) * 4);       // + (r*64 + j*32 + c*16) * 4
  const unsigned scr_r = scr_base + (unsigned)((e_row * 64 + e_c8) * 4);
  long long e_orow[8];               // output row (class (0,0)) of this lane's read-back row in each of the 8 pieces
  bool e_ok[8];
#pragma unroll
  for (int pz = 0; pz < 8; ++pz) {                            // piece = (i, row sub-tile rt, half p): rows i*256 + wave*32 + rt*16 + 8p + e_row
    const int m = bm + (pz >> 2) * 256 + wave * 32 + ((pz >> 1) & 1) * 16 + (pz & 1) * 8 + e_row;
    e_ok[pz] = m < g.M;
    const int mm = e_ok[pz] ? m : 0;
    const int wq = mm & (Wq - 1), hq = (mm >> g.lgW) & (Hq - 1), n = mm >> (g.lgW + g.lgH);
    e_orow[pz] = ((long long)n * (2 * Hq) + 2 * hq) * (2 * Wq) + 2 * wq;
  }
  // Fused LeakyReLU backward (data gradient of the discriminator's layer 1): the 8 mask vectors a lane needs for a class
  // are loaded (inline asm: an ordinary load waited for by the compiler would drain the LDS-DMA queue) two k-tiles
  // BEFORE the class ends and waited for with a counted vmcnt in the flush: the DMA instructions of those two k-tiles
  // (2 x (4 + 4 + NB + NB)) are younger and stay in flight.
  u32x4_t mreg[8];
  auto load_masks = [&](int cls) {
    const long long cls_off = (long long)(cls >> 1) * (2 * Wq) + (cls & 1);
#pragma unroll
    for (int pz = 0; pz < 8; ++pz) {
      const uint16_t* mp = g.mask + (e_orow[pz] + cls_off) * 64 + e_c8;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(mreg[pz]) : "v"(mp) : "memory");
    }
  };
  auto flush = [&](int cls) {
    const int ph = cls >> 1, pw = cls & 1;
    const long long cls_off = (long long)ph * (2 * Wq) + pw;
    if (g.mask) {
      if (wave < 4)
        asm volatile("s_waitcnt vmcnt(20)" : "+v"(mreg[0]), "+v"(mreg[1]), "+v"(mreg[2]), "+v"(mreg[3]), "+v"(mreg[4]),
                     "+v"(mreg[5]), "+v"(mreg[6]), "+v"(mreg[7])::"memory");
      else
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(mreg[0]), "+v"(mreg[1]), "+v"(mreg[2]), "+v"(mreg[3]), "+v"(mreg[4]),
                     "+v"(mreg[5]), "+v"(mreg[6]), "+v"(mreg[7])::"memory");
    }
    // BatchNorm partial sums of this class (column sums of the bf16-rounded values over the wave's 64 rows)
    if (g.stats) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = h16_to_f32(f32_to_h16(acc[i][j][rt * 2 + c][r]));
                s1 += v; s2 += v * v;
              }
          s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (fh == 0) {
            const size_t grow = ((size_t)cls * a2.tiles_m + tile_m) * 8 + wave;
            const int gcol = j * 32 + c * 16 + fr;
            g.stats[(grow * 2 + 0) * 64 + gcol] = s1;
            g.stats[(grow * 2 + 1) * 64 + gcol] = s2;
          }
        }
    }
#pragma unroll
    for (int pz = 0; pz < 8; ++pz) {
      const int i = pz >> 2, rt = (pz >> 1) & 1, p = pz & 1;
      // accumulator rows of a 16x16 tile: 4*fh + r; piece p holds rows 8p .. 8p+7, i.e. lanes with fh >> 1 == p.
      // Scratch accesses are inline asm: compiler-visible LDS accesses behind outstanding LDS-DMAs get an
      // s_waitcnt vmcnt(0) in front (hipcc assumes they alias), which would drain the prefetch at every class boundary.
      if ((fh >> 1) == p) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(scr_w), "v"(acc[i][j][rt * 2 + c][r]),
                           "n"(r * 256 + j * 128 + c * 64) : "memory");
      }
      // (same wave, LDS executes a wave's instructions in order: the reads see the writes without a barrier)
      u32x4_t q0, q1;
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(q0) : "v"(scr_r) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(q1) : "v"(scr_r) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1)::"memory");
      float4 v0 = __builtin_bit_cast(float4, q0), v1 = __builtin_bit_cast(float4, q1);
      if (e_ok[pz]) {
        const long long orow = e_orow[pz] + cls_off;
        if (g.mask) {
          const u32x4_t a = mreg[pz];
          v0.x *= rg_lmask(a.x, g.mslope); v0.y *= rg_lmask(a.x >> 16, g.mslope);
          v0.z *= rg_lmask(a.y, g.mslope); v0.w *= rg_lmask(a.y >> 16, g.mslope);
          v1.x *= rg_lmask(a.z, g.mslope); v1.y *= rg_lmask(a.z >> 16, g.mslope);
          v1.z *= rg_lmask(a.w, g.mslope); v1.w *= rg_lmask(a.w >> 16, g.mslope);
        }
        uint4 o;
        o.x = (uint32_t)f32_to_h16(v0.x) | ((uint32_t)f32_to_h16(v0.y) << 16);
        o.y = (uint32_t)f32_to_h16(v0.z) | ((uint32_t)f32_to_h16(v0.w) << 16);
        o.z = (uint32_t)f32_to_h16(v1.x) | ((uint32_t)f32_to_h16(v1.y) << 16);
        o.w = (uint32_t)f32_to_h16(v1.z) | ((uint32_t)f32_to_h16(v1.w) << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.C) + orow * 64 + e_c8) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][s][r] = 0.f;
  };

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][s][r] = 0.f;

  issue_b(ic<0>{}, ic<0>{}, 0);
  issue_a(ic<0>{}, ic<0>{}, 0);
  issue_b(ic<0>{}, ic<1>{}, 0);
  issue_a(ic<0>{}, ic<1>{}, 0);
  issue_b(ic<1>{}, ic<0>{}, 1);
  issue_a(ic<1>{}, ic<0>{}, 1);
  issue_b(ic<1>{}, ic<1>{}, 1);
  N8_WAITV(0);                                                 // 2 NA + 3 NB behind A0[0]
  N8_SYNC();
  N8_READ_B(0, 0, 0);
  N8_WAIT_B(0);
  if (wave >= 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int cls = 0; cls < 4; ++cls) {
    const int u0 = cls * nkc;
#pragma unroll 1
    for (int u = u0; u < u0 + nkc; u += 2) {
      if (g.mask && u == u0 + nkc - 2) load_masks(cls);
      N8_TILE(0, 0, 2, u);
      N8_TILE(1, 2, 0, u + 1);
    }
    flush(cls);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();
#undef N8_TILE
#undef N8_WAITV
#undef N8_SYNC
#undef N8_MFMAS
#undef N8_WAIT_A
#undef N8_WAIT_B
#undef N8_READ_A
#undef N8_READ_B
#undef N8_DSR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif  // RG_CONV8_KERNEL_ONLY

}  // namespace

#ifndef RG_CONV8_KERNEL_ONLY
// transposed conv with 64 output channels, all four parity classes per block (conv8n_kernel): grid = row tiles of 512
int rg_conv8n_launch(const void* args, unsigned tiles_m, hipStream_t st) {
  const G2Args& a2 = *reinterpret_cast<const G2Args*>(args);
  hipLaunchKernelGGL(conv8n_kernel, dim3(tiles_m), dim3(512), 0, st, a2);
  return RG_OK;
}

int rg_conv8_launch(int mode, const void* args, int bm, unsigned gx, unsigned gy, unsigned gz, hipStream_t st) {
  const G2Args& a2 = *reinterpret_cast<const G2Args*>(args);
  const dim3 grid(gx, gy, gz), block(512);
  if (a2.g.in_fp8) {                    // fp8 operands (generator-only inference): transposed conv and plain GEMM
    // fp8_mx (default 1): the MX-format MFMA with unit block scales (2x the matrix rate); 0: v_mfma_f32_16x16x32_fp8_fp8
    if (rg_option("fp8_mx", 1)) {
      if (bm == 256) {
        if (mode == MODE_UP) hipLaunchKernelGGL((conv8_kernel<MODE_UP, 2, 4, 16, 1, 1>), grid, block, 0, st, a2);
        else hipLaunchKernelGGL((conv8_kernel<MODE_PLAIN, 2, 4, 16, 1, 1>), grid, block, 0, st, a2);
      } else {
        if (mode == MODE_UP) hipLaunchKernelGGL((conv8_kernel<MODE_UP, 4, 2, 16, 1, 1>), grid, block, 0, st, a2);
        else hipLaunchKernelGGL((conv8_kernel<MODE_PLAIN, 4, 2, 16, 1, 1>), grid, block, 0, st, a2);
      }
      return RG_OK;
    }
    if (bm == 256) {
      if (mode == MODE_UP) hipLaunchKernelGGL((conv8_kernel<MODE_UP, 2, 4, 16, 1>), grid, block, 0, st, a2);
      else hipLaunchKernelGGL((conv8_kernel<MODE_PLAIN, 2, 4, 16, 1>), grid, block, 0, st, a2);
    } else {
      if (mode == MODE_UP) hipLaunchKernelGGL((conv8_kernel<MODE_UP, 4, 2, 16, 1>), grid, block, 0, st, a2);
      else hipLaunchKernelGGL((conv8_kernel<MODE_PLAIN, 4, 2, 16, 1>), grid, block, 0, st, a2);
    }
    return RG_OK;
  }
  const int mf = rg_option("conv8_mfma", 16);
#define C8_GO(MODE_, WM_, WN_)                                                                       \
  do {                                                                                               \
    if (mf == 16) hipLaunchKernelGGL((conv8_kernel<MODE_, WM_, WN_, 16>), grid, block, 0, st, a2);   \
    else hipLaunchKernelGGL((conv8_kernel<MODE_, WM_, WN_, 32>), grid, block, 0, st, a2);            \
  } while (0)
  if (bm == 256) {
    if (mode == MODE_DOWN) C8_GO(MODE_DOWN, 2, 4);
    else if (mode == MODE_UP) C8_GO(MODE_UP, 2, 4);
    else C8_GO(MODE_PLAIN, 2, 4);
  } else {
    if (mode == MODE_DOWN) C8_GO(MODE_DOWN, 4, 2);
    else if (mode == MODE_UP) C8_GO(MODE_UP, 4, 2);
    else C8_GO(MODE_PLAIN, 4, 2);
  }
#undef C8_GO
  return RG_OK;
}
#endif  // RG_CONV8_KERNEL_ONLY
