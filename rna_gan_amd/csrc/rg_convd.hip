// rg_convd.hip -- 4x4 stride-2 conv with 64 input channels and 128 output channels on a 128-pixel-wide input (the
// discriminator's layer 1 at 128 x 128 -> 64 x 64) with the input's PARITY PLANES resident in LDS.
//
// The implicit-GEMM kernel (rg_conv8.hip, 512 x 128 tile) pulls every input pixel of this layer through LDS-DMA four times
// (16 taps / stride^2) and stages 80 KB per 64-deep k-tile for 8.4 MFLOP: the layer runs at the L2 -> LDS rate (95 us at
// batch 64, 0.31-0.33 of the bf16 peak).  The four taps (kh, kw) = (1 - ry + 2 sy, 1 - rx + 2 sx) of one parity
// class (ry, rx) read ONE sub-image -- the input pixels with y = ry, x = rx (mod 2) -- at the four shifts (sy, sx) in {0, 1}^2
// measured in OUTPUT pixels, so here a workgroup loads each plane once per tile:
//
//  * tile = 4 output rows x 64 pixels x all 128 columns; per plane the patch is 5 plane rows x 65 plane pixels of 128 B
//    (41.6 KB; pixels outside the image are zero-filled by the DMA's range check), double-buffered across the four planes of
//    a tile (and across tiles): the shift of a tap is the plain row offset sy * 65 + sx.
//  * K is walked plane by plane, tap by tap: 16 steps per tile of one [128 columns][64 channels] weight slice (16 KB, a
//    ring of 4, DMA'd three steps ahead) = 2 k-steps of 32 channels.  Staged per tile: 166 KB of input + 256 KB of weights for
//    67 MFLOP (159 FLOP per staged byte against 102).
//  * 8 waves = 4 output rows x 2 column halves: a wave owns 64 pixels x 64 columns (16 accumulators of
//    v_mfma_f32_16x16x32_bf16) and issues per step 16 ds_read_b128 + 32 MFMAs.  LDS images are XOR-swizzled per 128-byte row
//    (pixels: segment ^ (row & 6), conflict-free at any shift -- rg_convp.hip; weights: segment ^ X(column), X = bit 1 of
//    the column | bits 4-5 << 1, conflict-free for the column order of the fragments below).
//  * the MFMAs compute the transposed tile (weights as the A operand): a lane's 16 accumulator values are 16 consecutive
//    output channels of one pixel, stored as two 16-byte pieces straight from registers; BatchNorm partial sums (of the
//    bf16-rounded values) by DPP row reductions, one partial row per (tile, output row).
//  * persistent: a workgroup walks consecutive tiles; one s_barrier per step behind a counted vmcnt (never 0 in the loop).
#include "rg_gather.h"
#include <type_traits>

namespace {

template <int V> using icd = std::integral_constant<int, V>;
typedef __attribute__((ext_vector_type(4))) float cd_f32x4;

constexpr int CD_PW = 65, CD_ROWS = 5;
constexpr int CD_PLANE_PX = CD_ROWS * CD_PW;               // 325 plane pixels of 128 B
constexpr int CD_NPI = (CD_PLANE_PX + 7) / 8;              // 41 DMA instructions (8 pixels each) per plane
constexpr int CD_PPW = 6;                                  // ... per wave (8 x 6 = 48 >= 41: the rest go to the dummy KB)
constexpr int CD_PLANE = 42 * 1024;                        // bytes reserved per plane buffer
constexpr int CD_OFF_P0 = 0, CD_OFF_P1 = CD_PLANE, CD_OFF_RING = 2 * CD_PLANE, CD_SLOT = 16384;
constexpr int CD_NSLOT = 4;                                // weight ring: slices are DMA'd three steps ahead
constexpr int CD_OFF_DUMMY = CD_OFF_RING + CD_NSLOT * CD_SLOT;
constexpr int CD_LDS = CD_OFF_DUMMY + 1024;                // 152576 B

// EPI: 0 plain, 1 BatchNorm partial sums
template <int EPI>
__global__ __launch_bounds__(512, 2) void convd_kernel(G2Args a2) {
  constexpr bool HAS_STATS = EPI == 1;
  __shared__ __attribute__((aligned(16))) uint4 lds[CD_LDS / 16];
  const GArgs& g = a2.g;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1;                 // output row of the tile, column half
  const int fr = lane & 15, fh = lane >> 4;
  const int Hs = g.Hs, Ws = g.Ws;                          // input: Hs x 128; output Ho x 64
  const int lgHo = g.lgH;
  const int tiles = g.M >> 8;
  const int per = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T0 = (int)blockIdx.x * per, T1 = min(tiles, T0 + per);
  if (T0 >= T1) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a2.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, a2.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  char* const ldsb = reinterpret_cast<char*>(lds);
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;

  // ---- plane DMA: instruction id (0 .. 40) covers plane pixels rho = 8 id + (lane >> 3), physical 16-byte segment lane & 7
  // holding logical segment (lane & 7) ^ (rho & 6).  Plane pixel (jr, ic) of plane (ry, rx) is input pixel
  // (y, x) = (2 (ho0 + jr) - ry, 2 ic - rx).  Wave w issues ids 6 w + k, k = 0 .. 5 (ids >= 41: all-out-of-range into the dummy KB)
  auto issue_plane = [&](int T, int p, int k) __attribute__((always_inline)) {
    const bool live_t = T < T1;
    const int id = wave * CD_PPW + k;
    const bool live = live_t && id < CD_NPI;
    const int m0 = T << 8;
    const int n = m0 >> (6 + lgHo), ho0 = (m0 >> 6) & ((1 << lgHo) - 1);
    const int ry = p >> 1, rx = p & 1;
    const int rho = id * 8 + (lane >> 3);
    const int jr = rho / CD_PW, ic = rho - jr * CD_PW;
    const int ls = (lane & 7) ^ (rho & 6);
    const int y = 2 * (ho0 + jr) - ry, x = 2 * ic - rx;
    const bool v = live && rho < CD_PLANE_PX && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
    const unsigned off = (unsigned)((((n * Hs + y) * Ws + x) * 64 + ls * 8) * 2);
    const int dst = live ? ((p & 1) ? CD_OFF_P1 : CD_OFF_P0) + id * 1024 : CD_OFF_DUMMY;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr_t)(ldsb + dst), 16, v ? off : OOB, 0, 0, 0);
  };
  // ---- weight slice DMA: [128 columns][64 channels] of tap (kh, kw) into ring slot `slot`; instruction id (0 .. 15) covers
  // columns 8 id + (lane >> 3); wave w issues ids 2 w, 2 w + 1
  auto issue_b = [&](int p, int tp, int slot) __attribute__((always_inline)) {
    const int ry = p >> 1, rx = p & 1, sy = tp >> 1, sx = tp & 1;
    const int tap = ((1 - ry) + 2 * sy) * 4 + (1 - rx) + 2 * sx;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int id = wave * 2 + e;
      const int col = id * 8 + (lane >> 3);
      const int X = ((col >> 1) & 1) | (((col >> 4) & 3) << 1);
      const int ls = (lane & 7) ^ X;
      const unsigned off = (unsigned)((col * g.b_col + tap * g.b_tap + ls * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr_t)(ldsb + CD_OFF_RING + slot * CD_SLOT + id * 1024), 16, off, 0, 0, 0);
    }
  };

  // ---- fragments (16x16x32: lane = (row fr of 16, k-group fh of 8 channels))
  // pixels (MFMA B operand): pixel 16 it + fr of output row wm at shift (sy, sx): rho = (wm + sy) * 65 + 16 it + fr + sx;
  //   the four row tiles are 2048 B apart with the same swizzle
  // weights (MFMA A operand): row i of column tile j is output channel 64 wn + 16 (i >> 2) + 4 j + (i & 3): the four tiles give
  //   a lane 16 consecutive channels of its pixel; tiles are 512 B apart with the same swizzle
  const int bcol0 = wn * 64 + 16 * (fr >> 2) + (fr & 3);
  const int bX = ((fr & 3) >> 1) | ((fr >> 2) << 1);
  cd_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = cd_f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4_t aF[2][4], bF[2][4];

#define CD_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define CD_WAIT8(N, A, B)                                                                                      \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                     \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3])::"memory")
  // fragment reads of k-half kc of a step: 4 pixel tiles (pbase: plane buffer, shift: the tap's row offset) and 4 column
  // tiles (rbase: ring slot)
  auto read_a = [&](int set, unsigned pbase, int shift, int kc) __attribute__((always_inline)) {
    const int rho = wm * CD_PW + fr + shift;
    const unsigned aa = pbase + (unsigned)((rho << 7) + ((((kc << 2) + fh) ^ (rho & 6)) << 4));
    CD_DSR(aF[set][0], aa, 0); CD_DSR(aF[set][1], aa, 2048); CD_DSR(aF[set][2], aa, 4096); CD_DSR(aF[set][3], aa, 6144);
  };
  auto read_b = [&](int set, unsigned rbase, int kc) __attribute__((always_inline)) {
    const unsigned ba = rbase + (unsigned)((bcol0 << 7) + ((((kc << 2) + fh) ^ bX) << 4));
    CD_DSR(bF[set][0], ba, 0); CD_DSR(bF[set][1], ba, 512); CD_DSR(bF[set][2], ba, 1024); CD_DSR(bF[set][3], ba, 1536);
  };
  // 4 MFMAs: column tile j of fragment set `set` against the four pixel tiles
  auto mfma4 = [&](int set, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
      acc[it][j] = rg_mfma_h16_16x16x32(__builtin_bit_cast(h16x8_t, bF[set][j]),
                                                           __builtin_bit_cast(h16x8_t, aF[set][it]), acc[it][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  auto dpp_sum16 = [&](float x) __attribute__((always_inline)) {        // sum over the 16 lanes of a DPP row (every lane gets it)
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
  };
  // ---- epilogue: lane (px = lane & 15, fq = lane >> 4) of tile (it, j) holds channels 64 wn + 16 fq + 4 j .. + 3 of pixel
  // 64 wm + 16 it + px of the tile
  auto epilogue = [&](int T) __attribute__((always_inline)) {
    const int px = lane & 15, fq = lane >> 4;
    const long long mrow = ((long long)T << 8) + wm * 64 + px;
    float s1[16], s2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      u32x4_t o[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const cd_f32x4 v = acc[it][j];
        const uint32_t h0 = f32_to_h16(v[0]), h1 = f32_to_h16(v[1]), h2 = f32_to_h16(v[2]), h3 = f32_to_h16(v[3]);
        o[j >> 1][(j & 1) * 2] = h0 | (h1 << 16);
        o[j >> 1][(j & 1) * 2 + 1] = h2 | (h3 << 16);
        if constexpr (HAS_STATS) {
          const float r0 = h16lo_to_f32(h0), r1 = h16lo_to_f32(h1);
          const float r2 = h16lo_to_f32(h2), r3 = h16lo_to_f32(h3);
          s1[4 * j] += r0; s1[4 * j + 1] += r1; s1[4 * j + 2] += r2; s1[4 * j + 3] += r3;
          s2[4 * j] += r0 * r0; s2[4 * j + 1] += r1 * r1; s2[4 * j + 2] += r2 * r2; s2[4 * j + 3] += r3 * r3;
        }
        acc[it][j] = cd_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.C) + (mrow + it * 16) * g.ldc + wn * 64 + fq * 16);
      dst[0] = __builtin_bit_cast(uint4, o[0]);
      dst[1] = __builtin_bit_cast(uint4, o[1]);
    }
    if constexpr (HAS_STATS) {
      float* row = g.stats + ((size_t)T * 4 + wm) * 2 * g.Ncols + wn * 64 + fq * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) { s1[c] = dpp_sum16(s1[c]); s2[c] = dpp_sum16(s2[c]); }
      if (px == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          reinterpret_cast<float4*>(row)[q] = make_float4(s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]);
          reinterpret_cast<float4*>(row + g.Ncols)[q] = make_float4(s2[4 * q], s2[4 * q + 1], s2[4 * q + 2], s2[4 * q + 3]);
        }
      }
    }
  };

#define CD_SYNC() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  int wslot = 3, rslot = 0;                                // ring slot the next issue_b writes / the current step reads
  // ---- prologue: plane 0 of the first tile, weight slices of steps 0 .. 2; the first k-half's fragments
#pragma unroll
  for (int k = 0; k < CD_PPW; ++k) issue_plane(T0, 0, k);
  issue_b(0, 0, 0);
  issue_b(0, 1, 1);
  issue_b(0, 2, 2);
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(4));
  CD_SYNC();
  read_a(0, lds_base + CD_OFF_P0, 0, 0);
  read_b(0, lds_base + CD_OFF_RING, 0);

  // One step = one tap of one plane (S = 4 p + tp) = two k-halves of 16 MFMAs.  On entry the first half's fragments are in
  // flight into set 0 (read at the end of the previous step, behind that step's barrier).  Everything that is not an MFMA is
  // placed BETWEEN the MFMA groups: the second half's reads and this step's DMA issue under the first half, the barrier
  // that publishes the next step's data and the next step's first reads in the middle of the second half -- the two waves
  // of a SIMD run the same program in step, so nothing else would fill the matrix pipe while they load.
  // Register sets: a set is re-read only after every MFMA group issued since its last use has been followed by another
  // group (the matrix pipe is in order: those MFMAs are done).
  auto step = [&](auto S, int T) __attribute__((always_inline)) {
    constexpr int s = decltype(S)::value, p = s >> 2, tp = s & 3;
    constexpr int s2 = (s + 3) & 15;                       // the step whose weight slice is issued now
    constexpr int sn = (s + 1) & 15, pn = sn >> 2, tpn = sn & 3;
    const unsigned pbase = lds_base + ((p & 1) ? CD_OFF_P1 : CD_OFF_P0);
    const unsigned rbase = lds_base + CD_OFF_RING + rslot * CD_SLOT;
    rslot = rslot == CD_NSLOT - 1 ? 0 : rslot + 1;
    constexpr int shift = (tp >> 1) * CD_PW + (tp & 1);
    CD_WAIT8(0, aF[0], bF[0]);
    mfma4(0, 0);
    read_a(1, pbase, shift, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(0, 1);
    read_b(1, rbase, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(0, 2);
    if constexpr (tp < 2) {                                // the next plane (of the next tile behind plane 3) into the other buffer
      issue_plane(p == 3 ? T + 1 : T, (p + 1) & 3, 3 * tp);
      issue_plane(p == 3 ? T + 1 : T, (p + 1) & 3, 3 * tp + 1);
      issue_plane(p == 3 ? T + 1 : T, (p + 1) & 3, 3 * tp + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma4(0, 3);
    issue_b(s2 >> 2, s2 & 3, wslot);                       // (its slot was read last in step s - 1: behind that step's barrier)
    wslot = wslot == CD_NSLOT - 1 ? 0 : wslot + 1;
    __builtin_amdgcn_sched_barrier(0);
    CD_WAIT8(0, aF[1], bF[1]);
    mfma4(1, 0);
    mfma4(1, 1);
    // the next step's data: everything this wave issued up to two steps ago has landed -- the next step's weight slice and, in
    // front of a plane's first step, the plane -- (outstanding: the previous and this step's instructions; a step issues 3
    // plane + 2 weight instructions at tp = 0, 1 and 2 weight instructions at tp = 2, 3), and behind the barrier everybody else's
    // (the epilogue's stores in front of a tile's step 0 are older than everything counted here: a stronger wait, never a
    // weaker one.  Leaving them outstanding over steps 0 and 1 -- what those steps need is issued before the stores -- measured
    // the same.)
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(tp == 0 ? 7 : tp == 1 ? 10 : tp == 2 ? 7 : 4));
    CD_SYNC();
    constexpr int shiftn = (tpn >> 1) * CD_PW + (tpn & 1);
    read_a(0, lds_base + ((pn & 1) ? CD_OFF_P1 : CD_OFF_P0), shiftn, 0);
    read_b(0, lds_base + CD_OFF_RING + rslot * CD_SLOT, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(1, 2);
    mfma4(1, 3);
  };

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    step(icd<0>{}, T);  step(icd<1>{}, T);  step(icd<2>{}, T);  step(icd<3>{}, T);
    step(icd<4>{}, T);  step(icd<5>{}, T);  step(icd<6>{}, T);  step(icd<7>{}, T);
    step(icd<8>{}, T);  step(icd<9>{}, T);  step(icd<10>{}, T); step(icd<11>{}, T);
    step(icd<12>{}, T); step(icd<13>{}, T); step(icd<14>{}, T); step(icd<15>{}, T);
    epilogue(T);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CD_SYNC
#undef CD_WAIT8
#undef CD_DSR
}

}  // namespace

// shapes convd_kernel takes: 64 input channels, 128 output channels, a 128-pixel-wide input whose output height is a power
// of two >= 4, whole 256-pixel tiles
bool rg_convd_supported(int M, int Ncols, int Cin, int Hs, int Ws) {
  return Ncols == 128 && Cin == 64 && Ws == 128 && Hs >= 8 && rg_is_pow2(Hs) && M % 256 == 0 && M >= 256;
}

int rg_convd_stats_rows(int M) { return M / 64; }

int rg_convd_launch(const void* args, hipStream_t st) {
  const G2Args& a2 = *reinterpret_cast<const G2Args*>(args);
  const int tiles = a2.g.M / 256;
  const int target = rg_option("convd_blocks", 256);
  int grid = tiles < target ? tiles : target;
  const int per = (tiles + grid - 1) / grid;
  grid = (tiles + per - 1) / per;
  if (a2.g.stats) hipLaunchKernelGGL((convd_kernel<1>), dim3((unsigned)grid), dim3(512), 0, st, a2);
  else hipLaunchKernelGGL((convd_kernel<0>), dim3((unsigned)grid), dim3(512), 0, st, a2);
  return RG_OK;
}
