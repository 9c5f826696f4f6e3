// rg_mfma.hip -- bf16 MFMA implicit-GEMM kernels for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 acc).
//
//  * gather_gemm_kernel<MODE,EPI>: C[M][Ncols] = sum_k A(m,k) * Bt[col][k]
//      MODE_DOWN : stride-2 4x4 conv      (A rows gathered from the high-res NHWC tensor, 16 taps)
//      MODE_UP   : its transpose          (4 output-parity classes, 2x2 taps each, grid.y = class)
//      MODE_PLAIN: ordinary row-major A   (generator layer 0, betaVAE Linear layers)
//    128x128 block tile, BK = 64 bf16 (one 128-byte line per row per k-tile), 4 waves as 2x2 with a
//    64x64 wave tile (2x2 MFMA 32x32x16).  Operands are staged global -> registers -> LDS
//    (issue-early / write-late, double-buffered, ONE barrier per k-tile); the LDS image is
//    [row][8 chunks of 16 B] with chunk ^= (row>>1)&7, which makes both the ds_write_b128 staging
//    stores and the ds_read_b128 fragment reads bank-conflict free (MI355X_MICROARCH LDS table).
//    The epilogue transposes the accumulators through LDS so every global store is 16 bytes wide
//    and row-contiguous.
//  * wgrad_kernel: dW[o][tap][i] = sum_pix low[pix][o] * high[src(pix,tap)][i].  The contraction
//    index (pixels) is the slow axis of both NHWC operands, so the [pixel][channel] LDS images are
//    read with ds_read_b64_tr_b16 (hardware transpose) to build k-contiguous MFMA fragments.
//    Split-K over pixels into fp32 slabs, reduced in fixed order (deterministic).
#include "rg_gather.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ uint4 ld16_if(const uint16_t* p, bool pred) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (pred) v = *reinterpret_cast<const uint4*>(p);
  return v;
}

__device__ __forceinline__ int lds_chunk_index(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

template <int MODE, int EPI>
__global__ __launch_bounds__(256, 2) void gather_gemm_kernel(GArgs g) {
  // 2 stages x (A,B) x 128 rows x 8 chunks x 16 B = 64 KB; reused as the fp32 C tile (128x128x4 B)
  __shared__ __attribute__((aligned(16))) uint4 lds[2 * 2 * 1024];

  const int t = threadIdx.x;
  const int tile_m = blockIdx.x / g.tiles_n, tile_n = blockIdx.x - tile_m * g.tiles_n;
  const int bm = tile_m * 128, bn = tile_n * 128;
  const int par = (MODE == MODE_UP) ? (int)blockIdx.y : 0;
  const int ph = par >> 1, pw = par & 1;

  // ---- per-thread staging assignment: chunk (16 B) of rows r0 + 32*j
  const int chunk = t & 7, r0 = t >> 3;
  const int Wq = 1 << g.lgW, Hq = 1 << g.lgH;
  const uint16_t *a_ptr0, *a_ptr1, *a_ptr2, *a_ptr3, *b_ptr0, *b_ptr1, *b_ptr2, *b_ptr3;
  unsigned a_mask0, a_mask1, a_mask2, a_mask3;
  bool b_ok0, b_ok1, b_ok2, b_ok3;
#define RG_ROW_SETUP(J, APTR, AMASK, BPTR, BOK)                                                   \
  do {                                                                                            \
    int m = bm + r0 + 32 * (J);                                                                   \
    bool ok = m < g.M;                                                                            \
    int mm = ok ? m : 0;                                                                          \
    int wq = mm & (Wq - 1), hq = (mm >> g.lgW) & (Hq - 1), n = mm >> (g.lgW + g.lgH);             \
    unsigned mask = 0;                                                                            \
    long long base;                                                                               \
    if (MODE == MODE_DOWN) {                                                                      \
      int hs0 = 2 * hq - 1, ws0 = 2 * wq - 1;                                                     \
      base = (((long long)n * g.Hs + hs0) * g.Ws + ws0) * g.Cin;                                  \
      _Pragma("unroll") for (int kh = 0; kh < 4; ++kh)                                            \
      _Pragma("unroll") for (int kw = 0; kw < 4; ++kw) {                                          \
        bool v = (unsigned)(hs0 + kh) < (unsigned)g.Hs && (unsigned)(ws0 + kw) < (unsigned)g.Ws;  \
        mask |= (v ? 1u : 0u) << (kh * 4 + kw);                                                   \
      }                                                                                           \
    } else if (MODE == MODE_UP) {                                                                 \
      base = (((long long)n * g.Hs + hq) * g.Ws + wq) * g.Cin;                                    \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                               \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                             \
        int kh, kw, dh, dw;                                                                       \
        up_tap_dev(ph, a, kh, dh);                                                                \
        up_tap_dev(pw, b, kw, dw);                                                                \
        bool v = (unsigned)(hq + dh) < (unsigned)g.Hs && (unsigned)(wq + dw) < (unsigned)g.Ws;    \
        mask |= (v ? 1u : 0u) << (a * 2 + b);                                                     \
      }                                                                                           \
    } else {                                                                                      \
      base = (long long)mm * g.Cin;                                                               \
      mask = 1u;                                                                                  \
    }                                                                                             \
    APTR = g.A + base + chunk * 8;                                                                \
    AMASK = ok ? mask : 0u;                                                                       \
    int col = bn + r0 + 32 * (J);                                                                 \
    BOK = col < g.Ncols;                                                                          \
    BPTR = g.B + (long long)(BOK ? col : 0) * g.b_col + chunk * 8;                             \
  } while (0)
  RG_ROW_SETUP(0, a_ptr0, a_mask0, b_ptr0, b_ok0);
  RG_ROW_SETUP(1, a_ptr1, a_mask1, b_ptr1, b_ok1);
  RG_ROW_SETUP(2, a_ptr2, a_mask2, b_ptr2, b_ok2);
  RG_ROW_SETUP(3, a_ptr3, a_mask3, b_ptr3, b_ok3);

  const int cpt = g.Cin >> 6;          // k-tiles per tap
  const int nkt = g.taps * cpt;

  // staging registers are named scalars (not arrays captured by a lambda): keeps them out of scratch
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define RG_LOAD_TILE(TAP, C0)                                                                     \
  do {                                                                                            \
    const int tap_ = (TAP), c0_ = (C0);                                                           \
    int a_delta, b_tap;                                                                           \
    if (MODE == MODE_DOWN) {                                                                      \
      a_delta = ((tap_ >> 2) * g.Ws + (tap_ & 3)) * g.Cin;                                        \
      b_tap = tap_;                                                                               \
    } else if (MODE == MODE_UP) {                                                                 \
      int kh, kw, dh, dw;                                                                         \
      up_tap_dev(ph, tap_ >> 1, kh, dh);                                                          \
      up_tap_dev(pw, tap_ & 1, kw, dw);                                                           \
      a_delta = (dh * g.Ws + dw) * g.Cin;                                                         \
      b_tap = kh * 4 + kw;                                                                        \
    } else {                                                                                      \
      a_delta = 0;                                                                                \
      b_tap = 0;                                                                                  \
    }                                                                                             \
    const int ao = a_delta + c0_, bo = b_tap * g.b_tap + c0_;                                       \
    ra0 = ld16_if(a_ptr0 + ao, (a_mask0 >> tap_) & 1u);                                           \
    ra1 = ld16_if(a_ptr1 + ao, (a_mask1 >> tap_) & 1u);                                           \
    ra2 = ld16_if(a_ptr2 + ao, (a_mask2 >> tap_) & 1u);                                           \
    ra3 = ld16_if(a_ptr3 + ao, (a_mask3 >> tap_) & 1u);                                           \
    rb0 = ld16_if(b_ptr0 + bo, b_ok0);                                                            \
    rb1 = ld16_if(b_ptr1 + bo, b_ok1);                                                            \
    rb2 = ld16_if(b_ptr2 + bo, b_ok2);                                                            \
    rb3 = ld16_if(b_ptr3 + bo, b_ok3);                                                            \
  } while (0)
#define RG_STORE_TILE(STAGE)                                                                      \
  do {                                                                                            \
    uint4* sa_ = lds + (STAGE) * 2048;                                                            \
    uint4* sb_ = sa_ + 1024;                                                                      \
    sa_[lds_chunk_index(r0, chunk)] = ra0;       sb_[lds_chunk_index(r0, chunk)] = rb0;           \
    sa_[lds_chunk_index(r0 + 32, chunk)] = ra1;  sb_[lds_chunk_index(r0 + 32, chunk)] = rb1;      \
    sa_[lds_chunk_index(r0 + 64, chunk)] = ra2;  sb_[lds_chunk_index(r0 + 64, chunk)] = rb2;      \
    sa_[lds_chunk_index(r0 + 96, chunk)] = ra3;  sb_[lds_chunk_index(r0 + 96, chunk)] = rb3;      \
  } while (0)

  // ---- MFMA assignment
  const int wave = t >> 6, lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int tap_n = 0, cc_n = 0;   // coordinates of the NEXT tile to load
  RG_LOAD_TILE(0, 0);
  RG_STORE_TILE(0);
  if (++cc_n == cpt) { cc_n = 0; ++tap_n; }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) {
      RG_LOAD_TILE(tap_n, cc_n << 6);
      if (++cc_n == cpt) { cc_n = 0; ++tap_n; }
    }
    const uint4* sa = lds + cur * 2048;
    const uint4* sb = sa + 1024;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = 2 * kk + fh;
      h16x8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = wm * 64 + i * 32 + fr;
        uint4 v = sa[lds_chunk_index(row, ch)];
        fa[i] = __builtin_bit_cast(h16x8_t, v);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = wn * 64 + j * 32 + fr;
        uint4 v = sb[lds_chunk_index(row, ch)];
        fb[j] = __builtin_bit_cast(h16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = rg_mfma_h16_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) RG_STORE_TILE(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#undef RG_LOAD_TILE
#undef RG_STORE_TILE
#undef RG_ROW_SETUP

  // ---- epilogue: accumulators -> LDS (fp32 [128][128]) -> 16-byte row-contiguous global stores
  float* cs = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        int col = wn * 64 + j * 32 + fr;
        cs[row * 128 + col] = acc[i][j][r];
      }
  __syncthreads();
  const int cg = t & 15, rr = t >> 4;
  const int col = bn + cg * 8;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int row = rr + 16 * p;
    int m = bm + row;
    if (m >= g.M || col >= g.Ncols) continue;
    float4 v0 = *reinterpret_cast<const float4*>(cs + row * 128 + cg * 8);
    float4 v1 = *reinterpret_cast<const float4*>(cs + row * 128 + cg * 8 + 4);
    long long orow;
    if (MODE == MODE_UP) {
      int wq = m & (Wq - 1), hq = (m >> g.lgW) & (Hq - 1), n = m >> (g.lgW + g.lgH);
      orow = ((long long)n * (2 * Hq) + 2 * hq + ph) * (2 * Wq) + 2 * wq + pw;
    } else {
      orow = m;
    }
    if (EPI == EPI_BF16) {
      if (g.mask) {
        const uint4 a = *reinterpret_cast<const uint4*>(g.mask + orow * g.ldc + col);
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
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.C) + orow * g.ldc + col) = o;
    } else {
      float vals[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      float* yo = reinterpret_cast<float*>(g.C) + orow * g.ldc + col;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = vals[e];
        if (g.scale) v *= g.scale[col + e];
        if (g.shift) v += g.shift[col + e];
        yo[e] = lrelu_f(v, g.slope);
      }
    }
  }
}

// ================================================================================================
// gather-GEMM v2: LDS-DMA staging (buffer_load_dwordx4 ... lds)
// ================================================================================================
// Same tile math as gather_gemm_kernel, but operands go global -> LDS directly: no staging VGPRs, no
// ds_write_b128 (the slowest LDS instruction: ~79 B/clk/CU, MI355X_MICROARCH LDS table), and the buffer
// descriptor's range check returns zeros for out-of-image taps / rows beyond M (voffset >= num_records).
// One wave-instruction writes 64 consecutive 16-byte LDS slots (8 rows x 8 chunks); the XOR swizzle is
// applied to the SOURCE chunk each lane fetches (rule 21 of the CDNA guide: linear destination,
// swizzled source, swizzled read).  2 stages, tile k+1 in flight during the MFMAs of tile k, one
// barrier per k-tile.  Tile = BM x BN with 64x64 wave tiles: 128x128 (2x2 waves) or 256x64 (4x1 waves,
// for 64-channel outputs).  SPLITK: blockIdx.z owns a k-tile range and writes fp32 partials.
// NP > 0: fp32 operands as K-concatenated bf16 planes (rg_conv8f.hip has the scheme; flat k-tile -> plane pair fastest), used
// with EPI_LINEAR (fp32 result) for the 64-column transposed conv of the fp32 mode, which the 8-wave kernel's tiles do not cover.
// NJ: live 32-column sub-tiles of a wave's 64 columns.  1 = outputs of <= 32 columns (the resize-convolution generator's image layer:
// 3 channels in an 8-column row): the second sub-tile's B fragments are not read and its MFMAs not issued (they would multiply zeros).
template <int MODE, int EPI, int BM, int BN, int NSTAGE, int NT, int WTM = 64, int NP = 0, int NJ = 2>
__global__ __launch_bounds__(NT, (NSTAGE * (BM + BN) * 128 <= 80 * 1024 && NT == 256) ? 2 : 1) void gather_gemm_dma_kernel(G2Args a2) {
  static_assert(NJ == 2 || (NJ == 1 && BN == 64), "NJ = 1: one wave column, first 32 columns live");
  constexpr int WN = BN / 64;                                // waves along N (WTM x 64 wave tiles)
  constexpr int TI = WTM / 32;                               // 32-row MFMA sub-tiles per wave (2 or 4)
  constexpr int RPI = NT / 8;                                // rows covered by one block-wide load instruction
  constexpr int A_SLOTS = BM * 8, B_SLOTS = BN * 8;          // 16-byte slots per stage
  constexpr int STAGE_SLOTS = A_SLOTS + B_SLOTS;
  constexpr int A_LD = BM / RPI, B_LD = BN / RPI;            // wave-instructions per thread per k-tile
  static_assert((BM / WTM) * WN * 64 == NT, "one WTM x 64 wave tile per wave");
  static_assert(TI == 2 || TI == 4, "wave tile is 64x64 or 128x64");
  // the epilogue goes through LDS in column slices of EP_COLS when the whole fp32 tile does not fit
  constexpr int EP_COLS = (BM * BN * 4 <= 128 * 1024) ? BN : BN / 2;
  constexpr int LDS_SLOTS = (NSTAGE * STAGE_SLOTS * 16 > BM * EP_COLS * 4) ? NSTAGE * STAGE_SLOTS : BM * EP_COLS / 4;
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_SLOTS];
  const GArgs& g = a2.g;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // consecutive workgroups go to consecutive XCDs; giving XCD x the x-th contiguous eighth of the tile list keeps
  // the A rows that neighbouring column tiles (and neighbouring output rows) share inside one XCD's L2
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

  // lane -> (row within an 8-row group, physical chunk); logical chunk = pc ^ ((row>>1)&7) = pc ^ ((t>>4)&7)
  const int pc = t & 7, r0 = t >> 3;
  const int lc = pc ^ ((t >> 4) & 7);
  const int Wq = 1 << g.lgW, Hq = 1 << g.lgH;
  int a_off[A_LD];          // byte offset of the row's base pixel + this lane's chunk
  unsigned a_mask[A_LD];
  int b_off[B_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    int m = bm + r0 + RPI * j;
    bool ok = m < g.M;
    int mm = ok ? m : 0;
    int wq = mm & (Wq - 1), hq = (mm >> g.lgW) & (Hq - 1), n = mm >> (g.lgW + g.lgH);
    unsigned mask = 0;
    long long base;
    if (MODE == MODE_DOWN) {
      int hs0 = 2 * hq - 1, ws0 = 2 * wq - 1;
      base = (((long long)n * g.Hs + hs0) * g.Ws + ws0) * g.Cin;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          bool v = (unsigned)(hs0 + kh) < (unsigned)g.Hs && (unsigned)(ws0 + kw) < (unsigned)g.Ws;
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
          bool v = (unsigned)(hq + dh) < (unsigned)g.Hs && (unsigned)(wq + dw) < (unsigned)g.Ws;
          mask |= (v ? 1u : 0u) << (a * 2 + b);
        }
    } else if (MODE == MODE_C3) {
      base = (((long long)n * g.Hs + hq) * g.Ws + wq) * g.Cin;     // top-left tap of the 3x3 window; all 9 exist
      mask = 0x1FFu;
    } else if (MODE == MODE_C3T) {
      const int Wp = g.Ws + 2, Hp = g.Hs + 2;
      const int jj = mm % Wp, tt = mm / Wp, ii = tt % Hp, nn = tt / Hp;
      base = (((long long)nn * g.Hs + ii) * g.Ws + jj) * g.Cin;    // tap (kh, kw) reads gy[ii - kh][jj - kw]
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          bool v = (unsigned)(ii - kh) < (unsigned)g.Hs && (unsigned)(jj - kw) < (unsigned)g.Ws;
          mask |= (v ? 1u : 0u) << (kh * 3 + kw);
        }
    } else {
      base = (long long)mm * g.Cin;
      mask = 1u;
    }
    a_off[j] = (int)((base + lc * 8) * 2);
    a_mask[j] = ok ? mask : 0u;
  }
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    int col = bn + r0 + RPI * j;
    b_off[j] = col < g.Ncols ? (int)(((long long)col * g.b_col + lc * 8) * 2) : -1;
  }

  const int cpt = g.Cin >> 6;
  const int nkt_all = g.taps * cpt * (NP > 0 ? NP : 1);
  const int per = (nkt_all + a2.nsplit - 1) / a2.nsplit;
  const int kt_begin = zs * per;
  const int kt_end = min(nkt_all, kt_begin + per);
  const int nkt = kt_end - kt_begin;

  const bool korder = a2.korder && (MODE == MODE_DOWN || MODE == MODE_UP);
  auto issue = [&](int stage, int tap, int c0, int pp) {
    int a_delta, b_tap;
    if (MODE == MODE_DOWN && korder) tap = rg_down_tap(tap);
    if (MODE == MODE_DOWN) {
      a_delta = ((tap >> 2) * g.Ws + (tap & 3)) * g.Cin;
      b_tap = tap;
    } else if (MODE == MODE_UP) {
      int kh, kw, dh, dw;
      up_tap_dev(ph, tap >> 1, kh, dh);
      up_tap_dev(pw, tap & 1, kw, dw);
      a_delta = (dh * g.Ws + dw) * g.Cin;
      b_tap = kh * 4 + kw;
    } else if (MODE == MODE_C3) {
      const int kh = tap / 3;
      a_delta = (kh * g.Ws + (tap - 3 * kh)) * g.Cin;
      b_tap = tap;
    } else if (MODE == MODE_C3T) {
      const int kh = tap / 3;
      a_delta = -(kh * g.Ws + (tap - 3 * kh)) * g.Cin;
      b_tap = tap;
    } else {
      a_delta = 0;
      b_tap = 0;
    }
    int ao = (a_delta + c0) * 2, bo = (b_tap * g.b_tap + c0) * 2;
    if constexpr (NP > 0) {              // plane pair pp: hh hm mh hl lh mm
      ao += ((0x120100 >> (4 * pp)) & 15) * (int)a2.a_plane;
      bo += ((0x102010 >> (4 * pp)) & 15) * (int)a2.b_plane;
    }
    uint4* sbase = lds + stage * STAGE_SLOTS;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      unsigned vo = ((a_mask[j] >> tap) & 1u) ? (unsigned)(a_off[j] + ao) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr_t)(sbase + j * NT + wave * 64), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      unsigned vo = b_off[j] >= 0 ? (unsigned)(b_off[j] + bo) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr_t)(sbase + A_SLOTS + j * NT + wave * 64), 16, vo, 0, 0,
                                               0);
    }
  };

  const int lane = t & 63;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16_t acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int pp_n = 0, q_begin = kt_begin;
  if constexpr (NP > 0) { pp_n = kt_begin % NP; q_begin = kt_begin / NP; }
  int tap_n = q_begin / cpt, cc_n = q_begin - tap_n * cpt;
  if (korder) { cc_n = q_begin / g.taps; tap_n = q_begin - cc_n * g.taps; }
  auto advance = [&]() {
    if constexpr (NP > 0) {
      if (++pp_n < NP) return;
      pp_n = 0;
    }
    if (korder) { if (++tap_n == g.taps) { tap_n = 0; ++cc_n; } }
    else if (++cc_n == cpt) { cc_n = 0; ++tap_n; }
  };
  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p)
    if (p < nkt) {
      issue(p, tap_n, cc_n << 6, pp_n);
      advance();
    }
  int st_c = 0, st_i = NSTAGE - 1;     // stage being computed / stage to issue into
  // Fragment reads are asm ds_read_b128 with hand-counted lgkmcnt waits.  (1) hipcc puts s_waitcnt vmcnt(0) in
  // front of every compiler-visible ds_read that follows an outstanding LDS-DMA (it assumes they alias), which
  // drains the prefetch queue each k-tile; (2) fragments are double-buffered per 16-wide k-step: the reads of
  // step kk+1 are in flight during the MFMAs of step kk (LDS returns in order, so lgkmcnt(TI+2) means "step kk
  // landed" even with scalar loads outstanding).
  // LDS bandwidth is what bounds this kernel: a WTM x 64 wave tile reads (WTM+64)*32 B per k-step for
  // WTM*64*32 flops, i.e. 32 flop/B at 64x64 (= the full 128 B/clk/CU LDS rate at MFMA peak) and 42.7 at 128x64.
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;
  const int rowA = wm * WTM + fr, rowB = wn * 64 + fr;
  const int c0a = fh ^ ((rowA >> 1) & 7), c0b = fh ^ ((rowB >> 1) & 7);   // chunk(kk) = c0 ^ 2kk; +32 rows keeps the swizzle
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    fa[kk] = lds_base + 16u * (unsigned)(rowA * 8 + (c0a ^ (2 * kk)));
    fb[kk] = lds_base + 16u * (unsigned)(A_SLOTS + rowB * 8 + (c0b ^ (2 * kk)));
  }
#define RG_DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr) : "memory")
#define RG_READ(buf, kk)                                  \
  do {                                                    \
    RG_DSR(qa[buf][0], fas[kk], 0);                       \
    RG_DSR(qb[buf][0], fbs[kk], 0);                       \
    RG_DSR(qa[buf][1], fas[kk], 4096);                    \
    if constexpr (NJ == 2) RG_DSR(qb[buf][1], fbs[kk], 4096); \
    if constexpr (TI == 4) {                              \
      RG_DSR(qa[buf][2], fas[kk], 8192);                  \
      RG_DSR(qa[buf][3], fas[kk], 12288);                 \
    }                                                     \
  } while (0)
// (cnt4 / cnt6: reads of the NEXT step that may stay in flight with TI = 2 / TI = 4 and both column sub-tiles; with NJ = 1 a step
// has one read fewer, so "next step in flight" is 3 / 5; 0 = everything landed)
#define RG_WAIT(buf, cnt4, cnt6)                                                                                    \
  do {                                                                                                              \
    if constexpr (NJ == 1 && TI == 2) {                                                                             \
      if constexpr (cnt4 == 0)                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qa[buf][0]), "+v"(qa[buf][1]), "+v"(qb[buf][0])::"memory");      \
      else                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(qa[buf][0]), "+v"(qa[buf][1]), "+v"(qb[buf][0])::"memory");      \
    } else if constexpr (NJ == 1) {                                                                                 \
      if constexpr (cnt6 == 0)                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                         \
                     : "+v"(qa[buf][0]), "+v"(qa[buf][1]), "+v"(qa[buf][2]), "+v"(qa[buf][3]), "+v"(qb[buf][0])::"memory"); \
      else                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(5)"                                                                         \
                     : "+v"(qa[buf][0]), "+v"(qa[buf][1]), "+v"(qa[buf][2]), "+v"(qa[buf][3]), "+v"(qb[buf][0])::"memory"); \
    } else if constexpr (TI == 4)                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(" #cnt6 ")"                                                                   \
                   : "+v"(qa[buf][0]), "+v"(qa[buf][1]), "+v"(qa[buf][2]), "+v"(qa[buf][3]), "+v"(qb[buf][0]),      \
                     "+v"(qb[buf][1])::"memory");                                                                   \
    else                                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(" #cnt4 ")"                                                                   \
                   : "+v"(qa[buf][0]), "+v"(qa[buf][1]), "+v"(qb[buf][0]), "+v"(qb[buf][1])::"memory");             \
  } while (0)
#define RG_MFMAS(buf)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] =        \
      rg_mfma_h16_32x32x16(__builtin_bit_cast(h16x8_t, qa[buf][i]),                            \
                                              __builtin_bit_cast(h16x8_t, qb[buf][j]), acc[i][j], 0, 0, 0)
  constexpr int WAIT_A = vmcnt_imm((A_LD + B_LD) * (NSTAGE > 2 ? NSTAGE - 2 : 0));
  constexpr int WAIT_B = vmcnt_imm((A_LD + B_LD) * (NSTAGE > 3 ? NSTAGE - 3 : 0));
  for (int kt = 0; kt < nkt; ++kt) {
    // counted wait: only tile kt has to be here, the NSTAGE-2 younger tiles stay in flight across the barrier
    if (NSTAGE > 2 && kt + NSTAGE - 2 < nkt) __builtin_amdgcn_s_waitcnt(WAIT_A);
    else if (NSTAGE > 3 && kt + NSTAGE - 3 < nkt) __builtin_amdgcn_s_waitcnt(WAIT_B);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // everyone's part of tile kt landed; everyone finished reading tile kt-1
    if (kt + NSTAGE - 1 < nkt) {       // refill the stage tile kt-1 lived in
      issue(st_i, tap_n, cc_n << 6, pp_n);
      advance();
    }
    const unsigned so = (unsigned)(st_c * STAGE_SLOTS * 16);
    unsigned fas[4], fbs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { fas[kk] = fa[kk] + so; fbs[kk] = fb[kk] + so; }
    u32x4_t qa[2][TI], qb[2][2];
    RG_READ(0, 0);
    RG_READ(1, 1);
    __builtin_amdgcn_s_setprio(1);
    RG_WAIT(0, 4, 6);
    RG_MFMAS(0);
    RG_READ(0, 2);
    RG_WAIT(1, 4, 6);
    RG_MFMAS(1);
    RG_READ(1, 3);
    RG_WAIT(0, 4, 6);
    RG_MFMAS(0);
    RG_WAIT(1, 0, 0);
    RG_MFMAS(1);
    __builtin_amdgcn_s_setprio(0);
    st_c = st_c + 1 == NSTAGE ? 0 : st_c + 1;
    st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
  }
#undef RG_DSR
#undef RG_READ
#undef RG_WAIT
#undef RG_MFMAS
  __syncthreads();

  // ---- BatchNorm statistics straight from the accumulators: a lane's 16*TI registers of column tile j all belong to
  // ONE output column (col = j*32 + lane%32, rows spread over the registers and the two half-waves), so the column
  // sums of this wave's WTM rows are register adds plus one cross-half shuffle.  Values are rounded to bf16 first (the
  // statistics of what is stored); rows beyond M were zero-filled operands and contribute exactly 0.
  if (EPI == EPI_BF16 && g.stats && a2.nsplit == 1) {
    constexpr int PARTS = BM / WTM;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = h16_to_f32(f32_to_h16(acc[i][j][r]));
          s1 += v; s2 += v * v;
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const int gcol = bn + wn * 64 + j * 32 + fr;
      if (fh == 0 && gcol < g.Ncols) {
        const size_t grow = ((size_t)par * a2.tiles_m + tile_m) * PARTS + wm;
        g.stats[(grow * 2 + 0) * g.Ncols + gcol] = s1;
        g.stats[(grow * 2 + 1) * g.Ncols + gcol] = s2;
      }
    }
  }

  // ---- epilogue through LDS (fp32 [BM][EP_COLS] per pass)
  float* cs = reinterpret_cast<float*>(lds);
  constexpr int CG = EP_COLS / 8, RPP = NT / CG;      // column groups per row, rows per pass
  const int cg = t % CG, rr = t / CG;
#pragma unroll 1
  for (int ep = 0; ep < BN / EP_COLS; ++ep) {
    if (ep) __syncthreads();
    if ((wn * 64) / EP_COLS == ep) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            int col = wn * 64 - ep * EP_COLS + j * 32 + fr;
            cs[row * EP_COLS + col] = acc[i][j][r];
          }
    }
    __syncthreads();
    const int col = bn + ep * EP_COLS + cg * 8;
#pragma unroll 4
    for (int p = 0; p < BM / RPP; ++p) {
      int row = rr + RPP * p;
      int m = bm + row;
      if (m >= g.M || col >= g.Ncols) continue;
      float4 v0 = *reinterpret_cast<const float4*>(cs + row * EP_COLS + cg * 8);
      float4 v1 = *reinterpret_cast<const float4*>(cs + row * EP_COLS + cg * 8 + 4);
      long long orow;
      if (MODE == MODE_UP) {
        int wq = m & (Wq - 1), hq = (m >> g.lgW) & (Hq - 1), n = m >> (g.lgW + g.lgH);
        orow = ((long long)n * (2 * Hq) + 2 * hq + ph) * (2 * Wq) + 2 * wq + pw;
      } else {
        orow = m;
      }
      if (MODE == MODE_C3 && EPI == EPI_BF16 && g.shift && zs == 0) {      // Conv2d bias of the resize-convolution block (first split only)
        const float* b = g.shift + col;
        v0.x += b[0]; v0.y += b[1]; v0.z += b[2]; v0.w += b[3];
        v1.x += b[4]; v1.y += b[5]; v1.z += b[6]; v1.w += b[7];
      }
      if (a2.nsplit > 1) {
        float* so = a2.slab + (long long)zs * a2.slab_stride + orow * g.ldc + col;
        *reinterpret_cast<float4*>(so) = v0;
        *reinterpret_cast<float4*>(so + 4) = v1;
      } else if (EPI == EPI_BF16) {
        if (g.mask) {
          const uint4 a = *reinterpret_cast<const uint4*>(g.mask + orow * g.ldc + col);
          v0.x *= rg_lmask(a.x, g.mslope); v0.y *= rg_lmask(a.x >> 16, g.mslope);
          v0.z *= rg_lmask(a.y, g.mslope); v0.w *= rg_lmask(a.y >> 16, g.mslope);
          v1.x *= rg_lmask(a.z, g.mslope); v1.y *= rg_lmask(a.z >> 16, g.mslope);
          v1.z *= rg_lmask(a.w, g.mslope); v1.w *= rg_lmask(a.w >> 16, g.mslope);
        }
        if (g.affine) rg_affine8(v0, v1, g.scale + col, g.shift + col, g.slope);
        uint4 o;
        o.x = (uint32_t)f32_to_h16(v0.x) | ((uint32_t)f32_to_h16(v0.y) << 16);
        o.y = (uint32_t)f32_to_h16(v0.z) | ((uint32_t)f32_to_h16(v0.w) << 16);
        o.z = (uint32_t)f32_to_h16(v1.x) | ((uint32_t)f32_to_h16(v1.y) << 16);
        o.w = (uint32_t)f32_to_h16(v1.z) | ((uint32_t)f32_to_h16(v1.w) << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.C) + orow * g.ldc + col) = o;
      } else {
        float vals[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        float* yo = reinterpret_cast<float*>(g.C) + orow * g.ldc + col;
        if (g.maskf) {                                         // LeakyReLU backward of the consumer (fp32 activation as the mask)
          const float4 m0 = *reinterpret_cast<const float4*>(g.maskf + orow * g.ldc + col);
          const float4 m1 = *reinterpret_cast<const float4*>(g.maskf + orow * g.ldc + col + 4);
          const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) vals[e] *= mm[e] > 0.f ? 1.f : g.mslope;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = vals[e];
          if (g.scale && col + e < g.Ncols) v *= g.scale[col + e];
          if (g.shift && col + e < g.Ncols) v += g.shift[col + e];
          vals[e] = lrelu_f(v, g.slope);
        }
        if ((g.ldc & 3) == 0 && col + 8 <= g.ldc) {          // rows 16-byte aligned: two 16-byte stores
          *reinterpret_cast<float4*>(yo) = make_float4(vals[0], vals[1], vals[2], vals[3]);
          *reinterpret_cast<float4*>(yo + 4) = make_float4(vals[4], vals[5], vals[6], vals[7]);
        } else {                                             // ragged row end (Ncols % 8 != 0) / unaligned rows
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < g.Ncols) yo[e] = vals[e];
        }
      }
    }
  }
}

// y[m][j] = act((sum_z slab[z][m][j]) * scale[j] + shift[j])   (split-K partials of the linear layers)
__global__ void reduce_linear_kernel(const float* __restrict__ slab, float* __restrict__ y, int M, int Nout, int ldy,
                                     size_t stride, int nsplit, const float* scale, const float* shift, float slope) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Nout) return;
  int m = (int)(i / Nout), j = (int)(i - (size_t)m * Nout);
  float v = 0.f;
  for (int z = 0; z < nsplit; ++z) v += slab[(size_t)z * stride + (size_t)m * ldy + j];
  if (scale) v *= scale[j];
  if (shift) v += shift[j];
  y[(size_t)m * ldy + j] = lrelu_f(v, slope);
}

// out_bf16[i] = sum_z slab[z][i]   (split-K partials of gather_gemm_dma_kernel; 8 elements per thread)
// S16: the slabs themselves are bf16 (conv8_kernel with G2Args::slab16)
template <bool S16>
__global__ void reduce_slabs_bf16_kernel(const float* __restrict__ slab, uint16_t* __restrict__ out, size_t n8,
                                         size_t stride, int nsplit) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float4 a = make_float4(0, 0, 0, 0), b = a;
  for (int z = 0; z < nsplit; ++z) {
    float4 x, y;
    if (S16) {
      const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(slab) + (size_t)z * stride + i * 8);
      x = make_float4(h16lo_to_f32(t.x), h16hi_to_f32(t.x), h16lo_to_f32(t.y),
                      h16hi_to_f32(t.y));
      y = make_float4(h16lo_to_f32(t.z), h16hi_to_f32(t.z), h16lo_to_f32(t.w),
                      h16hi_to_f32(t.w));
    } else {
      const float4* p = reinterpret_cast<const float4*>(slab + (size_t)z * stride + i * 8);
      x = p[0]; y = p[1];
    }
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
  }
  uint4 o;
  o.x = (uint32_t)f32_to_h16(a.x) | ((uint32_t)f32_to_h16(a.y) << 16);
  o.y = (uint32_t)f32_to_h16(a.z) | ((uint32_t)f32_to_h16(a.w) << 16);
  o.z = (uint32_t)f32_to_h16(b.x) | ((uint32_t)f32_to_h16(b.y) << 16);
  o.w = (uint32_t)f32_to_h16(b.z) | ((uint32_t)f32_to_h16(b.w) << 16);
  reinterpret_cast<uint4*>(out)[i] = o;
}

// ================================================================================================
// weight gradient
// ================================================================================================
struct WArgs {
  const uint16_t* low;   // [pix][O]
  const uint16_t* high;  // [N][Hh][Wh][I]
  float* slab;           // [nsplit][O][16][I]
  int O, I, K;           // K = N*Ho*Wo pixels
  int lgWo, lgHo, Hh, Wh;
  int tiles_c;           // column tiles (over 16*I)
  int klen;              // pixels per split (multiple of 64)
};

constexpr int WROW = 160;  // LDS row stride in bf16 elements (128 + 32 pad = 320 B): conflict-free tr reads

__device__ __forceinline__ s16x4_t lds_tr_read(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(WArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * 64 * WROW];   // 80 KB
  const int t = threadIdx.x;
  const int tile_o = blockIdx.x / g.tiles_c, tile_c = blockIdx.x - tile_o * g.tiles_c;
  const int o0 = tile_o * 128, c0 = tile_c * 128;
  const int zs = blockIdx.y;
  const int k_begin = zs * g.klen;
  const int k_end = min(g.K, k_begin + g.klen);
  const int nkt = (k_end - k_begin + 63) >> 6;

  // staging: chunk (8 channels) of pixel rows pr0 + 16*j
  const int chunk = t & 15, pr0 = t >> 4;
  const int gc = c0 + chunk * 8;           // global column = tap*I + i
  const int tap = gc / g.I, ci = gc - tap * g.I;
  const int kh = tap >> 2, kw = tap & 3;
  const int Wo = 1 << g.lgWo, Ho = 1 << g.lgHo;
  uint4 rl0, rl1, rl2, rl3, rh0, rh1, rh2, rh3;
#define RG_WLOAD_ROW(J, RL, RH, KT)                                                               \
  do {                                                                                            \
    int p = k_begin + (KT) * 64 + pr0 + 16 * (J);                                                 \
    bool ok = p < k_end;                                                                          \
    int pp = ok ? p : 0;                                                                          \
    RL = ld16_if(g.low + (long long)pp * g.O + o0 + chunk * 8, ok);                               \
    int wo = pp & (Wo - 1), ho = (pp >> g.lgWo) & (Ho - 1), n = pp >> (g.lgWo + g.lgHo);          \
    int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;                                               \
    bool v = ok && (unsigned)hi < (unsigned)g.Hh && (unsigned)wi < (unsigned)g.Wh;                \
    long long off = (((long long)n * g.Hh + (v ? hi : 0)) * g.Wh + (v ? wi : 0)) * g.I + ci;      \
    RH = ld16_if(g.high + off, v);                                                                \
  } while (0)
#define RG_WLOAD_TILE(KT)                                                                         \
  do {                                                                                            \
    RG_WLOAD_ROW(0, rl0, rh0, KT); RG_WLOAD_ROW(1, rl1, rh1, KT);                                 \
    RG_WLOAD_ROW(2, rl2, rh2, KT); RG_WLOAD_ROW(3, rl3, rh3, KT);                                 \
  } while (0)
#define RG_WSTORE_TILE(STAGE)                                                                     \
  do {                                                                                            \
    uint16_t* sl_ = lds + (STAGE) * (2 * 64 * WROW);                                              \
    uint16_t* sh_ = sl_ + 64 * WROW;                                                              \
    *reinterpret_cast<uint4*>(sl_ + (pr0)*WROW + chunk * 8) = rl0;                                \
    *reinterpret_cast<uint4*>(sh_ + (pr0)*WROW + chunk * 8) = rh0;                                \
    *reinterpret_cast<uint4*>(sl_ + (pr0 + 16) * WROW + chunk * 8) = rl1;                         \
    *reinterpret_cast<uint4*>(sh_ + (pr0 + 16) * WROW + chunk * 8) = rh1;                         \
    *reinterpret_cast<uint4*>(sl_ + (pr0 + 32) * WROW + chunk * 8) = rl2;                         \
    *reinterpret_cast<uint4*>(sh_ + (pr0 + 32) * WROW + chunk * 8) = rh2;                         \
    *reinterpret_cast<uint4*>(sl_ + (pr0 + 48) * WROW + chunk * 8) = rl3;                         \
    *reinterpret_cast<uint4*>(sh_ + (pr0 + 48) * WROW + chunk * 8) = rh3;                         \
  } while (0)

  const int wave = t >> 6, lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = lane >> 4, idx = lane & 15;
  const int q = idx >> 2, p4 = idx & 3, fh = grp >> 1, cb = grp & 1;
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) {
    RG_WLOAD_TILE(0);
    RG_WSTORE_TILE(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) RG_WLOAD_TILE(kt + 1);
    const uint16_t* sl = lds + cur * (2 * 64 * WROW);
    const uint16_t* sh = sl + 64 * WROW;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int prow = ks * 16 + 8 * fh + q;
      h16x8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint16_t* p = sl + prow * WROW + wm * 64 + i * 32 + 16 * cb + 4 * p4;
        s16x4_t lo = lds_tr_read(p);
        s16x4_t hi = lds_tr_read(p + 4 * WROW);
        s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        fa[i] = __builtin_bit_cast(h16x8_t, v);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint16_t* p = sh + prow * WROW + wn * 64 + j * 32 + 16 * cb + 4 * p4;
        s16x4_t lo = lds_tr_read(p);
        s16x4_t hi = lds_tr_read(p + 4 * WROW);
        s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        fb[j] = __builtin_bit_cast(h16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = rg_mfma_h16_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) RG_WSTORE_TILE(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#undef RG_WLOAD_ROW
#undef RG_WLOAD_TILE
#undef RG_WSTORE_TILE

  // epilogue: slab[zs][o][col]  (col = tap*I + i), 128-byte row segments per store instruction
  const int fr = lane & 31, fh2 = lane >> 5;
  const long long ldw = (long long)16 * g.I;
  float* slab = g.slab + (long long)zs * g.O * ldw;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = o0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh2;
        int c = c0 + wn * 64 + j * 32 + fr;
        slab[(long long)o * ldw + c] = acc[i][j][r];
      }
}

// ================================================================================================
// weight gradient v2: LDS-DMA staging, swizzled [pixel][128 ch] images (256-byte rows, no padding),
// optional second (low, high) segment so that two gradient contributions of the same layer
// (D step: real + fake batch; GP step: primal + tangent) are summed inside ONE launch.
// LDS image: off(row, chunk) = 256*row + 16*(chunk ^ f(row)), f(row) = ((row&3)<<2) | ((row>>2)&3)
// (CDNA guide T10 layout (b)): conflict-free for ds_read_b64_tr_b16 with 4 rows x 64 B per half-wave.
// ================================================================================================
struct W2Args {
  const uint16_t* low[2];
  const uint16_t* high[2];
  unsigned low_bytes[2], high_bytes[2];
  int Kseg[2];           // pixels per segment (Kseg[1] = 0: single segment); Kseg[0] % 64 == 0 when two segments
  float* slab;
  int O, I;
  int lgWo, lgHo, Hh, Wh;
  int tiles_c, klen;
  int tiles_o, nsplit;
  int accumulate;        // nsplit == 1 only: `slab` IS dW[O][16][I] and the tile is added to it
};

__device__ __forceinline__ int wswz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// C3 = true: the 3x3 stride-1 weight gradient of the resize-convolution block -- `high` is the padded upsampled image
// [N][Hh = Ho + 2][Wh = Wo + 2][I], 9 taps, slab columns tap*I + i (9*I need not fill the last 128-column tile) and
// O need not be a multiple of 128 (rows beyond O are never loaded nor stored).
template <bool C3>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(W2Args g) {
  constexpr int NTAP = C3 ? 9 : 16;
  constexpr int STAGE = 2 * 64 * 16;                           // 16-byte slots per stage (low + high)
  __shared__ __attribute__((aligned(16))) uint4 lds[2 * STAGE];   // 64 KB
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  // XCD-aware work mapping (blocks b and b+8 share an XCD / L2): each XCD walks a contiguous range of work
  // items ordered split-major, so the column tiles that read the SAME pixels run back to back on one L2.
  const int ntiles = g.tiles_o * g.tiles_c;
  const int total = ntiles * g.nsplit;
  int wid = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = wid & 7, j = wid >> 3;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int zs = wid / ntiles;
  const int tid = wid - zs * ntiles;
  const int tile_o = tid / g.tiles_c, tile_c = tid - tile_o * g.tiles_c;
  const int o0 = tile_o * 128, c0 = tile_c * 128;
  const int Ktot = g.Kseg[0] + g.Kseg[1];
  const int k_begin = zs * g.klen;
  const int k_end = min(Ktot, k_begin + g.klen);
  const int nkt = (k_end - k_begin + 63) >> 6;
  constexpr unsigned OOB = 0x80000000u;

  const __amdgpu_buffer_rsrc_t rsL0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.low[0], 0, g.low_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.high[0], 0, g.high_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.low[1], 0, g.low_bytes[1], 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.high[1], 0, g.high_bytes[1], 0x00020000);

  // loader lane -> (row within a 4-row group, physical chunk); instruction j of wave w covers rows j*16 + w*4 + (lane>>4)
  const int lrow = wave * 4 + (lane >> 4);
  const int lc = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (wave & 3));      // logical 16-byte chunk (8 channels)
  const int gc = c0 + lc * 8;                                              // global column = tap*I + i
  const int tap = gc / g.I, ci = gc - tap * g.I;
  const int kh = C3 ? tap / 3 : tap >> 2, kw = C3 ? tap - 3 * (tap / 3) : tap & 3;
  const bool col_ok = !C3 || gc < NTAP * g.I;
  const bool o_ok = !C3 || o0 + lc * 8 < g.O;
  const int Wo = 1 << g.lgWo, Ho = 1 << g.lgHo;

  auto issue = [&](int stage, int kt) {
    const int p0 = k_begin + kt * 64;                    // block-uniform
    const bool seg1 = p0 >= g.Kseg[0];
    const int pbase = seg1 ? p0 - g.Kseg[0] : p0;
    const int kend = (seg1 ? k_end - g.Kseg[0] : min(k_end, g.Kseg[0]));
    uint4* sl = lds + stage * STAGE;
    uint4* sh = sl + 64 * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int p = pbase + j * 16 + lrow;
      bool ok = p < kend;
      unsigned lo = (ok && o_ok) ? (unsigned)(((long long)p * g.O + o0 + lc * 8) * 2) : OOB;
      int wo = p & (Wo - 1), ho = (p >> g.lgWo) & (Ho - 1), n = p >> (g.lgWo + g.lgHo);
      int hi = C3 ? ho + kh : 2 * ho - 1 + kh, wi = C3 ? wo + kw : 2 * wo - 1 + kw;
      bool v = ok && col_ok && (unsigned)hi < (unsigned)g.Hh && (unsigned)wi < (unsigned)g.Wh;
      unsigned ho_ = v ? (unsigned)(((((long long)n * g.Hh + hi) * g.Wh + wi) * g.I + ci) * 2) : OOB;
      if (!seg1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL0, (lds_vptr_t)(sl + j * 256 + wave * 64), 16, lo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH0, (lds_vptr_t)(sh + j * 256 + wave * 64), 16, ho_, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL1, (lds_vptr_t)(sl + j * 256 + wave * 64), 16, lo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH1, (lds_vptr_t)(sh + j * 256 + wave * 64), 16, ho_, 0, 0, 0);
      }
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int grp = lane >> 4, idx = lane & 15;
  const int q = idx >> 2, p4 = idx & 3, fh = grp >> 1, cb = grp & 1;
  const int f1 = (q << 2) | (2 * fh), f2 = (q << 2) | (2 * fh + 1);        // wswz(row), wswz(row + 4)
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    const unsigned char* sl = reinterpret_cast<const unsigned char*>(lds + (kt & 1) * STAGE);
    const unsigned char* sh = sl + 64 * 256;
    // all transposed fragment reads of this tile, then the next tile's DMA, then the MFMAs (see gather kernel)
    s16x8_t qa[4][2], qb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int prow = ks * 16 + 8 * fh + q;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ca = wm * 8 + i * 4 + 2 * cb + (p4 >> 1), cbk = wn * 8 + i * 4 + 2 * cb + (p4 >> 1);
        s16x4_t lo = lds_tr_read(reinterpret_cast<const uint16_t*>(sl + prow * 256 + ((ca ^ f1) << 4) + (p4 & 1) * 8));
        s16x4_t hi = lds_tr_read(reinterpret_cast<const uint16_t*>(sl + (prow + 4) * 256 + ((ca ^ f2) << 4) + (p4 & 1) * 8));
        qa[ks][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        lo = lds_tr_read(reinterpret_cast<const uint16_t*>(sh + prow * 256 + ((cbk ^ f1) << 4) + (p4 & 1) * 8));
        hi = lds_tr_read(reinterpret_cast<const uint16_t*>(sh + (prow + 4) * 256 + ((cbk ^ f2) << 4) + (p4 & 1) * 8));
        qb[ks][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    if (kt + 1 < nkt) issue((kt + 1) & 1, kt + 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = rg_mfma_h16_32x32x16(__builtin_bit_cast(h16x8_t, qa[ks][i]),
                                                              __builtin_bit_cast(h16x8_t, qb[ks][j]), acc[i][j], 0, 0,
                                                              0);
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();

  // epilogue: accumulators -> LDS fp32 [128][128] -> 16-byte stores, 512 contiguous bytes per slab row
  float* cs = reinterpret_cast<float*>(lds);
  const int fr = lane & 31, fh2 = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh2) * 128 + wn * 64 + j * 32 + fr] = acc[i][j][r];
  __syncthreads();
  const long long ldw = (long long)NTAP * g.I;
  float* slab = g.slab + (long long)zs * g.O * ldw;
  const int c4 = (t & 31) * 4, rr = t >> 5;
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int row = rr + 8 * p;
    if (C3 && (o0 + row >= g.O || c0 + c4 >= ldw)) continue;
    float4* d = reinterpret_cast<float4*>(slab + (long long)(o0 + row) * ldw + c0 + c4);
    float4 v = *reinterpret_cast<const float4*>(cs + row * 128 + c4);
    if (g.accumulate) {
      const float4 a = *d;
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    *d = v;
  }
}

// ================================================================================================
// packs
// ================================================================================================
// wdn = bf16(w): the fp32 masters of the 4x4 conv layers are tap-major [O][16][I] already (8 elements per thread)
__global__ void cast_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, size_t n8) {
  for (size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x; d < n8; d += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    Vec<float, 8>::ld(w + d * 8, v);
    Vec<h16_t, 8>::st(reinterpret_cast<h16_t*>(out) + d * 8, v);
  }
}
// generic tiled transpose-pack: src[R][Cc] (fp32 or bf16) -> dst[perm(col)][R] bf16.  permute 0: identity;
// 1: col = c*16 + tap -> tap*(Cc/16) + c;  2: the inverse, col = tap*(Cc/16) + c -> c*16 + tap.
__device__ __forceinline__ float tp_load(const float* p) { return *p; }
__device__ __forceinline__ float tp_load(const uint16_t* p) { return h16_to_f32(*p); }
template <typename S>
__global__ __launch_bounds__(256) void transpose_pack_kernel(const S* src, uint16_t* dst, int R, int Cc, int permute) {
  __shared__ float tile[64][65];
  int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    int r = r0 + rr, c = c0 + tx;
    tile[rr][tx] = (r < R && c < Cc) ? tp_load(src + (size_t)r * Cc + c) : 0.f;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    int c = c0 + cc, r = r0 + tx;
    if (c < Cc && r < R) {
      int pc = c;
      if (permute == 1) { int tap = c & 15, ch = c >> 4; pc = tap * (Cc >> 4) + ch; }
      else if (permute == 2) { int q = Cc >> 4, tap = c / q, ch = c - tap * q; pc = ch * 16 + tap; }
      dst[(size_t)pc * R + r] = f32_to_h16(tile[tx][cc]);
    }
  }
}
// bf16 -> bf16 transpose (source = the bf16 shadow the fused Adam keeps next to the fp32 masters): dst[perm(c)][r] =
// src[r][c], 64x64 tiles, 4-byte accesses on both sides.  R and Cc even.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                             int R, int Cc, int permute) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = t + 256 * k, rr = e >> 5, cp = (e & 31) * 2;
    uint32_t v = 0;
    if (r0 + rr < R && c0 + cp < Cc) v = *reinterpret_cast<const uint32_t*>(src + (size_t)(r0 + rr) * Cc + c0 + cp);
    tile[rr][cp] = (uint16_t)v; tile[rr][cp + 1] = (uint16_t)(v >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = t + 256 * k, cc = e >> 5, rp = (e & 31) * 2;
    const int c = c0 + cc, r = r0 + rp;
    if (c < Cc && r < R) {
      int pc = c;
      if (permute == 1) { int tap = c & 15, ch = c >> 4; pc = tap * (Cc >> 4) + ch; }
      *reinterpret_cast<uint32_t*>(dst + (size_t)pc * R + r) = (uint32_t)tile[rp][cc] | ((uint32_t)tile[rp + 1][cc] << 16);
    }
  }
}
// The same transpose for R % 64 == 0, Cc % 128 == 0 (every conv weight of the reference model) with 16-byte accesses on both
// sides: 64 x 128 tile, rows stored as 16 chunks of 8 elements with chunk' = chunk ^ ((row >> 3) & 7), which keeps the
// 16-byte stores aligned and makes the column gathers of the second phase conflict-free (8 row groups x 8 adjacent columns
// per wave-instruction land in 32 different banks).  dst rows are written as 128-byte segments.
// permute 1 (G.0: column c = ch * 16 + tap -> dst row tap * (Cc / 16) + ch): only the destination row changes.
__global__ __launch_bounds__(256) void transpose_bf16_wide_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                                  int R, int Cc, int permute) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64][128];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 128;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = t + 256 * k, row = e >> 4, chunk = e & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + row) * Cc + c0 + chunk * 8);
    *reinterpret_cast<uint4*>(&tile[row][(chunk ^ ((row >> 3) & 7)) * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = t + 256 * k, c = e >> 3, rg = e & 7;
    const int pc = (((c >> 3) ^ rg) << 3) | (c & 7);
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (uint32_t)tile[8 * rg + 2 * j][pc] | ((uint32_t)tile[8 * rg + 2 * j + 1][pc] << 16);
    int orow = c0 + c;
    if (permute == 1) orow = (orow & 15) * (Cc >> 4) + (orow >> 4);
    *reinterpret_cast<uint4*>(dst + (size_t)orow * R + r0 + 8 * rg) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
// The same transpose for up to 8 tensors in ONE launch (the transposed-conv weight images of a whole network after an optimizer
// step: 5 launches of 10-40 us became one; blocks are dealt to the tensors by a prefix table of tile counts).
struct TransMulti {
  const uint16_t* src[8];
  uint16_t* dst[8];
  int R[8], Cc[8];
  int tile_end[8];          // exclusive prefix of 64 x 128 tiles
  int n;
};
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(TransMulti tab) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64][128];
  int i = 0;
  while (i + 1 < tab.n && (int)blockIdx.x >= tab.tile_end[i]) ++i;
  const int local = (int)blockIdx.x - (i ? tab.tile_end[i - 1] : 0);
  const int R = tab.R[i], Cc = tab.Cc[i];
  const uint16_t* __restrict__ src = tab.src[i];
  uint16_t* __restrict__ dst = tab.dst[i];
  const int tx = Cc / 128;
  const int r0 = (local / tx) * 64, c0 = (local % tx) * 128;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = t + 256 * k, row = e >> 4, chunk = e & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + row) * Cc + c0 + chunk * 8);
    *reinterpret_cast<uint4*>(&tile[row][(chunk ^ ((row >> 3) & 7)) * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = t + 256 * k, c = e >> 3, rg = e & 7;
    const int pc = (((c >> 3) ^ rg) << 3) | (c & 7);
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (uint32_t)tile[8 * rg + 2 * j][pc] | ((uint32_t)tile[8 * rg + 2 * j + 1][pc] << 16);
    *reinterpret_cast<uint4*>(dst + (size_t)(c0 + c) * R + r0 + 8 * rg) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
// fp32 [Nout][K] -> bf16 [Np][Kp] (zero padded), 8 outputs = one 16-byte store per thread; the source row pitch K is even on
// every layer of the models here (8-byte aligned rows): four float2 loads, scalar loads at a ragged row end / odd K
__global__ void pack_linear_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Nout, int K, int Np, int Kp) {
  const int kp8 = Kp >> 3;                                   // Kp % 8 == 0 (host)
  const size_t n8 = (size_t)Np * kp8;
  const bool even = (K & 1) == 0 && (reinterpret_cast<uintptr_t>(w) & 7) == 0;
  for (size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x; d < n8; d += (size_t)gridDim.x * blockDim.x) {
    const size_t j = d / kp8;
    const int k = (int)(d - j * kp8) * 8;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (j < (size_t)Nout && k < K) {
      const float* src = w + j * (size_t)K + k;
      if (even && k + 8 <= K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 v = *reinterpret_cast<const float2*>(src + 2 * i);
          x[2 * i] = v.x; x[2 * i + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (k + i < K) x[i] = src[i];
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (uint32_t)f32_to_h16(x[2 * i]) | ((uint32_t)f32_to_h16(x[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(wp + d * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ================================================================================================
// hardware layout self-test (exact small-integer data)
// ================================================================================================
__global__ void selftest_kernel(int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[32 * 16 * 2 + 16 * WROW];
  uint16_t* A = sm;              // A[32][16] row-major (k contiguous)
  uint16_t* Bt = sm + 32 * 16;   // Bt[32 cols][16] (k contiguous)
  uint16_t* P = sm + 2 * 32 * 16;  // [16 pixels][WROW] channel-contiguous image for the tr read
  int lane = threadIdx.x;
  for (int i = lane; i < 32 * 16; i += 64) {
    int r = i / 16, k = i % 16;
    A[i] = f32_to_h16((float)((r * 3 + k * 5) % 7 - 3));
    Bt[i] = f32_to_h16((float)((r * 2 + k * 7 + 1) % 5 - 2));   // asymmetric in (col r, k)
  }
  for (int i = lane; i < 16 * 32; i += 64) {
    int p = i / 32, c = i % 32;
    P[p * WROW + c] = (uint16_t)(p * 64 + c);
  }
  __syncthreads();
  int fr = lane & 31, fh = lane >> 5;
  int bad_mfma = 0, bad_tr = 0;
  // (1) MFMA operand / accumulator maps
  h16x8_t fa = *reinterpret_cast<const h16x8_t*>(A + fr * 16 + fh * 8);
  h16x8_t fb = *reinterpret_cast<const h16x8_t*>(Bt + fr * 16 + fh * 8);
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = rg_mfma_h16_32x32x16(fa, fb, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * fh, col = fr;
    float ref = 0.f;
    for (int k = 0; k < 16; ++k) ref += h16_to_f32(A[row * 16 + k]) * h16_to_f32(Bt[col * 16 + k]);
    if (ref != acc[r]) ++bad_mfma;
  }
  // (2) transposed LDS read: lane must receive P[8*fh + e][16*cb + idx], e = 0..7
  int grp = lane >> 4, idx = lane & 15, q = idx >> 2, p4 = idx & 3, h = grp >> 1, cb = grp & 1;
  const uint16_t* p = P + (8 * h + q) * WROW + 16 * cb + 4 * p4;
  s16x4_t lo = lds_tr_read(p);
  s16x4_t hi = lds_tr_read(p + 4 * WROW);
  for (int e = 0; e < 4; ++e) {
    int want_lo = (8 * h + e) * 64 + 16 * cb + idx;
    int want_hi = (8 * h + 4 + e) * 64 + 16 * cb + idx;
    if ((uint16_t)lo[e] != (uint16_t)want_lo) ++bad_tr;
    if ((uint16_t)hi[e] != (uint16_t)want_hi) ++bad_tr;
  }
  atomicAdd(&out[0], bad_mfma);
  atomicAdd(&out[1], bad_tr);
}

inline unsigned grid_cap(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
bool rg_mfma_conv_supported(int N, int Hq, int Wq, int Kc, int Ncols) {
  return N > 0 && rg_is_pow2(Hq) && rg_is_pow2(Wq) && Kc % 64 == 0 && Ncols % 8 == 0 && Ncols >= 64;
}
bool rg_mfma_plain_supported(int M, int K, int Ncols) { return M > 0 && K % 64 == 0 && Ncols % 8 == 0; }

template <int MODE, int EPI>
static int launch_gather(const char* name, GArgs& g, int nclass, hipStream_t st) {
  int tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.Ncols + 127) / 128;
  hipLaunchKernelGGL((gather_gemm_kernel<MODE, EPI>), dim3(tiles_m * g.tiles_n, nclass), dim3(256), 0, st, g);
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}

static bool use_v1() { return rg_option("conv_v1", 0) == 1; }

// split-K policy of the DMA kernel: only when the grid cannot fill the chip (< 1 block per CU) and K is long
// tile variant / split-K decision of the DMA kernel, shared by the launcher and by rg_mfma_conv_stats_rows
struct GPlan { bool narrow, wide; int nsplit; bool c8; int bm, bn; bool n8, pp, pd; };

static int gather_split(int M, int Ncols, int nclass, int nkt, bool allow, int cap = 4, int min_kt = 16) {
  if (!allow) return 1;
  int bn = Ncols <= 64 ? 64 : 128, bmm = bn == 128 ? 128 : 256;
  long long tiles = (long long)((M + bmm - 1) / bmm) * ((Ncols + bn - 1) / bn) * nclass;
  // fewer than 2 blocks per CU leaves half of the wave slots empty (1 wave/SIMD cannot hide the LDS/DMA waits)
  if (tiles >= 512 || nkt < 2 * min_kt) return 1;
  int s = (int)((512 + tiles - 1) / tiles);
  if (s > cap) s = cap;
  if (s > nkt / min_kt) s = nkt / min_kt;
  return s < 1 ? 1 : s;
}

// 128x128 block with 64x64 wave tiles (4 waves, 2 blocks/CU) by default; 256x64 for <= 64 output columns.  The
// 256x256 block with 128x64 wave tiles (8 waves, 1 block/CU; 42.7 instead of 32 flop per LDS byte) is faster
// (+9..13 %) exactly when its tile count fills the chip without split-K (whose fp32 partials cost more HBM time than
// the tile saves): RNAGAN_CONV_TILE=0 disables it, =5 forces it wherever it fits (with split-K).
// Split-K: conv outputs up to 4 splits; dense weight-streaming layers (M = batch) up to 16 (HBM-bound, the grid
// must cover all CUs to pull full bandwidth); none when a mask is fused into the bf16 epilogue.
// c8: the 8-wave ping-pong kernel of rg_conv8.hip (256x256 or 512x128 tiles) takes the bf16-output conv / plain
// GEMMs whose shape it supports (RNAGAN_CONV8=0: off).
// conv8: 0 off; bit 0 on; bit 2 (default) also where the tile count needs split-K.  With the tap-major k order the split form
// lost to the 128 x 128 kernel (81-88 vs 68-78 us per deep layer at batch 64); with the channel-block-major order (korder) a
// split covers all 16 taps of a channel range, re-reads its input rows from L2 and wins: 72-83 vs 81-91 us, the benchmark
// 12.3 -> 11.8 ms.  Bit 1: parity classes fastest in the block order (neutral).
static int conv8_mode() { return rg_option("conv8", 5); }
static int conv8_blocks_target() { return rg_option("conv8_blocks", 256); }

static GPlan gather_plan(int mode, bool bf16_out, int M, int Ncols, int Cin, int taps, int nclass, bool masked, int Hs = 0,
                         int Ws = 0, bool has_mask = false) {
  const int variant = rg_option("conv_tile", 1);
  const int nkt = taps * (Cin >> 6);
  GPlan pl{};
  const int cpt = Cin >> 6;
  // n8: the 64-column transposed conv with all four parity classes per block (conv8n_kernel).  Measured at batch 64
  // (128 -> 64 channels at 128 x 128): 161-168 us against 110-112 us for the one-class 256 x 64 kernel -- correct
  // (bit-identical) but slower: the layer re-reads every input pixel 16 times (4 classes x 4 taps) through LDS-DMA,
  // the 160 KB of LDS bound the bytes in flight per CU and one wave group issuing at a time halves them again.  Off
  // by default (RNAGAN_NARROW8=1 / rg_set_option("narrow8", 1) selects it); kept as the starting point of a kernel
  // that keeps the input patch resident in LDS instead of re-fetching it per tap.
  // pp: the same layer with the input patch resident in LDS (rg_convp.hip; RNAGAN_CONVP=0: off)
  if (rg_option("convp", RG_CONVP_DEFAULT) && !has_mask && bf16_out && mode == MODE_UP && nclass == 4 && taps == 4 && Ws > 0 &&
      rg_convp_supported(M, Ncols, Cin, Hs, Ws)) {
    pl.pp = true; pl.nsplit = 1; pl.bm = 256; pl.bn = 64;
    return pl;
  }
  // pd: the 64 -> 128 channel stride-2 conv on a 128-pixel-wide input with its parity planes resident in LDS (rg_convd.hip;
  // RNAGAN_CONVD=0: off).  Only where the implicit-GEMM kernel would not split K either (>= 256 of its 512-row tiles): the
  // split / no-split answer of the plan queries then does not depend on which of the two kernels runs.
  if (rg_option("convd", 1) && !masked && bf16_out && mode == MODE_DOWN && nclass == 1 && taps == 16 && Ws > 0 &&
      rg_convd_supported(M, Ncols, Cin, Hs, Ws) && M / 512 >= conv8_blocks_target()) {
    pl.pd = true; pl.nsplit = 1; pl.bm = 256; pl.bn = 128;
    return pl;
  }
  if (rg_option("narrow8", 0) && bf16_out && mode == MODE_UP && nclass == 4 && Ncols == 64 && rg_is_pow2(cpt) && M >= 512) {
    pl.n8 = true; pl.nsplit = 1; pl.bm = 512; pl.bn = 64;
    return pl;
  }
  if (conv8_mode() && bf16_out && (mode == MODE_DOWN || mode == MODE_UP || mode == MODE_PLAIN) &&
      (taps == 1 || rg_is_pow2(cpt))) {
    int bm = 0, bn = 0;
    if (Ncols % 256 == 0 && M >= 256) { bm = 256; bn = 256; }
    else if (Ncols % 128 == 0 && M >= 512) { bm = 512; bn = 128; }
    if (bm && nkt >= 4 && nkt % 2 == 0) {
      const long long tiles = (long long)((M + bm - 1) / bm) * (Ncols / bn) * nclass;
      int ns = 1;
      if (!masked)
        while (tiles * ns < conv8_blocks_target() && ns < 8 && nkt % (ns * 4) == 0 && nkt / (ns * 2) >= 8) ns *= 2;
      // conv8 = 1: only where the tile count fills the chip without split-K (measured at batch 64: with split-K the
      // fp32 slabs of a 256x256 tile cost more than the pipeline gains; the 128x128 2-stage kernel splits less);
      // conv8 & 4: also with split-K
      if (ns == 1 || (conv8_mode() & 4)) {
        pl.c8 = true; pl.bm = bm; pl.bn = bn; pl.nsplit = ns;
        return pl;
      }
    }
  }
  pl.narrow = Ncols <= 64;
  const long long tiles256 = (long long)((M + 255) / 256) * ((Ncols + 255) / 256) * nclass;
  pl.wide = !pl.narrow && bf16_out && M >= 256 && Ncols >= 256 && Ncols % 256 == 0 &&
            (variant == 5 || (variant == 1 && tiles256 >= 256 && tiles256 % 256 == 0));
  if (masked) {
    pl.nsplit = 1;
  } else if (pl.wide) {
    pl.nsplit = 1;
    while (tiles256 * pl.nsplit < 256 && pl.nsplit < 8 && nkt / (pl.nsplit * 2) >= 8) pl.nsplit *= 2;   // one block per CU
  } else {
    pl.nsplit = bf16_out ? gather_split(M, Ncols, nclass, nkt, true) : gather_split(M, Ncols, nclass, nkt, true, 16, 8);
  }
  return pl;
}

size_t rg_mfma_gather_ws_bytes(int mode, int M_out_rows, int M, int Ncols, int Cin, int taps, int nclass) {
  const GPlan pl = gather_plan(mode, true, M, Ncols, Cin, taps, nclass, false);
  return pl.nsplit > 1 ? (size_t)pl.nsplit * M_out_rows * Ncols * sizeof(float) : 0;
}


// partial rows the conv epilogue writes when asked for BatchNorm statistics (0: this launch cannot produce them)
int rg_mfma_conv_stats_rows(int up, int N, int Hlow, int Wlow, int O, int I) {
  const int M = N * Hlow * Wlow, Ncols = up ? I : O, Cin = up ? O : I, taps = up ? 4 : 16, nclass = up ? 4 : 1;
  const size_t a_bytes = up ? (size_t)M * O * 2 : (size_t)M * 4 * I * 2, b_bytes = (size_t)O * 16 * I * 2;
  if (use_v1() || a_bytes >= 0x7fffff00ull || b_bytes >= 0x7fffff00ull) return 0;
  GPlan pl = gather_plan(up ? MODE_UP : MODE_DOWN, true, M, Ncols, Cin, taps, nclass, false, up ? Hlow : 2 * Hlow,
                         up ? Wlow : 2 * Wlow);
  if (pl.nsplit > 1) return 0;
  if (pl.pp) return 4 * rg_convp_tiles(M) * 2;
  if (pl.pd) return rg_convd_stats_rows(M);
  if (pl.n8) return 4 * ((M + 511) / 512) * 8;
  if (pl.c8) return nclass * ((M + pl.bm - 1) / pl.bm) * (pl.bm / 128);
  const int bmm = (pl.narrow || pl.wide) ? 256 : 128, parts = pl.narrow ? 4 : 2;      // BM / wave-tile rows
  return nclass * ((M + bmm - 1) / bmm) * parts;
}

template <int MODE, int EPI>
static int launch_gather2(const char* name, GArgs& g, int nclass, long long rows_out, size_t a_bytes, size_t b_bytes,
                          void* ws, size_t ws_bytes, hipStream_t st) {
  if (use_v1() || a_bytes >= 0x7fffff00ull || b_bytes >= 0x7fffff00ull) return launch_gather<MODE, EPI>(name, g, nclass, st);
  G2Args a2{};
  const GPlan pl = gather_plan(MODE, EPI == EPI_BF16, g.M, g.Ncols, g.Cin, g.taps, nclass, g.mask != nullptr || g.affine,
                               g.Hs, g.Ws, g.mask != nullptr && !g.mask_packed);
  const bool narrow = pl.narrow, wide = pl.wide;
  int nsplit = pl.nsplit;
  size_t need = (size_t)nsplit * rows_out * (EPI == EPI_BF16 ? g.Ncols : g.ldc) * sizeof(float);
  if (nsplit > 1 && (!ws || ws_bytes < need)) nsplit = 1;
  RG_REQUIRE(!g.defer_reduce || (nsplit > 1 && nsplit == pl.nsplit), RG_EUNSUPPORTED,
             "%s: partial (split-K slab) output asked for a launch that does not split (rg_conv_split)", name);
  if (EPI == EPI_LINEAR && (g.Ncols % 8 != 0 || g.ldc % 4 != 0)) nsplit = 1;      // slab rows are written 8 wide
  a2.a_bytes = (unsigned)a_bytes; a2.b_bytes = (unsigned)b_bytes;
  a2.korder = (MODE == MODE_DOWN || MODE == MODE_UP) ? rg_option("korder", 1) : 0;
  a2.nsplit = nsplit; a2.slab = (float*)ws; a2.slab_stride = rows_out * (EPI == EPI_BF16 ? g.Ncols : g.ldc);
  // weight-streaming GEMMs (a batch of <= 64 rows against a large weight matrix: betaVAE layers, G.0): 64-row tile with a
  // 3-deep DMA ring -- the bound is HBM latency x bytes in flight, not the matrix cores (RNAGAN_STREAM_TILE=0: off)
  const int stream_tile = rg_option("stream_tile", 1);
  const bool stream = stream_tile && EPI == EPI_LINEAR && MODE == MODE_PLAIN && g.M <= 64 && !narrow;
  const bool c8 = pl.c8 && nsplit == pl.nsplit;      // (a missing split-K workspace falls back to the 2-stage kernel)
  // bf16 partial tiles (option slab16): only the 8-wave kernel writes them; rg_mfma_conv_slab16 answers the same for the consumer
  a2.slab16 = (EPI == EPI_BF16 && nsplit > 1 && c8 && !pl.pp && !pl.pd && !pl.n8 && rg_option("slab16", RG_SLAB16_DEFAULT)) ? 1 : 0;
  const int bn = c8 ? pl.bn : narrow ? 64 : wide ? 256 : 128, bmm = c8 ? pl.bm : stream ? 64 : (narrow || wide) ? 256 : 128;
  g.tiles_n = (g.Ncols + bn - 1) / bn;
  a2.g = g;
  a2.tiles_m = (g.M + bmm - 1) / bmm;
  dim3 grid(((g.M + bmm - 1) / bmm) * g.tiles_n, nclass, nsplit);
  const int xcd = rg_option("xcd", 1), cfast = rg_option("class_fast", 1);
  if (MODE == MODE_UP && nclass == 4 && cfast && a_bytes > b_bytes && !wide && (!c8 || conv8_mode() & 2)) {   // (measured: -4 % on the 256x256 tile)
    a2.class_fast = 1;
    grid = dim3(grid.x * 4, 1, nsplit);
  }
  a2.xcd_swizzle = (xcd && grid.x % 8 == 0 && grid.x >= 16 && (xcd == 2 || a_bytes > b_bytes)) ? 1 : 0;
  // (measured and rejected for the 64-column tile: 2 waves with 128 x 64 wave tiles, 147-154 us vs 109-112 us)
  RG_REQUIRE(!g.mask_packed || pl.pp, RG_EUNSUPPORTED, "%s: packed mask bits without the patch-resident kernel", name);
  RG_REQUIRE(!g.bwd_z || (c8 && nsplit == 1 && !g.mask && !g.affine && !pl.pp && !pl.n8 && !pl.pd &&
                          (g.bwd_half_m == 0 || g.bwd_half_m % bmm == 0)), RG_EUNSUPPORTED,
             "%s: BatchNorm-backward sums in the epilogue need the unsplit 8-wave kernel (rg_conv_bnbwd_rows)", name);
  if (pl.pd) {
    if constexpr (EPI == EPI_BF16 && MODE == MODE_DOWN) {
      RG_REQUIRE(g.ldc == 128 && g.lgW == 6 && !g.mask && !g.affine && nsplit == 1, RG_EUNSUPPORTED, "%s: convd layout", name);
      a2.g.tiles_n = 1;
      a2.tiles_m = g.M / 256;
      rg_convd_launch(&a2, st);
    }
  } else if (pl.pp) {
    if constexpr (EPI == EPI_BF16 && MODE == MODE_UP) {
      RG_REQUIRE(g.ldc == 64 && g.b_col == g.Cin, RG_EUNSUPPORTED, "%s: convp layout", name);
      a2.g.tiles_n = 1;
      a2.tiles_m = rg_convp_tiles(g.M);
      rg_convp_launch(&a2, st);
    }
  } else if (pl.n8) {
    if constexpr (EPI == EPI_BF16 && MODE == MODE_UP) {
      a2.class_fast = 0;
      a2.tiles_m = (g.M + 511) / 512;
      a2.xcd_swizzle = (xcd && a2.tiles_m % 8 == 0 && a2.tiles_m >= 16) ? 1 : 0;
      a2.lgcpt = rg_ilog2(g.Cin >> 6);
      a2.cmask = (g.Cin >> 6) - 1;
      a2.g.tiles_n = 1;
      rg_conv8n_launch(&a2, (unsigned)a2.tiles_m, st);
    }
  } else if (c8) {
    a2.lgcpt = g.taps == 1 ? 30 : rg_ilog2(g.Cin >> 6);
    a2.cmask = g.taps == 1 ? 0x3fffffff : (g.Cin >> 6) - 1;
    if constexpr (EPI == EPI_BF16 && (MODE == MODE_DOWN || MODE == MODE_UP || MODE == MODE_PLAIN))
      rg_conv8_launch(MODE, &a2, pl.bm, grid.x, grid.y, grid.z, st);
  } else if (stream) {
    if constexpr (EPI == EPI_LINEAR && MODE == MODE_PLAIN)
      hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE, EPI, 64, 128, 3, 128>), grid, dim3(128), 0, st, a2);
  } else if (narrow) {
    bool done = false;
    if constexpr (MODE == MODE_C3 && EPI == EPI_LINEAR) {
      if (g.Ncols <= 32 && rg_option("narrow32", 1)) {     // the image layer of the resize-convolution generator (3 of 8 columns)
        hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE, EPI, 256, 64, 2, 256, 64, 0, 1>), grid, dim3(256), 0, st, a2);
        done = true;
      }
    }
    if (!done) hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE, EPI, 256, 64, 2, 256>), grid, dim3(256), 0, st, a2);
  } else if (wide) {
    hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE, EPI, 256, 256, 2, 512, 128>), grid, dim3(512), 0, st, a2);
  } else {
    hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE, EPI, 128, 128, 2, 256>), grid, dim3(256), 0, st, a2);
  }
  RG_LAUNCH_CHECK(name);
  if (nsplit > 1 && EPI == EPI_LINEAR) {
    size_t n = (size_t)g.M * g.Ncols;
    hipLaunchKernelGGL(reduce_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws,
                       (float*)g.C, g.M, g.Ncols, g.ldc, (size_t)a2.slab_stride, nsplit, g.scale, g.shift, g.slope);
    RG_LAUNCH_CHECK(name);
  } else if (nsplit > 1 && !g.defer_reduce) {
    size_t n8 = (size_t)rows_out * g.Ncols / 8;
    if (a2.slab16)
      hipLaunchKernelGGL(reduce_slabs_bf16_kernel<true>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const float*)ws,
                         (uint16_t*)g.C, n8, (size_t)rows_out * g.Ncols, nsplit);
    else
      hipLaunchKernelGGL(reduce_slabs_bf16_kernel<false>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const float*)ws,
                         (uint16_t*)g.C, n8, (size_t)rows_out * g.Ncols, nsplit);
    RG_LAUNCH_CHECK(name);
  }
  return RG_OK;
}

static void set_bwd_fuse(GArgs& g, const RgBnBwdFuse* bf, int M) {
  if (!bf) return;
  g.bwd_z = (const uint16_t*)bf->z; g.bwd_mean = bf->mean; g.bwd_invstd = bf->invstd; g.bwd_gamma = bf->gamma;
  g.bwd_beta = bf->beta; g.bwd_slope = bf->slope; g.bwd_sums = bf->sums;
  g.bwd_half_m = bf->groups == 2 ? M / 2 : 0;
}

// partial rows the epilogue writes for the consumer's BatchNorm backward (0: this launch has no such form)
int rg_mfma_conv_bnbwd_rows(int up, int N, int Hlow, int Wlow, int O, int I, int groups) {
  const int M = N * Hlow * Wlow, Ncols = up ? I : O, Cin = up ? O : I, taps = up ? 4 : 16, nclass = up ? 4 : 1;
  const size_t a_bytes = up ? (size_t)M * O * 2 : (size_t)M * 4 * I * 2, b_bytes = (size_t)O * 16 * I * 2;
  if (use_v1() || a_bytes >= 0x7fffff00ull || b_bytes >= 0x7fffff00ull || groups < 1 || groups > 2 || Ncols % 8) return 0;
  const GPlan pl = gather_plan(up ? MODE_UP : MODE_DOWN, true, M, Ncols, Cin, taps, nclass, false, up ? Hlow : 2 * Hlow,
                               up ? Wlow : 2 * Wlow);
  if (!pl.c8 || pl.nsplit != 1 || pl.pp || pl.n8) return 0;
  if (groups == 2 && (M % 2 || (M / 2) % pl.bm)) return 0;
  return nclass * ((M + pl.bm - 1) / pl.bm);
}

int rg_mfma_conv_down(const void* x, const void* wdn, void* y, int N, int Hi, int Wi, int I, int O, float* stats,
                      void* ws, size_t ws_bytes, hipStream_t st, int defer_reduce, const RgBnBwdFuse* bf) {
  GArgs g{};
  g.defer_reduce = defer_reduce;
  set_bwd_fuse(g, bf, N * (Hi / 2) * (Wi / 2));
  g.stats = stats;
  g.A = (const uint16_t*)x; g.B = (const uint16_t*)wdn; g.C = y;
  int Ho = Hi / 2, Wo = Wi / 2;
  g.M = N * Ho * Wo; g.Ncols = O; g.Cin = I; g.taps = 16;
  g.lgW = rg_ilog2(Wo); g.lgH = rg_ilog2(Ho); g.Hs = Hi; g.Ws = Wi; g.ldc = O; g.b_col = 16 * I; g.b_tap = I;       // wdn[O][16][I]
  return launch_gather2<MODE_DOWN, EPI_BF16>("conv_down(mfma)", g, 1, g.M, (size_t)N * Hi * Wi * I * 2,
                                             (size_t)O * 16 * I * 2, ws, ws_bytes, st);
}

int rg_mfma_conv_up(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I, const void* mask,
                    float mslope, float* stats, void* ws, size_t ws_bytes, hipStream_t st, const float* scale,
                    const float* shift, float slope, int mask_packed, int defer_reduce, const RgBnBwdFuse* bf) {
  GArgs g{};
  g.defer_reduce = defer_reduce;
  set_bwd_fuse(g, bf, N * Ho * Wo);
  RG_REQUIRE(!mask_packed || (mask && rg_mfma_conv_up_maskbits_supported(N, Ho, Wo, O, I)), RG_EUNSUPPORTED,
             "conv_up: packed mask bits need the patch-resident kernel's shape (128 -> 64 channels, width 16..64)");
  g.mask_packed = mask_packed;
  g.affine = scale != nullptr; g.scale = scale; g.shift = shift; g.slope = slope;
  g.stats = mask ? nullptr : stats;
  g.mask = (const uint16_t*)mask; g.mslope = mslope;
  g.A = (const uint16_t*)x; g.B = (const uint16_t*)wup; g.C = y;
  g.M = N * Ho * Wo; g.Ncols = I; g.Cin = O; g.taps = 4;
  g.lgW = rg_ilog2(Wo); g.lgH = rg_ilog2(Ho); g.Hs = Ho; g.Ws = Wo; g.ldc = I; g.b_col = O; g.b_tap = I * O;        // wup[16][I][O]
  return launch_gather2<MODE_UP, EPI_BF16>("conv_up(mfma)", g, 4, (long long)g.M * 4, (size_t)N * Ho * Wo * O * 2,
                                           (size_t)I * 16 * O * 2, ws, ws_bytes, st);
}

bool rg_mfma_conv_up_maskbits_supported(int N, int Ho, int Wo, int O, int I) {
  return rg_option("convp", RG_CONVP_DEFAULT) && rg_convp_supported(N * Ho * Wo, I, O, Ho, Wo) && (size_t)N * Ho * Wo * O * 2 < 0x7fffff00ull;
}

// ---- fp32 mode, 64-column transposed conv on bf16 planes (rg_conv8f.hip): the 256 x 64 tile of the 2-stage kernel with an fp32
// result; x planes [3][N][Ho][Wo][O], w planes wup[3][16][I][O], y fp32 [N][2Ho][2Wo][I]
bool rg_mfma_conv_up_planes64_supported(int N, int Ho, int Wo, int O, int I, int products) {
  return (products == 3 || products == 6) && I == 64 && O % 64 == 0 && rg_is_pow2(Ho) && rg_is_pow2(Wo) && N * Ho * Wo >= 256 &&
         3ull * N * Ho * Wo * O * 2 < 0x7fffff00ull;
}
int rg_mfma_conv_up_planes64(const void* xp, const void* wp, float* y, int N, int Ho, int Wo, int O, int I, int products,
                             hipStream_t st, const float* maskf, float mslope) {
  G2Args a2{};
  GArgs& g = a2.g;
  g.A = (const uint16_t*)xp; g.B = (const uint16_t*)wp; g.C = y;
  g.maskf = maskf; g.mslope = mslope;
  g.M = N * Ho * Wo; g.Ncols = I; g.Cin = O; g.taps = 4;
  g.lgW = rg_ilog2(Wo); g.lgH = rg_ilog2(Ho); g.Hs = Ho; g.Ws = Wo; g.ldc = I; g.b_col = O; g.b_tap = I * O;
  g.slope = 1.f; g.tiles_n = 1;
  a2.a_plane = (unsigned)((size_t)g.M * O * 2); a2.b_plane = (unsigned)((size_t)I * 16 * O * 2);
  a2.a_bytes = 3 * a2.a_plane; a2.b_bytes = 3 * a2.b_plane;
  a2.korder = rg_option("korder", 1); a2.nsplit = 1;
  a2.tiles_m = (g.M + 255) / 256;
  dim3 grid((unsigned)a2.tiles_m, 4, 1);
  a2.xcd_swizzle = (rg_option("xcd", 1) && grid.x % 8 == 0 && grid.x >= 16) ? 1 : 0;
  if (products == 6) hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE_UP, EPI_LINEAR, 256, 64, 2, 256, 64, 6>), grid, dim3(256), 0, st, a2);
  else hipLaunchKernelGGL((gather_gemm_dma_kernel<MODE_UP, EPI_LINEAR, 256, 64, 2, 256, 64, 3>), grid, dim3(256), 0, st, a2);
  RG_LAUNCH_CHECK("conv_up_planes64");
  return RG_OK;
}

// ---- fp8 e4m3 operands (generator-only inference, BASELINE configs[4]): conv8_kernel<.., EB = 1> only, no split-K
bool rg_mfma_fp8_supported(int M, int K, int Ncols, int taps) {
  const int cpt = K / taps / 128;
  if (K % (taps * 128) != 0 || !(taps == 1 || rg_is_pow2(cpt))) return false;
  const int nkt = K / 128;
  if (nkt < 4 || (nkt & 1)) return false;
  return (Ncols % 256 == 0 && M >= 256) || (Ncols % 128 == 0 && M >= 512);
}
static int launch_conv8_fp8(const char* name, int mode, GArgs& g, int nclass, size_t a_bytes, size_t b_bytes, hipStream_t st) {
  RG_REQUIRE(a_bytes < 0x7fffff00ull && b_bytes < 0x7fffff00ull, RG_EUNSUPPORTED, "%s: operand of 2 GB or more", name);
  G2Args a2{};
  const int bm = g.Ncols % 256 == 0 && g.M >= 256 ? 256 : 512, bn = bm == 256 ? 256 : 128;
  g.in_fp8 = 1;
  g.tiles_n = g.Ncols / bn;
  a2.g = g;
  a2.a_bytes = (unsigned)a_bytes; a2.b_bytes = (unsigned)b_bytes;
  a2.nsplit = 1;
  a2.tiles_m = (g.M + bm - 1) / bm;
  const int cpt = g.Cin / 128;
  a2.lgcpt = g.taps == 1 ? 30 : rg_ilog2(cpt);
  a2.cmask = g.taps == 1 ? 0x3fffffff : cpt - 1;
  dim3 grid((unsigned)(a2.tiles_m * g.tiles_n), (unsigned)nclass, 1);
  a2.xcd_swizzle = (rg_option("xcd", 1) && grid.x % 8 == 0 && grid.x >= 16 && a_bytes > b_bytes) ? 1 : 0;
  rg_conv8_launch(mode, &a2, bm, grid.x, grid.y, grid.z, st);
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}
int rg_mfma_conv_up_fp8(const void* x8, const void* wup8, void* y, int N, int Ho, int Wo, int O, int I, const float* scale,
                        const float* shift, float slope, int out_fp8, hipStream_t st) {
  GArgs g{};
  g.affine = 1; g.scale = scale; g.shift = shift; g.slope = slope; g.out_fp8 = out_fp8;
  g.A = (const uint16_t*)x8; g.B = (const uint16_t*)wup8; g.C = y;
  g.M = N * Ho * Wo; g.Ncols = I; g.Cin = O; g.taps = 4;
  g.lgW = rg_ilog2(Wo); g.lgH = rg_ilog2(Ho); g.Hs = Ho; g.Ws = Wo; g.ldc = I; g.b_col = O; g.b_tap = I * O;        // wup8[16][I][O]
  return launch_conv8_fp8("conv_up_fp8", MODE_UP, g, 4, (size_t)N * Ho * Wo * O, (size_t)I * 16 * O, st);
}
int rg_mfma_gemm_fp8(const void* a8, const void* b8, void* y, int M, int K, int Ncols, const float* scale, const float* shift,
                     float slope, int out_fp8, hipStream_t st) {
  GArgs g{};
  g.affine = 1; g.scale = scale; g.shift = shift; g.slope = slope; g.out_fp8 = out_fp8;
  g.A = (const uint16_t*)a8; g.B = (const uint16_t*)b8; g.C = y;
  g.M = M; g.Ncols = Ncols; g.Cin = K; g.taps = 1; g.lgW = 0; g.lgH = 0; g.Hs = 1; g.Ws = 1; g.ldc = Ncols; g.b_col = K; g.b_tap = 0;
  return launch_conv8_fp8("gemm_fp8", MODE_PLAIN, g, 1, (size_t)M * K, (size_t)Ncols * K, st);
}

// split factor of the plain (unmasked, bf16-output) conv launch of this shape: > 1 means the launch leaves fp32 slabs
// [nsplit][rows_out][Ncols] that a reduction pass (or a fused consumer, rg_splitbn.hip) sums
int rg_mfma_conv_nsplit(int up, int N, int Hlow, int Wlow, int O, int I) {
  const int M = N * Hlow * Wlow, Ncols = up ? I : O, Cin = up ? O : I, taps = up ? 4 : 16, nclass = up ? 4 : 1;
  const size_t a_bytes = up ? (size_t)M * O * 2 : (size_t)M * 4 * I * 2, b_bytes = (size_t)O * 16 * I * 2;
  if (use_v1() || a_bytes >= 0x7fffff00ull || b_bytes >= 0x7fffff00ull) return 1;
  const GPlan pl = gather_plan(up ? MODE_UP : MODE_DOWN, true, M, Ncols, Cin, taps, nclass, false, up ? Hlow : 2 * Hlow,
                               up ? Wlow : 2 * Wlow);
  return pl.nsplit;
}

// 1: the split-K launch of this layer (rg_conv_*_partial, no mask / affine) leaves bf16 partial tiles (launch_gather2: slab16)
int rg_mfma_conv_slab16(int up, int N, int Hlow, int Wlow, int O, int I) {
  const int M = N * Hlow * Wlow, Ncols = up ? I : O, Cin = up ? O : I, taps = up ? 4 : 16, nclass = up ? 4 : 1;
  const size_t a_bytes = up ? (size_t)M * O * 2 : (size_t)M * 4 * I * 2, b_bytes = (size_t)O * 16 * I * 2;
  if (use_v1() || a_bytes >= 0x7fffff00ull || b_bytes >= 0x7fffff00ull) return 0;
  const GPlan pl = gather_plan(up ? MODE_UP : MODE_DOWN, true, M, Ncols, Cin, taps, nclass, false, up ? Hlow : 2 * Hlow,
                               up ? Wlow : 2 * Wlow);
  return (pl.nsplit > 1 && pl.c8 && !pl.pp && !pl.pd && !pl.n8 && rg_option("slab16", RG_SLAB16_DEFAULT)) ? 1 : 0;
}

size_t rg_mfma_conv_ws_bytes(int up, int N, int Hlow, int Wlow, int O, int I) {
  int M = N * Hlow * Wlow;
  if (up) return rg_mfma_gather_ws_bytes(MODE_UP, M * 4, M, I, O, 4, 4);
  return rg_mfma_gather_ws_bytes(MODE_DOWN, M, M, O, I, 16, 1);
}

int rg_mfma_gemm_plain(const void* a, const void* bt, void* c, int M, int K, int Ncols, int ldc, hipStream_t st,
                       const float* scale, const float* shift, float slope) {
  GArgs g{};
  g.affine = scale != nullptr; g.scale = scale; g.shift = shift; g.slope = slope;
  g.A = (const uint16_t*)a; g.B = (const uint16_t*)bt; g.C = c;
  g.M = M; g.Ncols = Ncols; g.Cin = K; g.taps = 1; g.lgW = 0; g.lgH = 0; g.Hs = 1; g.Ws = 1; g.ldc = ldc; g.b_col = K; g.b_tap = 0;
  RG_REQUIRE(ldc == Ncols, RG_EINVAL, "gemm_plain: dense output expected");
  return launch_gather2<MODE_PLAIN, EPI_BF16>("gemm_plain(mfma)", g, 1, M, (size_t)M * K * 2, (size_t)Ncols * K * 2,
                                              nullptr, 0, st);
}

size_t rg_mfma_linear_ws_bytes(int M, int Kpad, int Nout) {
  int s = gather_split(M, Nout, 1, Kpad >> 6, true, 16, 8);
  return s > 1 ? (size_t)s * M * Nout * sizeof(float) : 0;
}

int rg_mfma_linear(const void* a, const void* bt, const float* scale, const float* shift, float* y, int ldy, int M,
                   int Kpad, int Nout, float slope, void* ws, size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(Kpad % 64 == 0, RG_EUNSUPPORTED, "linear(mfma): K_pad %% 64 required");
  // ragged widths (Nout % 8 != 0) are an epilogue feature of the LDS-DMA kernel only (operands below 2 GB)
  RG_REQUIRE(Nout % 8 == 0 || (!use_v1() && (size_t)M * Kpad * 2 < 0x7fffff00ull && (size_t)Nout * Kpad * 2 < 0x7fffff00ull),
             RG_EUNSUPPORTED, "linear(mfma): Nout %% 8 required for operands of 2 GB and more");
  GArgs g{};
  g.A = (const uint16_t*)a; g.B = (const uint16_t*)bt; g.C = y;
  g.M = M; g.Ncols = Nout; g.Cin = Kpad; g.taps = 1; g.lgW = 0; g.lgH = 0; g.Hs = 1; g.Ws = 1; g.ldc = ldy; g.b_col = Kpad; g.b_tap = 0;
  g.scale = scale; g.shift = shift; g.slope = slope;
  return launch_gather2<MODE_PLAIN, EPI_LINEAR>("linear(mfma)", g, 1, M, (size_t)M * Kpad * 2, (size_t)Nout * Kpad * 2,
                                                ws, ws_bytes, st);
}

bool rg_mfma_wgrad_supported(int N, int Ho, int Wo, int O, int I) {
  return N > 0 && rg_is_pow2(Ho) && rg_is_pow2(Wo) && O % 128 == 0 && I % 64 == 0;
}
static int mfma_wgrad_split(int N, int Ho, int Wo, int O, int I) {
  int tiles = (O / 128) * (16 * I / 128);
  int K = N * Ho * Wo;
  int want = (768 + tiles - 1) / tiles;
  int maxs = K / 256;
  if (maxs < 1) maxs = 1;
  int s = want < maxs ? want : maxs;
  return s < 1 ? 1 : s;
}
size_t rg_mfma_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I) {
  return (size_t)mfma_wgrad_split(N, Ho, Wo, O, I) * O * I * 16 * sizeof(float);
}
int rg_mfma_conv_wgrad(const void* low, const void* high, float* dw, int N, int Ho, int Wo, int O, int I,
                       int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  int nsplit = mfma_wgrad_split(N, Ho, Wo, O, I);
  size_t elems = (size_t)O * I * 16;
  RG_REQUIRE(ws && ws_bytes >= (size_t)nsplit * elems * sizeof(float), RG_EWORKSPACE,
             "conv_wgrad(mfma): workspace too small");
  WArgs g{};
  g.low = (const uint16_t*)low; g.high = (const uint16_t*)high; g.slab = (float*)ws;
  g.O = O; g.I = I; g.K = N * Ho * Wo;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_c = 16 * I / 128;
  int klen = (g.K + nsplit - 1) / nsplit;
  g.klen = (klen + 63) / 64 * 64;
  hipLaunchKernelGGL(wgrad_kernel, dim3((O / 128) * g.tiles_c, nsplit), dim3(256), 0, st, g);
  RG_LAUNCH_CHECK("conv_wgrad(mfma)");
  return rg_reduce_slabs((const float*)ws, dw, elems, nsplit, accumulate, 0, 0, st);
}

static int mfma_wgrad_split_k(int K, int O, int I) {
  // 2 blocks are resident per CU: a grid of 512 (or a multiple) fills the 256 CUs without a half-empty last round
  const int target = rg_option("wgrad_blocks", 512);
  int tiles = (O / 128) * (16 * I / 128);
  int want = (target + tiles - 1) / tiles;
  int maxs = K / 256;
  if (maxs < 1) maxs = 1;
  int s = want < maxs ? want : maxs;
  return s < 1 ? 1 : s;
}

// nsplit_out (optional): the split-K launch leaves its fp32 slabs [nsplit][O][16][I] in `ws` and the REDUCTION IS SKIPPED
// (*nsplit_out = nsplit > 1; dw untouched) -- the caller's optimizer step sums the slabs itself (rg_adam_step_slabs); a plan
// without a split writes dw as usual and reports 1.
int rg_mfma_conv_wgrad2(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N,
                        int Ho, int Wo, int O, int I, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                        int* nsplit_out, int* slab_dtype_out) {
  if (nsplit_out) *nsplit_out = 1;
  if (slab_dtype_out) *slab_dtype_out = RG_F32;
  // slabs LEFT to the caller may be bf16 (the two ping-pong kernels; option wslab16): each partial sum rounded once, added in
  // fp32 by rg_adam_step_slabs -- half the bytes written here and read there
  const int s16 = (nsplit_out && slab_dtype_out && rg_option("wslab16", RG_WSLAB16_DEFAULT)) ? 1 : 0;
  const int Kseg = N * Ho * Wo;
  const bool two = low1 != nullptr;
  size_t lowb = (size_t)Kseg * O * 2, highb = (size_t)Kseg * 4 * I * 2;
  if (use_v1() || lowb >= 0x7fffff00ull || highb >= 0x7fffff00ull || (two && Kseg % 64 != 0)) {
    int rc = rg_mfma_conv_wgrad(low0, high0, dw, N, Ho, Wo, O, I, accumulate, ws, ws_bytes, st);
    if (rc || !two) return rc;
    return rg_mfma_conv_wgrad(low1, high1, dw, N, Ho, Wo, O, I, 1, ws, ws_bytes, st);
  }
  const int K = two ? 2 * Kseg : Kseg;
  size_t elems = (size_t)O * I * 16;
  if (rg_option("wgrad8", 1) && rg_wgrad8_supported(K, O, I)) {        // 8-wave ping-pong kernel (rg_wgrad8.hip)
    int per = 0;
    const int ns = rg_wgrad8_split(K, O, I, &per);
    // (measured, batch 64: where the 128 x 128 kernel needs no split-K but this one does -- 128 tiles of 256 x 256 --
    // a single segment is faster there: 76 vs 83 us, no slab pass; with two segments the longer k-loop wins back)
    // (at twice the pixels -- the batched D step -- the longer k-loop wins there too: 134 vs 141 us)
    const bool old_direct = mfma_wgrad_split_k(K, O, I) == 1 && ns > 1 && !two && K < 8192 && rg_option("wgrad8", 1) == 1;
    if (!old_direct && (ns == 1 || (ws && ws_bytes >= (size_t)ns * elems * sizeof(float)))) {
      int rc = rg_wgrad8_launch(low0, high0, low1, high1, ns == 1 ? dw : (float*)ws, Kseg, two ? 1 : 0, O, I, Ho, Wo, ns,
                                per, accumulate, st, s16);
      if (rc || ns == 1) return rc;
      if (nsplit_out) { *nsplit_out = ns; if (s16) *slab_dtype_out = RG_H16; return RG_OK; }
      return rg_reduce_slabs((const float*)ws, dw, elems, ns, accumulate, 0, 0, st);
    }
  }
  // the 128 x 512 form of that kernel (O = 128 layers); option value 2: only for K >= 2^19 pixels (two segments / batch 128)
  if (rg_option("wgrad8n", 1) && rg_wgrad8n_supported(K, O, I) && (rg_option("wgrad8n", 1) != 2 || K >= (1 << 19))) {
    int per = 0;
    const int ns = rg_wgrad8n_split(K, O, I, &per);
    if (ns == 1 || (ws && ws_bytes >= (size_t)ns * elems * sizeof(float))) {
      int rc = rg_wgrad8n_launch(low0, high0, low1, high1, ns == 1 ? dw : (float*)ws, Kseg, two ? 1 : 0, O, I, Ho, Wo, ns,
                                 per, accumulate, st, s16);
      if (rc || ns == 1) return rc;
      if (nsplit_out) { *nsplit_out = ns; if (s16) *slab_dtype_out = RG_H16; return RG_OK; }
      return rg_reduce_slabs((const float*)ws, dw, elems, ns, accumulate, 0, 0, st);
    }
  }
  int nsplit = mfma_wgrad_split_k(K, O, I);
  RG_REQUIRE(ws && ws_bytes >= (size_t)nsplit * elems * sizeof(float), RG_EWORKSPACE,
             "conv_wgrad(mfma): workspace too small");
  W2Args g{};
  g.low[0] = (const uint16_t*)low0; g.high[0] = (const uint16_t*)high0;
  g.low[1] = (const uint16_t*)(two ? low1 : low0); g.high[1] = (const uint16_t*)(two ? high1 : high0);
  g.low_bytes[0] = g.low_bytes[1] = (unsigned)lowb; g.high_bytes[0] = g.high_bytes[1] = (unsigned)highb;
  g.Kseg[0] = Kseg; g.Kseg[1] = two ? Kseg : 0;
  g.slab = (float*)ws; g.O = O; g.I = I;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_c = 16 * I / 128;
  int klen = (K + nsplit - 1) / nsplit;
  g.klen = (klen + 63) / 64 * 64;
  nsplit = (K + g.klen - 1) / g.klen;
  g.tiles_o = O / 128; g.nsplit = nsplit;
  // dW is tap-major [O][16][I] = the slab layout: a single split writes (or adds to) dW straight from its epilogue
  if (nsplit == 1) { g.slab = dw; g.accumulate = accumulate; }
  hipLaunchKernelGGL(wgrad_dma_kernel<false>, dim3((unsigned)(g.tiles_o * g.tiles_c * nsplit)), dim3(256), 0, st, g);
  RG_LAUNCH_CHECK("conv_wgrad(mfma)");
  if (nsplit == 1) return RG_OK;
  if (nsplit_out) { *nsplit_out = nsplit; return RG_OK; }
  return rg_reduce_slabs((const float*)ws, dw, elems, nsplit, accumulate, 0, 0, st);
}

// ---- weight gradient + optimizer step in one launch: the plans of rg_mfma_conv_wgrad2 that run wgrad8_kernel WITHOUT split-K
// (the two 33.5 M-parameter layers at batch 64: D.5 and G.1, 512 tiles) can apply Adam to their tile where it sits in LDS
// instead of writing 4 bytes per parameter that the streaming Adam reads back.
bool rg_mfma_conv_wgrad_adam_supported(int N, int Ho, int Wo, int O, int I, bool two) {
  const int Kseg = N * Ho * Wo;
  const size_t lowb = (size_t)Kseg * O * 2, highb = (size_t)Kseg * 4 * I * 2;
  if (use_v1() || lowb >= 0x7fffff00ull || highb >= 0x7fffff00ull || (two && Kseg % 64 != 0)) return false;
  const int K = two ? 2 * Kseg : Kseg;
  if (!rg_option("wgrad8", 1) || !rg_wgrad8_supported(K, O, I)) return false;
  int per = 0;
  return rg_wgrad8_split(K, O, I, &per) == 1;
}
int rg_mfma_conv_wgrad_adam(const void* low0, const void* high0, const void* low1, const void* high1, int N, int Ho, int Wo,
                            int O, int I, float* p, float* m, float* v, uint16_t* shadow, const float* hyper, hipStream_t st) {
  const bool two = low1 != nullptr;
  RG_REQUIRE(rg_mfma_conv_wgrad_adam_supported(N, Ho, Wo, O, I, two), RG_EUNSUPPORTED,
             "conv_wgrad_adam: no single-split plan of the 256 x 256 kernel for this shape");
  int per = 0;
  rg_wgrad8_split((two ? 2 : 1) * N * Ho * Wo, O, I, &per);
  return rg_wgrad8_adam_launch(low0, high0, low1, high1, N * Ho * Wo, two ? 1 : 0, O, I, Ho, Wo, per, p, m, v, shadow, hyper, st);
}

int rg_mfma_conv_wgrad_wire(const void* low0, const void* high0, const void* low1, const void* high1, int N, int Ho, int Wo,
                            int O, int I, uint16_t* out16, hipStream_t st) {
  const bool two = low1 != nullptr;
  RG_REQUIRE(rg_mfma_conv_wgrad_adam_supported(N, Ho, Wo, O, I, two), RG_EUNSUPPORTED,
             "conv_wgrad_wire: no single-split plan of the 256 x 256 kernel for this shape");
  int per = 0;
  rg_wgrad8_split((two ? 2 : 1) * N * Ho * Wo, O, I, &per);
  return rg_wgrad8_wire_launch(low0, high0, low1, high1, out16, N * Ho * Wo, two ? 1 : 0, O, I, Ho, Wo, per, st);
}

size_t rg_mfma_wgrad2_ws_bytes(int N, int Ho, int Wo, int O, int I) {
  size_t a = (size_t)mfma_wgrad_split_k(N * Ho * Wo, O, I) * O * I * 16 * sizeof(float);
  size_t b = (size_t)mfma_wgrad_split_k(2 * N * Ho * Wo, O, I) * O * I * 16 * sizeof(float);
  size_t m = a > b ? a : b;
  for (int k = 1; k <= 2; ++k)
    if (rg_wgrad8_supported(k * N * Ho * Wo, O, I)) {
      int per = 0;
      const size_t c = (size_t)rg_wgrad8_split(k * N * Ho * Wo, O, I, &per) * O * I * 16 * sizeof(float);
      if (c > m) m = c;
    } else if (rg_wgrad8n_supported(k * N * Ho * Wo, O, I)) {
      int per = 0;
      const size_t c = (size_t)rg_wgrad8n_split(k * N * Ho * Wo, O, I, &per) * O * I * 16 * sizeof(float);
      if (c > m) m = c;
    }
  return m;
}

int rg_mfma_pack_conv_weight(const float* w, void* wdn, void* wup, int O, int I, hipStream_t st) {
  if (wdn) {
    const size_t n8 = (size_t)O * I * 2;     // I % 8 == 0 on the MFMA path
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_cap(n8)), dim3(256), 0, st, w, (uint16_t*)wdn, n8);
    RG_LAUNCH_CHECK("pack_wdn");
  }
  if (wup) {
    // w viewed as [O][16*I] (col = tap*I + i) -> wup[tap][i][O]: a plain transpose
    hipLaunchKernelGGL(transpose_pack_kernel<float>, dim3((I * 16 + 63) / 64, (O + 63) / 64), dim3(256), 0, st, w,
                       (uint16_t*)wup, O, I * 16, 0);
    RG_LAUNCH_CHECK("pack_wup");
  }
  return RG_OK;
}
int rg_mfma_transpose_bf16(const void* src, void* dst, int R, int Cc, int permute, hipStream_t st) {
  RG_REQUIRE(R % 2 == 0 && Cc % 2 == 0, RG_EUNSUPPORTED, "transpose_bf16: even dimensions required");
  if ((permute == 0 || permute == 1) && R % 64 == 0 && Cc % 128 == 0 && ((uintptr_t)src & 15) == 0 &&
      ((uintptr_t)dst & 15) == 0) {
    hipLaunchKernelGGL(transpose_bf16_wide_kernel, dim3(Cc / 128, R / 64), dim3(256), 0, st, (const uint16_t*)src,
                       (uint16_t*)dst, R, Cc, permute);
    RG_LAUNCH_CHECK("transpose_bf16");
    return RG_OK;
  }
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((Cc + 63) / 64, (R + 63) / 64), dim3(256), 0, st, (const uint16_t*)src,
                     (uint16_t*)dst, R, Cc, permute);
  RG_LAUNCH_CHECK("transpose_bf16");
  return RG_OK;
}
// dst_i[Cc_i][R_i] = transpose(src_i[R_i][Cc_i]) for n <= 8 bf16 matrices, one launch (R % 64 == 0, Cc % 128 == 0, 16-byte aligned)
int rg_mfma_transpose_bf16_multi(int n, const void* const* src, void* const* dst, const int* R, const int* Cc, hipStream_t st) {
  RG_REQUIRE(n >= 1 && n <= 8, RG_EINVAL, "transpose_bf16_multi: 1..8 tensors");
  TransMulti tab{};
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    RG_REQUIRE(src[i] && dst[i] && R[i] > 0 && Cc[i] > 0 && R[i] % 64 == 0 && Cc[i] % 128 == 0 && ((uintptr_t)src[i] & 15) == 0 &&
                   ((uintptr_t)dst[i] & 15) == 0, RG_EUNSUPPORTED, "transpose_bf16_multi: tensor %d is not 64 x 128 tileable", i);
    tab.src[i] = (const uint16_t*)src[i]; tab.dst[i] = (uint16_t*)dst[i]; tab.R[i] = R[i]; tab.Cc[i] = Cc[i];
    tiles += (R[i] / 64) * (Cc[i] / 128);
    tab.tile_end[i] = tiles;
  }
  tab.n = n;
  hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, st, tab);
  RG_LAUNCH_CHECK("transpose_bf16_multi");
  return RG_OK;
}
int rg_mfma_pack_g0_weight(const float* w, void* wp, int E, int C, hipStream_t st) {
  // w viewed as [E][C*16] (col = c*16+tap) -> wp[(tap*C + c)][E]
  hipLaunchKernelGGL(transpose_pack_kernel<float>, dim3((C * 16 + 63) / 64, (E + 63) / 64), dim3(256), 0, st, w,
                     (uint16_t*)wp, E, C * 16, 1);
  RG_LAUNCH_CHECK("pack_g0");
  return RG_OK;
}
// G.0 weight gradient dw[e][c][tap] = sum_n z[n][e] * gy[n][tap][c]: a [E x N] x [N x 16C] GEMM whose 16*E*C fp32
// output (268 MB at E = C = 2048) is all that matters -- K is the batch.  Both operands are tiny: they are
// transposed into k-contiguous bf16 images (gy's columns permuted to the master's (c, tap) order on the way),
// then the plain gather-GEMM streams the result out with 16-byte stores.
size_t rg_mfma_g0_wgrad_ws_bytes(int N, int E, int C) { return ((size_t)E + (size_t)16 * C) * N * 2 + 512; }
bool rg_mfma_g0_wgrad_supported(int N, int E, int C) { return N % 64 == 0 && E % 8 == 0 && C % 8 == 0; }
int rg_mfma_g0_wgrad(const float* z, const void* gy, float* dw, int N, int E, int C, void* ws, size_t ws_bytes,
                     hipStream_t st) {
  RG_REQUIRE(ws && ws_bytes >= rg_mfma_g0_wgrad_ws_bytes(N, E, C), RG_EWORKSPACE, "g0_wgrad(mfma): workspace too small");
  uint16_t* zT = (uint16_t*)ws;
  uint16_t* gyP = zT + rg_align_up((size_t)E * N, 128);
  hipLaunchKernelGGL(transpose_pack_kernel<float>, dim3((E + 63) / 64, (N + 63) / 64), dim3(256), 0, st, z, zT, N, E, 0);
  RG_LAUNCH_CHECK("g0_wgrad(pack z)");
  hipLaunchKernelGGL(transpose_pack_kernel<uint16_t>, dim3((16 * C + 63) / 64, (N + 63) / 64), dim3(256), 0, st,
                     (const uint16_t*)gy, gyP, N, 16 * C, 2);
  RG_LAUNCH_CHECK("g0_wgrad(pack gy)");
  return rg_mfma_linear(zT, gyP, nullptr, nullptr, dw, 16 * C, E, N, 16 * C, 1.0f, nullptr, 0, st);
}

// ---- resize-convolution block on the matrix cores (forward): materialise pad = ReflectionPad(1)(bilinear_x2(x)) as
// bf16 NHWC once (HBM-bound pass, 16 bytes per thread), then a 9-tap stride-1 implicit GEMM over it (MODE_C3) with
// the Conv2d bias in the epilogue.
// One workgroup per padded output row (n, i): the two source rows and the vertical weight are block constants, a thread walks
// (j, 8-channel group) items of the row with U = 4 items -- 16 x 16-byte loads -- in flight and NO control flow around the loads
// (items beyond the row end read a clamped address and are not stored: with a branch around them hipcc serialises the loads
// behind s_waitcnt vmcnt(0), rg_skinny.hip's round-5 finding).  The four source pixels of neighbouring outputs overlap: the 4 x
// read amplification is served by L2.
__global__ __launch_bounds__(256) void uppad_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ pad, int N, int H,
                                                         int W, int C) {
  const int C8 = C >> 3, Hp = 2 * H + 2, Wp = 2 * W + 2;
  const int n = blockIdx.x / Hp, i = blockIdx.x - n * Hp;
  int h0, h1;
  float lh;
  up_taps(up_reflect(i, 2 * H), H, h0, h1, lh);
  const h16_t* r0 = reinterpret_cast<const h16_t*>(x) + ((size_t)n * H + h0) * W * C;
  const h16_t* r1 = reinterpret_cast<const h16_t*>(x) + ((size_t)n * H + h1) * W * C;
  h16_t* orow = reinterpret_cast<h16_t*>(pad) + ((size_t)n * Hp + i) * Wp * C;
  const int items = Wp * C8;
  constexpr int U = 4;
  for (int base = threadIdx.x; base < items; base += U * 256) {
    RawVec<h16_t, 8> v[U][4];
    float lw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = min(base + u * 256, items - 1);
      const int j = it / C8, c8 = it - j * C8;
      int w0, w1;
      up_taps(up_reflect(j, 2 * W), W, w0, w1, lw[u]);
      v[u][0].ld(r0 + (size_t)w0 * C + c8 * 8);
      v[u][1].ld(r0 + (size_t)w1 * C + c8 * 8);
      v[u][2].ld(r1 + (size_t)w0 * C + c8 * 8);
      v[u][3].ld(r1 + (size_t)w1 * C + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a[4][8], o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k].cvt(a[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        o[k] = (1.f - lh) * ((1.f - lw[u]) * a[0][k] + lw[u] * a[1][k]) + lh * ((1.f - lw[u]) * a[2][k] + lw[u] * a[3][k]);
      const int it = base + u * 256;
      if (it < items) Vec<h16_t, 8>::st(orow + (size_t)it * 8, o);
    }
  }
}
// w3[o][c][3][3] fp32 -> wp[o][tap][c] bf16
__global__ void pack_w3_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Cout, int Cin) {
  const size_t n = (size_t)Cout * 9 * Cin;
  for (size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (size_t)gridDim.x * blockDim.x) {
    const size_t c = d % Cin, ot = d / Cin;
    const size_t tap = ot % 9, o = ot / 9;
    wp[d] = f32_to_h16(w[(o * Cin + c) * 9 + tap]);
  }
}
bool rg_mfma_upconv3_supported(int N, int H, int W, int Cin, int Cout) {
  const long long M = (long long)N * 4 * H * W;
  return Cin % 64 == 0 && Cout % 8 == 0 && rg_is_pow2(H) && rg_is_pow2(W) && M < 0x7fffffffLL && !use_v1() &&
         (size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * 2 < 0x7fffff00ull;
}
size_t rg_mfma_upconv3_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  return rg_align_up((size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * 2, 256) + rg_align_up((size_t)Cout * 9 * Cin * 2, 256) +
         rg_mfma_gather_ws_bytes(MODE_C3, N * 4 * H * W, N * 4 * H * W, Cout, Cin, 9, 1);
}
int rg_mfma_upconv3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                        int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(ws && ws_bytes >= rg_mfma_upconv3_fwd_ws_bytes(N, H, W, Cin, Cout), RG_EWORKSPACE,
             "upconv3_fwd(mfma): workspace too small");
  const int Hp = 2 * H + 2, Wp = 2 * W + 2;
  const size_t padb = rg_align_up((size_t)N * Hp * Wp * Cin * 2, 256), wpb = rg_align_up((size_t)Cout * 9 * Cin * 2, 256);
  uint16_t* pad = (uint16_t*)ws;
  uint16_t* wp = (uint16_t*)((char*)ws + padb);
  hipLaunchKernelGGL(uppad_bf16_kernel, dim3((unsigned)(N * Hp)), dim3(256), 0, st, (const uint16_t*)x, pad, N, H, W, Cin);
  RG_LAUNCH_CHECK("upconv3_fwd(pad)");
  hipLaunchKernelGGL(pack_w3_kernel, dim3(grid_cap((size_t)Cout * 9 * Cin)), dim3(256), 0, st, w, wp, Cout, Cin);
  RG_LAUNCH_CHECK("upconv3_fwd(pack)");
  GArgs g{};
  g.A = pad; g.B = wp; g.C = y;
  g.M = N * 4 * H * W; g.Ncols = Cout; g.Cin = Cin; g.taps = 9;
  g.lgW = rg_ilog2(2 * W); g.lgH = rg_ilog2(2 * H); g.Hs = Hp; g.Ws = Wp; g.ldc = Cout; g.b_col = 9 * Cin; g.b_tap = Cin;
  g.shift = bias;
  return launch_gather2<MODE_C3, EPI_BF16>("upconv3_fwd(mfma)", g, 1, g.M, (size_t)N * Hp * Wp * Cin * 2,
                                           (size_t)Cout * 9 * Cin * 2, (char*)ws + padb + wpb, ws_bytes - padb - wpb, st);
}

// ---- the generator's image block (Cout <= 8, NCHW fp32 out): same pad + 9-tap GEMM with the output columns padded
// to one 8-wide group (fp32 rows [M][8] in the workspace), then a transposing pass adds the bias and writes NCHW.
__global__ void rows8_to_nchw_kernel(const float* __restrict__ tmp, const float* __restrict__ bias, float* __restrict__ y,
                                     int Cout, unsigned HW, size_t M) {
  for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(tmp + m * 8), b = *reinterpret_cast<const float4*>(tmp + m * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const size_t n = m / HW, p = m - n * HW;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < Cout) y[(n * Cout + c) * HW + p] = v[c] + (bias ? bias[c] : 0.f);
  }
}
bool rg_mfma_upconv3_image_supported(int N, int H, int W, int Cin, int Cout) {
  return Cout <= 8 && rg_mfma_upconv3_supported(N, H, W, Cin, 8);
}
size_t rg_mfma_upconv3_image_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  return rg_align_up((size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * 2, 256) + rg_align_up((size_t)Cout * 9 * Cin * 2, 256) +
         (size_t)N * 4 * H * W * 8 * sizeof(float);
}
int rg_mfma_upconv3_image_fwd(const void* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                              int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(ws && ws_bytes >= rg_mfma_upconv3_image_fwd_ws_bytes(N, H, W, Cin, Cout), RG_EWORKSPACE,
             "upconv3_fwd(mfma, image): workspace too small");
  const int Hp = 2 * H + 2, Wp = 2 * W + 2;
  const size_t padb = rg_align_up((size_t)N * Hp * Wp * Cin * 2, 256), wpb = rg_align_up((size_t)Cout * 9 * Cin * 2, 256);
  uint16_t* pad = (uint16_t*)ws;
  uint16_t* wp = (uint16_t*)((char*)ws + padb);
  float* tmp = (float*)((char*)ws + padb + wpb);
  hipLaunchKernelGGL(uppad_bf16_kernel, dim3((unsigned)(N * Hp)), dim3(256), 0, st, (const uint16_t*)x, pad, N, H, W, Cin);
  RG_LAUNCH_CHECK("upconv3_fwd(pad)");
  hipLaunchKernelGGL(pack_w3_kernel, dim3(grid_cap((size_t)Cout * 9 * Cin)), dim3(256), 0, st, w, wp, Cout, Cin);
  RG_LAUNCH_CHECK("upconv3_fwd(pack)");
  GArgs g{};
  g.A = pad; g.B = wp; g.C = tmp;
  g.M = N * 4 * H * W; g.Ncols = Cout; g.Cin = Cin; g.taps = 9;
  g.lgW = rg_ilog2(2 * W); g.lgH = rg_ilog2(2 * H); g.Hs = Hp; g.Ws = Wp; g.ldc = 8; g.b_col = 9 * Cin; g.b_tap = Cin;
  g.slope = 1.0f;
  int rc = launch_gather2<MODE_C3, EPI_LINEAR>("upconv3_fwd(mfma, image)", g, 1, g.M, (size_t)N * Hp * Wp * Cin * 2,
                                               (size_t)Cout * 9 * Cin * 2, nullptr, 0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(rows8_to_nchw_kernel, dim3(grid_cap((size_t)g.M)), dim3(256), 0, st, tmp, bias, y, Cout,
                     (unsigned)(4 * H * W), (size_t)g.M);
  RG_LAUNCH_CHECK("upconv3_fwd(nchw)");
  return RG_OK;
}

// w3[o][c][3][3] fp32 -> wt[c][tap][o] bf16 (B operand of the data gradient)
__global__ void pack_w3t_kernel(const float* __restrict__ w, uint16_t* __restrict__ wt, int Cout, int Cin) {
  const size_t n = (size_t)Cout * 9 * Cin;
  for (size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (size_t)gridDim.x * blockDim.x) {
    const size_t o = d % Cout, ct = d / Cout;
    const size_t tap = ct % 9, c = ct / 9;
    wt[d] = f32_to_h16(w[(o * Cin + c) * 9 + tap]);
  }
}
// adjoint of (reflection pad o bilinear x2) on a bf16 padded-grid gradient, 8 channels per thread:
// gx[n][h][w][c] = sum over the upsampled pixels (u, v) that read x[h][w], each collecting the padded positions that
// mirror it (same arithmetic as the functor path's adjoint kernel, fp32 accumulation)
__global__ void uppad_adjoint_bf16_kernel(const uint16_t* __restrict__ gpad, uint16_t* __restrict__ gx, int N, int H, int W,
                                          int C) {
  const int C8 = C >> 3, H2 = 2 * H, W2 = 2 * W, Wp = W2 + 2;
  const size_t tot = (size_t)N * H * W * C8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % C8);
    size_t t = idx / C8;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const h16_t* gp = reinterpret_cast<const h16_t*>(gpad) + (size_t)n * (H2 + 2) * Wp * C + c8 * 8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int du = -1; du <= 2; ++du) {
      const int u = 2 * h + du;
      if (u < 0 || u >= H2) continue;
      int a0, a1; float la;
      up_taps(u, H, a0, a1, la);
      const float ch = (a0 == h ? 1.f - la : 0.f) + (a1 == h ? la : 0.f);
      if (ch == 0.f) continue;
      for (int dv = -1; dv <= 2; ++dv) {
        const int v = 2 * w + dv;
        if (v < 0 || v >= W2) continue;
        int b0, b1; float lb;
        up_taps(v, W, b0, b1, lb);
        const float cw = (b0 == w ? 1.f - lb : 0.f) + (b1 == w ? lb : 0.f);
        if (cw == 0.f) continue;
        const float f = ch * cw;
        for (int ri = 0; ri < 3; ++ri) {       // padded rows that read upsampled row u: u+1, and the mirrored border rows
          const int i = ri == 0 ? u + 1 : (ri == 1 ? (u == 1 ? 0 : -1) : (u == H2 - 2 ? H2 + 1 : -1));
          if (i < 0) continue;
          for (int rj = 0; rj < 3; ++rj) {
            const int j = rj == 0 ? v + 1 : (rj == 1 ? (v == 1 ? 0 : -1) : (v == W2 - 2 ? W2 + 1 : -1));
            if (j < 0) continue;
            float gv[8];
            Vec<h16_t, 8>::ld(gp + ((size_t)i * Wp + j) * C, gv);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += f * gv[k];
          }
        }
      }
    }
    Vec<h16_t, 8>::st(reinterpret_cast<h16_t*>(gx) + idx * 8, acc);
  }
}
// data gradient: Cout % 64 == 0 (K = 9 * Cout), Cin % 8 == 0
bool rg_mfma_upconv3_bwd_supported(int N, int H, int W, int Cin, int Cout) {
  const long long Mp = (long long)N * (2 * H + 2) * (2 * W + 2);
  return Cout % 64 == 0 && Cin % 8 == 0 && Mp < 0x7fffffffLL && !use_v1() &&
         (size_t)N * 4 * H * W * Cout * 2 < 0x7f000000ull;
}
size_t rg_mfma_upconv3_bwd_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  const int Mp = N * (2 * H + 2) * (2 * W + 2);
  return rg_align_up((size_t)Mp * Cin * 2, 256) + rg_align_up((size_t)Cout * 9 * Cin * 2, 256) +
         rg_mfma_gather_ws_bytes(MODE_C3T, Mp, Mp, Cin, Cout, 9, 1);
}
int rg_mfma_upconv3_bwd_data(const void* gy, const float* w, void* gx, int N, int H, int W, int Cin, int Cout, void* ws,
                             size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(ws && ws_bytes >= rg_mfma_upconv3_bwd_ws_bytes(N, H, W, Cin, Cout), RG_EWORKSPACE,
             "upconv3_bwd_data(mfma): workspace too small");
  const int Hp = 2 * H + 2, Wp = 2 * W + 2, Mp = N * Hp * Wp;
  const size_t gpb = rg_align_up((size_t)Mp * Cin * 2, 256), wtb = rg_align_up((size_t)Cout * 9 * Cin * 2, 256);
  uint16_t* gpad = (uint16_t*)ws;
  uint16_t* wt = (uint16_t*)((char*)ws + gpb);
  hipLaunchKernelGGL(pack_w3t_kernel, dim3(grid_cap((size_t)Cout * 9 * Cin)), dim3(256), 0, st, w, wt, Cout, Cin);
  RG_LAUNCH_CHECK("upconv3_bwd_data(pack)");
  GArgs g{};
  g.A = (const uint16_t*)gy; g.B = wt; g.C = gpad;
  g.M = Mp; g.Ncols = Cin; g.Cin = Cout; g.taps = 9;
  g.Hs = 2 * H; g.Ws = 2 * W; g.ldc = Cin; g.b_col = 9 * Cout; g.b_tap = Cout;
  int rc = launch_gather2<MODE_C3T, EPI_BF16>("upconv3_bwd_data(mfma)", g, 1, g.M, (size_t)N * 4 * H * W * Cout * 2,
                                              (size_t)Cout * 9 * Cin * 2, (char*)ws + gpb + wtb, ws_bytes - gpb - wtb, st);
  if (rc) return rc;
  hipLaunchKernelGGL(uppad_adjoint_bf16_kernel, dim3(grid_cap((size_t)N * H * W * (Cin >> 3))), dim3(256), 0, st, gpad,
                     (uint16_t*)gx, N, H, W, Cin);
  RG_LAUNCH_CHECK("upconv3_bwd_data(adjoint)");
  return RG_OK;
}

// weight gradient: dw[o][c][kh][kw] (+)= sum_pixels gy[p][o] * pad[p + (kh, kw)][c] -- the pixel-contracting kernel of
// the 4x4 layers with 9 stride-1 taps over the re-materialised padded image; slabs are [o][tap][c], the reduction
// permutes them into the PyTorch layout of the 3x3 master.
__global__ void reduce_w3_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int nsplit,
                                       int accumulate, int slab_rows) {
  const size_t n = (size_t)Cout * 9 * Cin, zstride = (size_t)slab_rows * 9 * Cin;
  for (size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (size_t)gridDim.x * blockDim.x) {
    const size_t c = d % Cin, ot = d / Cin;
    const size_t tap = ot % 9, o = ot / 9;
    float v = 0.f;
    for (int z = 0; z < nsplit; ++z) v += slab[z * zstride + d];
    float* out = dw + (o * Cin + c) * 9 + tap;
    *out = accumulate ? *out + v : v;
  }
}
static int upconv3_wgrad_split(int K, int Cout, int Cin) {
  const int tiles = ((Cout + 127) / 128) * ((9 * Cin + 127) / 128);
  int want = (512 + tiles - 1) / tiles, maxs = K / 256;
  if (maxs < 1) maxs = 1;
  const int s = want < maxs ? want : maxs;
  return s < 1 ? 1 : s;
}
bool rg_mfma_upconv3_wgrad_supported(int N, int H, int W, int Cin, int Cout) {
  return Cout % 8 == 0 && Cin % 8 == 0 && rg_is_pow2(H) && rg_is_pow2(W) && !use_v1() &&
         (size_t)N * 4 * H * W * Cout * 2 < 0x7fffff00ull && (size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * 2 < 0x7fffff00ull;
}
size_t rg_mfma_upconv3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  return rg_align_up((size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * 2, 256) +
         (size_t)upconv3_wgrad_split(N * 4 * H * W, Cout, Cin) * Cout * 9 * Cin * sizeof(float);
}
int rg_mfma_upconv3_wgrad(const void* gy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(ws && ws_bytes >= rg_mfma_upconv3_wgrad_ws_bytes(N, H, W, Cin, Cout), RG_EWORKSPACE,
             "upconv3_wgrad(mfma): workspace too small");
  const int Hp = 2 * H + 2, Wp = 2 * W + 2, K = N * 4 * H * W;
  const size_t padb = rg_align_up((size_t)N * Hp * Wp * Cin * 2, 256);
  uint16_t* pad = (uint16_t*)ws;
  float* slab = (float*)((char*)ws + padb);
  hipLaunchKernelGGL(uppad_bf16_kernel, dim3((unsigned)(N * Hp)), dim3(256), 0, st, (const uint16_t*)x, pad, N, H, W, Cin);
  RG_LAUNCH_CHECK("upconv3_wgrad(pad)");
  int nsplit = upconv3_wgrad_split(K, Cout, Cin);
  W2Args g{};
  g.low[0] = g.low[1] = (const uint16_t*)gy; g.high[0] = g.high[1] = pad;
  g.low_bytes[0] = g.low_bytes[1] = (unsigned)((size_t)K * Cout * 2);
  g.high_bytes[0] = g.high_bytes[1] = (unsigned)((size_t)N * Hp * Wp * Cin * 2);
  g.Kseg[0] = K; g.Kseg[1] = 0;
  g.slab = slab; g.O = Cout; g.I = Cin;
  g.lgWo = rg_ilog2(2 * W); g.lgHo = rg_ilog2(2 * H); g.Hh = Hp; g.Wh = Wp;
  g.tiles_c = (9 * Cin + 127) / 128;
  const int klen = (K + nsplit - 1) / nsplit;
  g.klen = (klen + 63) / 64 * 64;
  nsplit = (K + g.klen - 1) / g.klen;
  g.tiles_o = (Cout + 127) / 128; g.nsplit = nsplit;
  hipLaunchKernelGGL(wgrad_dma_kernel<true>, dim3((unsigned)(g.tiles_o * g.tiles_c * nsplit)), dim3(256), 0, st, g);
  RG_LAUNCH_CHECK("upconv3_wgrad(mfma)");
  hipLaunchKernelGGL(reduce_w3_slabs_kernel, dim3(grid_cap((size_t)Cout * 9 * Cin)), dim3(256), 0, st, slab, dw, Cout, Cin,
                     nsplit, accumulate, Cout);
  RG_LAUNCH_CHECK("upconv3_wgrad(reduce)");
  return RG_OK;
}
// image block (gy NCHW fp32, Cout <= 8): gy is re-laid as bf16 rows [pixel][8] (zero padded) and takes the same kernel
// with an 8-row output tile
__global__ void nchw_to_rows8_kernel(const float* __restrict__ gy, uint16_t* __restrict__ rows, int Cout, unsigned HW, size_t M) {
  for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
    const size_t n = m / HW, p = m - n * HW;
    uint32_t h[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) h[c] = c < Cout ? (uint32_t)f32_to_h16(gy[(n * Cout + c) * HW + p]) : 0u;
    uint4 o;
    o.x = h[0] | (h[1] << 16); o.y = h[2] | (h[3] << 16); o.z = h[4] | (h[5] << 16); o.w = h[6] | (h[7] << 16);
    *reinterpret_cast<uint4*>(rows + m * 8) = o;
  }
}
size_t rg_mfma_upconv3_image_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  return rg_align_up((size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * 2, 256) + rg_align_up((size_t)N * 4 * H * W * 8 * 2, 256) +
         (size_t)upconv3_wgrad_split(N * 4 * H * W, 8, Cin) * 8 * 9 * Cin * sizeof(float);
}
int rg_mfma_upconv3_image_wgrad(const float* gy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout,
                                int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(ws && ws_bytes >= rg_mfma_upconv3_image_wgrad_ws_bytes(N, H, W, Cin, Cout), RG_EWORKSPACE,
             "upconv3_wgrad(mfma, image): workspace too small");
  const int Hp = 2 * H + 2, Wp = 2 * W + 2, K = N * 4 * H * W;
  const size_t padb = rg_align_up((size_t)N * Hp * Wp * Cin * 2, 256), rowb = rg_align_up((size_t)K * 8 * 2, 256);
  uint16_t* pad = (uint16_t*)ws;
  uint16_t* rows = (uint16_t*)((char*)ws + padb);
  float* slab = (float*)((char*)ws + padb + rowb);
  hipLaunchKernelGGL(uppad_bf16_kernel, dim3((unsigned)(N * Hp)), dim3(256), 0, st, (const uint16_t*)x, pad, N, H, W, Cin);
  RG_LAUNCH_CHECK("upconv3_wgrad(pad)");
  hipLaunchKernelGGL(nchw_to_rows8_kernel, dim3(grid_cap((size_t)K)), dim3(256), 0, st, gy, rows, Cout, (unsigned)(4 * H * W),
                     (size_t)K);
  RG_LAUNCH_CHECK("upconv3_wgrad(rows)");
  int nsplit = upconv3_wgrad_split(K, 8, Cin);
  W2Args g{};
  g.low[0] = g.low[1] = rows; g.high[0] = g.high[1] = pad;
  g.low_bytes[0] = g.low_bytes[1] = (unsigned)((size_t)K * 8 * 2);
  g.high_bytes[0] = g.high_bytes[1] = (unsigned)((size_t)N * Hp * Wp * Cin * 2);
  g.Kseg[0] = K; g.Kseg[1] = 0;
  g.slab = slab; g.O = 8; g.I = Cin;
  g.lgWo = rg_ilog2(2 * W); g.lgHo = rg_ilog2(2 * H); g.Hh = Hp; g.Wh = Wp;
  g.tiles_c = (9 * Cin + 127) / 128;
  const int klen = (K + nsplit - 1) / nsplit;
  g.klen = (klen + 63) / 64 * 64;
  nsplit = (K + g.klen - 1) / g.klen;
  g.tiles_o = 1; g.nsplit = nsplit;
  hipLaunchKernelGGL(wgrad_dma_kernel<true>, dim3((unsigned)(g.tiles_c * nsplit)), dim3(256), 0, st, g);
  RG_LAUNCH_CHECK("upconv3_wgrad(mfma, image)");
  hipLaunchKernelGGL(reduce_w3_slabs_kernel, dim3(grid_cap((size_t)Cout * 9 * Cin)), dim3(256), 0, st, slab, dw, Cout, Cin,
                     nsplit, accumulate, 8);
  RG_LAUNCH_CHECK("upconv3_wgrad(reduce)");
  return RG_OK;
}

int rg_mfma_pack_linear_weight(const float* w, void* wp, int Nout, int K, int Np, int Kp, hipStream_t st) {
  RG_REQUIRE(Kp % 8 == 0 && (((uintptr_t)wp) & 15) == 0, RG_EINVAL, "pack_linear: Kp %% 8 == 0 and a 16-byte aligned image required");
  hipLaunchKernelGGL(pack_linear_kernel, dim3(grid_cap((size_t)Np * Kp / 8)), dim3(256), 0, st, w, (uint16_t*)wp, Nout, K,
                     Np, Kp);
  RG_LAUNCH_CHECK("pack_linear");
  return RG_OK;
}

extern "C" int rg_selftest_layouts(int* detail, void* stream) {
  RG_REQUIRE(detail, RG_EINVAL, "selftest: detail must point at 2 device ints");
  hipStream_t st = rg_stream(stream);
  hipError_t e = hipMemsetAsync(detail, 0, 2 * sizeof(int), st);
  RG_REQUIRE(e == hipSuccess, RG_EHIP, "selftest: memset failed");
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, detail);
  RG_LAUNCH_CHECK("selftest");
  return RG_OK;
}
