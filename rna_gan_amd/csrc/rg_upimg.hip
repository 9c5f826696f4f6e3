// rg_upimg.hip -- the IMAGE block of DCGANUpGenerator (src/dcgan.py:45-56,76-84: Upsample(x2, bilinear) + ReflectionPad2d(1) +
// Conv2d(64 -> 3, 3 x 3)) without the materialised upsample + pad image: at 128 -> 256 pixels and batch 64 that
// image is 545 MB (bf16, 64 channels) written by one kernel and gathered nine times by the next, for an output of 50 MB.
//
// Forward, one workgroup (4 waves) per 8 x 32 output tile of one image, three phases, everything between them in LDS:
//   0. the 6 x 18 low-resolution pixels the tile depends on (clamped at the image border)           global -> LDS, 13.5 KB
//   1. P = pad(upsample(x)) for the tile's 10 x 34 padded positions: bilinear weights 0.25 / 0.75 in fp32 from four staged
//      pixels, rounded ONCE to the 16-bit type (the arithmetic and the rounding of uppad_bf16_kernel: the results of the two
//      paths are bit-identical)                                                                         LDS -> LDS, 42.5 KB
//   2. nine taps x two 32-channel steps of v_mfma_f32_16x16x32: A = the weights (row = output channel, 3 of 16 live; the 18
//      fragments stay in registers for the whole launch), B = 16 consecutive pixels of a P row at the tap's offset (one
//      ds_read_b128 per MFMA; the channel chunk of a pixel is XOR-swizzled with bits 1..3 of its column so that the 16 lanes
//      of a read hit 16 different bank groups), D = [channel][pixel]: lanes 0..15 hold a pixel's three channels -> fp32 NCHW rows.
// HBM traffic: x once (+ the tiles' halo, from L2) and y once.  Batch 64 at 128 -> 256: 240 us against 633 us for uppad + 9-tap GEMM +
// NCHW pass (the remaining time is phase 1's vector arithmetic, ~110 us, the 64-byte output segments and the tile prologues).
#include "rg_common.h"
#include "rg_internal.h"

namespace {

typedef rg_h16x8 h16x8_t;
typedef rg_f32x4 f32x4_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_vptr_t;

constexpr int UI_TR = 8, UI_TC = 32;                            // output tile
constexpr int UI_PR = UI_TR + 2, UI_PC = UI_TC + 2;             // padded positions of the tile: 10 x 34
constexpr int UI_XR = UI_TR / 2 + 2, UI_XC = UI_TC / 2 + 2;     // low-resolution pixels behind them: 6 x 18
constexpr int UI_XITEMS = UI_XR * UI_XC * 8;                    // 16-byte items (8 channels): 864
constexpr int UI_PITEMS = UI_PR * UI_PC * 8;                    // 2720
static_assert(UI_PC == 34, "phase 1 maps threads to 32 + 2 padded columns");

__device__ __forceinline__ void ui_unpack(const uint4& v, float* o) {
  o[0] = h16lo_to_f32(v.x); o[1] = h16hi_to_f32(v.x); o[2] = h16lo_to_f32(v.y); o[3] = h16hi_to_f32(v.y);
  o[4] = h16lo_to_f32(v.z); o[5] = h16hi_to_f32(v.z); o[6] = h16lo_to_f32(v.w); o[7] = h16hi_to_f32(v.w);
}
__device__ __forceinline__ uint32_t ui_pack2(float a, float b) { return (uint32_t)f32_to_h16(a) | ((uint32_t)f32_to_h16(b) << 16); }

// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() is a fence too: it drains the vector-memory
// queue (s_waitcnt vmcnt(0)), i.e. the NEXT tile's patch loads that are meant to stay in flight across a tile's phases.
#define UI_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// byte offset of (padded row i, padded column j, 8-channel chunk c) in the P tile
__device__ __forceinline__ int ui_paddr(int i, int j, int c) { return ((i * UI_PC + j) * 8 + (c ^ ((j >> 1) & 7))) * 16; }

// MODE 0: forward (w, bias -> y).  MODE 1: weight gradient: `y` is the image gradient gy (fp32 NCHW, READ), `part` receives this
// workgroup's partial dW[o][64][3][3] over its tiles (summed by upimg_wgrad_reduce_kernel: a fixed order, no atomics).  Phases 0
// and 1 are the forward's; phase 2 contracts over the PIXELS of an output row: D[o][ci] += gy^T[o][32 pixels] P[32 pixels][ci],
// A = 8 consecutive pixels of one gy channel per lane (rounded to the 16-bit type as the materialising path rounds gy), B = the
// P tile read through ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered pixel-contiguous per channel);
// wave w owns input channels 16 w .. + 15 and keeps its nine tap accumulators over all its tiles.
template <int MODE>
__global__ __launch_bounds__(256, 2) void upimg_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ part, int H, int W, int Cout, int tiles_x, int tiles_y,
                                                       int total) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[UI_XITEMS * 16];
  __shared__ __attribute__((aligned(16))) unsigned char pt[UI_PITEMS * 16];
  __shared__ __attribute__((aligned(16))) float gys[MODE == 1 ? 4 * UI_TR * UI_TC : 4];      // [o][row][32] (wgrad)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lp = lane & 15, lq = lane >> 4;
  const int H2 = 2 * H, W2 = 2 * W;

  // forward: A operand: row lp = output channel (zero rows beyond Cout), k = input channels 32 ks + 8 lq .. + 7 of tap (dy, dx)
  h16x8_t af[9][2];
  float bv[4];
  f32x4_t wacc[9];                                    // weight gradient: D[o = 4 lq + r][ci = 16 wave + lp] per tap
  if constexpr (MODE == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = lp < Cout ? w[((size_t)lp * 64 + ks * 32 + lq * 8 + e) * 9 + tap] : 0.f;
        const uint4 pk = make_uint4(ui_pack2(v[0], v[1]), ui_pack2(v[2], v[3]), ui_pack2(v[4], v[5]), ui_pack2(v[6], v[7]));
        af[tap][ks] = __builtin_bit_cast(h16x8_t, pk);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (bias && 4 * lq + r < Cout) ? bias[4 * lq + r] : 0.f;
  } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wacc[tap] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // the low-resolution patch of a tile: four 16-byte loads per thread (clamped addresses, no branch around them), issued ONE TILE
  // AHEAD into registers -- a tile's three phases take a few microseconds, an HBM / L2 round trip in front of each would double that
  // (four NAMED registers and a macro: as an array captured by a lambda the compiler turned the buffer into an LDS allocation --
  // AMDGPUPromoteAlloca -- and waited for every load right behind its issue)
  uint4 xv0, xv1, xv2, xv3;
  xv0 = xv1 = xv2 = xv3 = make_uint4(0, 0, 0, 0);
  float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);        // weight gradient: this thread's float4 of the gy tile [o = t >> 6][row][32]
#define UI_FETCH1(dst, k, xn_, hb_, wb_)                                                          \
  do {                                                                                            \
    const int it_ = min(t + 256 * (k), UI_XITEMS - 1);                                            \
    const int px_ = it_ >> 3, ch_ = it_ & 7;                                                      \
    const int r_ = px_ / UI_XC, c_ = px_ - r_ * UI_XC;                                            \
    const int h_ = min(max((hb_) + r_, 0), H - 1), w_ = min(max((wb_) + c_, 0), W - 1);           \
    dst = *reinterpret_cast<const uint4*>((xn_) + ((size_t)h_ * W + w_) * 64 + ch_ * 8);          \
  } while (0)
#define UI_FETCH(tile_)                                                                           \
  do {                                                                                            \
    const int tx_ = (tile_) % tiles_x, ty_ = ((tile_) / tiles_x) % tiles_y, n_ = (tile_) / (tiles_x * tiles_y);   \
    const int hb_ = ((ty_ * UI_TR) >> 1) - 1, wb_ = ((tx_ * UI_TC) >> 1) - 1;                     \
    const uint16_t* xn_ = x + (size_t)n_ * H * W * 64;                                            \
    UI_FETCH1(xv0, 0, xn_, hb_, wb_); UI_FETCH1(xv1, 1, xn_, hb_, wb_);                           \
    UI_FETCH1(xv2, 2, xn_, hb_, wb_); UI_FETCH1(xv3, 3, xn_, hb_, wb_);                           \
    if constexpr (MODE == 1) {                                                                    \
      const int o_ = min(t >> 6, Cout - 1), row_ = (t >> 3) & 7, x4_ = (t & 7) * 4;               \
      gv = *reinterpret_cast<const float4*>(y + (((size_t)n_ * Cout + o_) * H2 + ty_ * UI_TR + row_) * W2 + tx_ * UI_TC + x4_); \
    }                                                                                             \
  } while (0)
  if ((int)blockIdx.x < total) UI_FETCH((int)blockIdx.x);

  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int oy0 = ty * UI_TR, ox0 = tx * UI_TC;
    const int hb = (oy0 >> 1) - 1, wb = (ox0 >> 1) - 1;            // low-resolution pixel of xs[0][0] (before clamping)

    // ---- phase 0: this tile's patch registers -> LDS (every wave is past the previous tile's phase 1, the only reader of xs),
    // then the NEXT tile's loads go out
    *reinterpret_cast<uint4*>(xs + t * 16) = xv0;
    *reinterpret_cast<uint4*>(xs + (t + 256) * 16) = xv1;
    *reinterpret_cast<uint4*>(xs + (t + 512) * 16) = xv2;
    if (t + 768 < UI_XITEMS) *reinterpret_cast<uint4*>(xs + (t + 768) * 16) = xv3;
    UI_BARRIER();          // (also: every wave is done with the MFMAs of the previous tile before P is overwritten below)
    if constexpr (MODE == 1)       // gy tile: read in phase 2, so it is replaced only behind the barrier above (channels beyond Cout: zero)
      *reinterpret_cast<float4*>(gys + t * 4) = (t >> 6) < Cout ? gv : make_float4(0.f, 0.f, 0.f, 0.f);
    if (tile + (int)gridDim.x < total) UI_FETCH(tile + (int)gridDim.x);

    // ---- phase 1: P for the tile's padded positions.  A thread owns ONE (padded column, 8-channel chunk) pair -- column
    // t >> 3 (threads 0..15 also column 32 + (t >> 3)) -- and walks the ten padded rows.  The horizontal interpolation of its
    // column is formed ONCE per staged low-resolution row (six of them, kept in registers); padded row i then blends staged rows
    // i >> 1 and (i >> 1) + 1 with 0.75 / 0.25 (i even) or 0.25 / 0.75 (i odd) -- compile-time everywhere except the image's
    // first and last padded row, where the reflection picks other rows (redone below for those tiles).  Packed fp32 arithmetic
    // (v_pk_mul_f32 / v_pk_add_f32) in the order of uppad_bf16_kernel's expression: the same bits.
    {
      const int ch = t & 7;
      auto column = [&](int j) __attribute__((always_inline)) {
        int w0, w1;
        float lw;
        up_taps(up_reflect(ox0 + j, W2), W, w0, w1, lw);
        const int c0 = w0 - wb, c1 = w1 - wb;
        const f32x2_t vlw = {lw, lw}, vlw1 = {1.f - lw, 1.f - lw};
        auto hrow = [&](int r, f32x2_t* o) __attribute__((always_inline)) {     // (1 - lw) x[r][w0] + lw x[r][w1], 8 channels
          const uint4 q0 = *reinterpret_cast<const uint4*>(xs + ((r * UI_XC + c0) * 8 + ch) * 16);
          const uint4 q1 = *reinterpret_cast<const uint4*>(xs + ((r * UI_XC + c1) * 8 + ch) * 16);
          const uint32_t d0[4] = {q0.x, q0.y, q0.z, q0.w}, d1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2_t a0 = {h16lo_to_f32(d0[e]), h16hi_to_f32(d0[e])}, a1 = {h16lo_to_f32(d1[e]), h16hi_to_f32(d1[e])};
            o[e] = vlw1 * a0 + vlw * a1;
          }
        };
        auto blend = [&](int i, const f32x2_t* x0, const f32x2_t* x1, float lh) __attribute__((always_inline)) {
          const f32x2_t vlh = {lh, lh}, vlh1 = {1.f - lh, 1.f - lh};
          uint32_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2_t v = vlh1 * x0[e] + vlh * x1[e];
            o[e] = ui_pack2(v[0], v[1]);
          }
          *reinterpret_cast<uint4*>(pt + ui_paddr(i, j, ch)) = make_uint4(o[0], o[1], o[2], o[3]);
        };
        f32x2_t hh[UI_XR][4];
#pragma unroll
        for (int r = 0; r < UI_XR; ++r) hrow(r, hh[r]);
#pragma unroll
        for (int i = 0; i < UI_PR; ++i) blend(i, hh[i >> 1], hh[(i >> 1) + 1], (i & 1) ? 0.75f : 0.25f);
        // the image's first / last padded row mirror upsampled row 1 / 2 H - 2: other staged rows than the pattern above
        auto mirror = [&](int i) __attribute__((always_inline)) {
          int h0, h1;
          float lh;
          up_taps(up_reflect(oy0 + i, H2), H, h0, h1, lh);
          f32x2_t x0[4], x1[4];
          hrow(h0 - hb, x0);
          hrow(h1 - hb, x1);
          blend(i, x0, x1, lh);
        };
        if (oy0 == 0) mirror(0);
        if (oy0 + UI_TR == H2) mirror(UI_PR - 1);
      };
      column(t >> 3);
      if (t < 16) column(32 + (t >> 3));
    }
    UI_BARRIER();

    if constexpr (MODE == 0) {
      // ---- phase 2: wave `wave` owns output rows 2 wave, 2 wave + 1 of the tile: four groups of 16 pixels
      f32x4_t acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int i = 2 * wave + (g >> 1) + dy, j = 16 * (g & 1) + lp + dx;
              const h16x8_t b = *reinterpret_cast<const h16x8_t*>(pt + ui_paddr(i, j, ks * 4 + lq));
              acc[g] = rg_mfma_h16_16x16x32(af[dy * 3 + dx][ks], b, acc[g], 0, 0, 0);
            }
      // D[row = channel 4 lq + r][col = pixel lp]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oy = oy0 + 2 * wave + (g >> 1), ox = ox0 + 16 * (g & 1) + lp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * lq + r;
          if (c < Cout) y[(((size_t)n * Cout + c) * H2 + oy) * W2 + ox] = acc[g][r] + bv[r];
        }
      }
    } else {
      // ---- phase 2 (weight gradient): per output row of the tile one 32-pixel k-step per tap
      const int q4 = lp >> 2, p4 = lp & 3;            // transposed read: this lane addresses pixel q4 of its group's four,
      const int chunk = 2 * wave + (p4 >> 1);         // channels 16 wave + 4 p4 .. + 3 (8 bytes of the pixel's 128)
#pragma unroll 2
      for (int row = 0; row < UI_TR; ++row) {
        // A: gy[o = lp][row][8 lq .. + 7] (rows beyond 3 of the 16: zero)
        const int o = min(lp, 3);
        const float4 g0 = *reinterpret_cast<const float4*>(gys + (o * UI_TR + row) * UI_TC + 8 * lq);
        const float4 g1 = *reinterpret_cast<const float4*>(gys + (o * UI_TR + row) * UI_TC + 8 * lq + 4);
        const bool live = lp < 4;
        const uint4 pk = make_uint4(live ? ui_pack2(g0.x, g0.y) : 0u, live ? ui_pack2(g0.z, g0.w) : 0u,
                                    live ? ui_pack2(g1.x, g1.y) : 0u, live ? ui_pack2(g1.z, g1.w) : 0u);
        const h16x8_t a = __builtin_bit_cast(h16x8_t, pk);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            u32x2_t b0, b1;
            const int i = row + dy, j0 = 8 * lq + q4 + dx, j1 = j0 + 4;
            const unsigned a0 = (unsigned)(size_t)(lds_vptr_t)pt + (unsigned)(ui_paddr(i, j0, chunk) + (p4 & 1) * 8);
            const unsigned a1 = (unsigned)(size_t)(lds_vptr_t)pt + (unsigned)(ui_paddr(i, j1, chunk) + (p4 & 1) * 8);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0) : "v"(a0) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1) : "v"(a1) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1)::"memory");
            const uint4 bb = make_uint4(b0.x, b0.y, b1.x, b1.y);
            wacc[dy * 3 + dx] = rg_mfma_h16_16x16x32(a, __builtin_bit_cast(h16x8_t, bb), wacc[dy * 3 + dx], 0, 0, 0);
          }
      }
    }
  }
  if constexpr (MODE == 1) {
    // this workgroup's partial sums: part[block][o][ci][tap], lanes of the first 16-lane group hold o = 0 .. 3
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lq == 0 && r < Cout)
          part[(size_t)blockIdx.x * Cout * 576 + ((size_t)r * 64 + 16 * wave + lp) * 9 + tap] = wacc[tap][r];
  }
}

// dw[idx] (+)= sum over the workgroups' partials, fixed order
__global__ void upimg_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int blocks, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += part[(size_t)b * n + idx];
  dw[idx] = accumulate ? dw[idx] + s : s;
}

}  // namespace

// shapes upimg_fwd_kernel takes: 64 input channels, at most 8 output channels, whole 8 x 32 output tiles
bool rg_upimg_fwd_supported(int N, int H, int W, int Cin, int Cout) {
  return N > 0 && Cin == 64 && Cout >= 1 && Cout <= 8 && H >= 4 && W >= 16 && (2 * H) % UI_TR == 0 && (2 * W) % UI_TC == 0 &&
         (long long)N * (2 * H / UI_TR) * (2 * W / UI_TC) < 0x7fffffffLL;
}
static int upimg_blocks(int total) {
  int blocks = rg_option("upimg_blocks", 512);      // two workgroups per CU, each walks its tiles (measured: 512 < 1024 < 2048)
  if (blocks > total) blocks = total;
  return blocks < 1 ? 1 : blocks;
}
int rg_upimg_fwd(const void* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                 hipStream_t st) {
  RG_REQUIRE(rg_upimg_fwd_supported(N, H, W, Cin, Cout), RG_EUNSUPPORTED, "upimg_fwd: shape");
  const int tiles_x = 2 * W / UI_TC, tiles_y = 2 * H / UI_TR;
  const int total = N * tiles_x * tiles_y;
  hipLaunchKernelGGL(upimg_kernel<0>, dim3((unsigned)upimg_blocks(total)), dim3(256), 0, st, (const uint16_t*)x, w, bias, y,
                     (float*)nullptr, H, W, Cout, tiles_x, tiles_y, total);
  RG_LAUNCH_CHECK("upimg_fwd");
  return RG_OK;
}

// weight gradient of the same block from the image gradient gy (fp32 NCHW): at most 4 output channels (one 16-lane group of the
// accumulator holds them)
bool rg_upimg_wgrad_supported(int N, int H, int W, int Cin, int Cout) {
  return Cout <= 4 && rg_upimg_fwd_supported(N, H, W, Cin, Cout);
}
size_t rg_upimg_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  if (!rg_upimg_wgrad_supported(N, H, W, Cin, Cout)) return 0;
  const int total = N * (2 * W / UI_TC) * (2 * H / UI_TR);
  return (size_t)upimg_blocks(total) * Cout * 576 * sizeof(float);
}
int rg_upimg_wgrad(const float* gy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout, int accumulate, void* ws,
                   size_t ws_bytes, hipStream_t st) {
  RG_REQUIRE(rg_upimg_wgrad_supported(N, H, W, Cin, Cout), RG_EUNSUPPORTED, "upimg_wgrad: shape");
  RG_REQUIRE(ws && ws_bytes >= rg_upimg_wgrad_ws_bytes(N, H, W, Cin, Cout), RG_EWORKSPACE, "upimg_wgrad: workspace too small");
  const int tiles_x = 2 * W / UI_TC, tiles_y = 2 * H / UI_TR;
  const int total = N * tiles_x * tiles_y, blocks = upimg_blocks(total), n = Cout * 576;
  hipLaunchKernelGGL(upimg_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint16_t*)x, (const float*)nullptr,
                     (const float*)nullptr, const_cast<float*>(gy), (float*)ws, H, W, Cout, tiles_x, tiles_y, total);
  RG_LAUNCH_CHECK("upimg_wgrad");
  hipLaunchKernelGGL(upimg_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, dw, n, blocks,
                     accumulate);
  RG_LAUNCH_CHECK("upimg_wgrad(reduce)");
  return RG_OK;
}
