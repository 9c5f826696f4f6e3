// rg_upimg.hip -- the IMAGE block of DCGANUpGenerator (src/dcgan.py:45-56,76-84: Upsample(x2, bilinear) + ReflectionPad2d(1) +
// Conv2d(64 -> 3, 3 x 3)) without the materialised upsample + pad image: at 128 -> 256 pixels and batch 64 that
// image is 545 MB (bf16, 64 channels) written by one kernel and gathered nine times by the next, for an output of 50 MB.
// Three kernels, all per 8 x 32 output tile (= 4 x 16 low-resolution pixels) of one image, persistent workgroups of 4 waves, the
// next tile's global loads in flight (registers) during the current tile's phases, LDS-only barriers between phases:
//
// upimg_kernel<0>, forward:
//   0. the 6 x 18 low-resolution pixels the tile depends on (clamped at the image border)           global -> LDS, 13.5 KB
//   1. P = pad(upsample(x)) for the tile's 10 x 34 padded positions: bilinear weights 0.25 / 0.75 in fp32 from four staged
//      pixels, rounded ONCE to the 16-bit type (the arithmetic and the rounding of uppad_bf16_kernel: the results of the two
//      paths are bit-identical)                                                                         LDS -> LDS, 42.5 KB
//   2. nine taps x two 32-channel steps of v_mfma_f32_16x16x32: A = the weights (row = output channel, 3 of 16 live; the 18
//      fragments stay in registers for the whole launch), B = 16 consecutive pixels of a P row at the tap's offset (one
//      ds_read_b128 per MFMA; the channel chunk of a pixel is XOR-swizzled with bits 1..3 of its column so that the 16 lanes
//      of a read hit 16 different bank groups), D = [channel][pixel]: lanes 0..15 hold a pixel's three channels -> fp32 NCHW rows.
//   HBM traffic: x once (+ the tiles' halo, from L2) and y once.  Batch 64 at 128 -> 256: 228 us against 633 us for uppad + 9-tap
//   GEMM + NCHW pass (what is left: phase 1's vector arithmetic, ~110 us, the 64-byte output segments, the tile prologues).
// upimg_kernel<1>, weight gradient (368 us against 1562): phases 0 / 1 as above, phase 2 contracts the tile over its pixels.
// upimg_bwd_kernel, data gradient (477 us against 1083): transposed conv into LDS, then the adjoint of pad o upsample.
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
        // all nine taps' transposed reads go out before the first wait (18 reads in flight; waited for once per row)
        u32x2_t b0[9], b1[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int i = row + dy, j0 = 8 * lq + q4 + dx, j1 = j0 + 4;
            const unsigned a0 = (unsigned)(size_t)(lds_vptr_t)pt + (unsigned)(ui_paddr(i, j0, chunk) + (p4 & 1) * 8);
            const unsigned a1 = (unsigned)(size_t)(lds_vptr_t)pt + (unsigned)(ui_paddr(i, j1, chunk) + (p4 & 1) * 8);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[dy * 3 + dx]) : "v"(a0) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1[dy * 3 + dx]) : "v"(a1) : "memory");
          }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2]), "+v"(b0[3]), "+v"(b1[3]),
                       "+v"(b0[4]), "+v"(b1[4])::"memory");
        asm volatile("" : "+v"(b0[5]), "+v"(b1[5]), "+v"(b0[6]), "+v"(b1[6]), "+v"(b0[7]), "+v"(b1[7]), "+v"(b0[8]), "+v"(b1[8]));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const uint4 bb = make_uint4(b0[tap].x, b0[tap].y, b1[tap].x, b1[tap].y);
          wacc[tap] = rg_mfma_h16_16x16x32(a, __builtin_bit_cast(h16x8_t, bb), wacc[tap], 0, 0, 0);
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

// ---- data gradient of the same block: gx[n][h][w][64] (16-bit NHWC) from the image gradient gy (fp32 NCHW), per low-resolution
// tile of 4 x 16 pixels (= the forward's 8 x 32 output tile):
//   0. the 12 x 36 window of gy the tile's padded positions depend on (zero outside the image)          global -> LDS, 6.8 KB
//   1. gP = transposed 3 x 3 conv at the 10 x 34 padded positions: one k-step of v_mfma_f32_16x16x32 per 16 positions and 16
//      channels (k = (output channel, tap): 27 of 32 live), A = the weights, B = eight gy values per lane gathered from the
//      window; rounded to the 16-bit type (as the matrix-core path of the other blocks rounds its padded-grid gradient)  -> LDS
//   2. the adjoint of (reflection pad o bilinear x2): every low-resolution pixel collects its up to 16 upsampled positions
//      (each with the padded positions that mirror it at the image border) with the weights 0.25 / 0.75 squared      LDS -> global
constexpr int UB_GR = UI_PR + 2, UB_GC = UI_PC + 2;             // gy window: 12 x 36
constexpr int UB_GITEMS = 4 * UB_GR * UB_GC;                    // floats, four channel planes: 1728
constexpr int UB_NONE = 1 << 30;                                // "this k is padding" (valid offsets are small, of either sign)

__global__ __launch_bounds__(256, 2) void upimg_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           uint16_t* __restrict__ gx, int H, int W, int Cout, int tiles_x,
                                                           int tiles_y, int total) {
  __shared__ __attribute__((aligned(16))) float gyw[UB_GITEMS];
  __shared__ __attribute__((aligned(16))) unsigned char gpt[UI_PITEMS * 16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lp = lane & 15, lq = lane >> 4;
  const int H2 = 2 * H, W2 = 2 * W;

  // A operand of phase 1: row = input channel 16 ct + lp, k = 8 lq + e = (o, dy, dx) flattened o * 9 + dy * 3 + dx
  h16x8_t aw[4];
  int koff[8];                                        // gy-window offset of k relative to (padded row + 2, padded column + 2)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * lq + e, o = k / 9, tt = k - 9 * o, dy = tt / 3, dx = tt - 3 * dy;
    koff[e] = (k < 27 && o < Cout) ? (o * UB_GR - dy) * UB_GC - dx : UB_NONE;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * lq + e, o = k / 9, tt = k - 9 * o;
      v[e] = (k < 27 && o < Cout) ? w[((size_t)o * 64 + 16 * ct + lp) * 9 + tt] : 0.f;
    }
    const uint4 pk = make_uint4(ui_pack2(v[0], v[1]), ui_pack2(v[2], v[3]), ui_pack2(v[4], v[5]), ui_pack2(v[6], v[7]));
    aw[ct] = __builtin_bit_cast(h16x8_t, pk);
  }

  // the gy window of a tile, one tile ahead in registers (seven floats per thread; clamped addresses, zero outside the image)
  float g0, g1, g2, g3, g4, g5, g6;
  g0 = g1 = g2 = g3 = g4 = g5 = g6 = 0.f;
#define UB_FETCH1(dst, k, n_, oyb_, oxb_)                                                          \
  do {                                                                                             \
    const int e_ = min(t + 256 * (k), UB_GITEMS - 1);                                              \
    const int o_ = e_ / (UB_GR * UB_GC), rem_ = e_ - o_ * (UB_GR * UB_GC);                         \
    const int r_ = rem_ / UB_GC, c_ = rem_ - r_ * UB_GC;                                           \
    const int oy_ = (oyb_) + r_, ox_ = (oxb_) + c_;                                                \
    const bool ok_ = o_ < Cout && (unsigned)oy_ < (unsigned)H2 && (unsigned)ox_ < (unsigned)W2;    \
    const float v_ = gy[(((size_t)(n_) * Cout + min(o_, Cout - 1)) * H2 + min(max(oy_, 0), H2 - 1)) * W2 + min(max(ox_, 0), W2 - 1)]; \
    dst = ok_ ? v_ : 0.f;                                                                          \
  } while (0)
#define UB_FETCH(tile_)                                                                            \
  do {                                                                                             \
    const int tx_ = (tile_) % tiles_x, ty_ = ((tile_) / tiles_x) % tiles_y, n_ = (tile_) / (tiles_x * tiles_y);   \
    const int oyb_ = ty_ * UI_TR - 2, oxb_ = tx_ * UI_TC - 2;                                      \
    UB_FETCH1(g0, 0, n_, oyb_, oxb_); UB_FETCH1(g1, 1, n_, oyb_, oxb_); UB_FETCH1(g2, 2, n_, oyb_, oxb_);          \
    UB_FETCH1(g3, 3, n_, oyb_, oxb_); UB_FETCH1(g4, 4, n_, oyb_, oxb_); UB_FETCH1(g5, 5, n_, oyb_, oxb_);          \
    UB_FETCH1(g6, 6, n_, oyb_, oxb_);                                                              \
  } while (0)
  if ((int)blockIdx.x < total) UB_FETCH((int)blockIdx.x);

  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int oy0 = ty * UI_TR, ox0 = tx * UI_TC;      // padded coordinates of the window's first row / column
    const int hb = ty * (UI_TR / 2), wb = tx * (UI_TC / 2);

    // ---- phase 0 (the previous tile's phase 2 reads gpt only; its phase 1, the reader of gyw, is behind that tile's barrier)
    gyw[t] = g0; gyw[t + 256] = g1; gyw[t + 512] = g2; gyw[t + 768] = g3; gyw[t + 1024] = g4; gyw[t + 1280] = g5;
    if (t + 1536 < UB_GITEMS) gyw[t + 1536] = g6;
    UI_BARRIER();          // (also: every wave is done with phase 2 of the previous tile before gpt is overwritten)
    if (tile + (int)gridDim.x < total) UB_FETCH(tile + (int)gridDim.x);

    // ---- phase 1: gP for 16 padded positions per MFMA column group; wave w takes groups w, w + 4, ...
#pragma unroll 1
    for (int g = wave; g < (UI_PR * UI_PC + 15) / 16; g += 4) {
      const int p = min(16 * g + lp, UI_PR * UI_PC - 1);        // (the last group's spare lanes repeat position 339)
      const int il = p / UI_PC, jl = p - il * UI_PC;
      const int base = (il + 2) * UB_GC + jl + 2;
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = gyw[koff[e] == UB_NONE ? 0 : base + koff[e]];
        bv[e] = koff[e] == UB_NONE ? 0.f : v;
      }
      const uint4 pk = make_uint4(ui_pack2(bv[0], bv[1]), ui_pack2(bv[2], bv[3]), ui_pack2(bv[4], bv[5]), ui_pack2(bv[6], bv[7]));
      const h16x8_t b = __builtin_bit_cast(h16x8_t, pk);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const f32x4_t d = rg_mfma_h16_16x16x32(aw[ct], b, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // D[row = channel 16 ct + 4 lq + r][col = position lp]: four consecutive channels of one position = 8 bytes
        *reinterpret_cast<uint2*>(gpt + ui_paddr(il, jl, 2 * ct + (lq >> 1)) + (lq & 1) * 8) =
            make_uint2(ui_pack2(d[0], d[1]), ui_pack2(d[2], d[3]));
      }
    }
    UI_BARRIER();

    // ---- phase 2: adjoint of pad o upsample; a thread owns (low-resolution pixel, 8-channel chunk) pairs
#pragma unroll 1
    for (int it = t; it < (UI_TR / 2) * (UI_TC / 2) * 8; it += 256) {
      const int ch = it & 7, px = it >> 3;
      const int hl = px / (UI_TC / 2), wl = px - hl * (UI_TC / 2);
      const int h = hb + hl, wq = wb + wl;
      // per direction: up to four upsampled positions with their weight, each read at up to two padded positions
      float cu[4], cv[4];
      int iu[4][2], jv[4][2];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int u = 2 * h + d - 1, v = 2 * wq + d - 1;
        int a0, a1, b0, b1;
        float la, lb;
        up_taps(min(max(u, 0), H2 - 1), H, a0, a1, la);
        up_taps(min(max(v, 0), W2 - 1), W, b0, b1, lb);
        const bool uok = u >= 0 && u < H2, vok = v >= 0 && v < W2;
        cu[d] = uok ? (a0 == h ? 1.f - la : 0.f) + (a1 == h ? la : 0.f) : 0.f;
        cv[d] = vok ? (b0 == wq ? 1.f - lb : 0.f) + (b1 == wq ? lb : 0.f) : 0.f;
        iu[d][0] = u + 1 - oy0;                                         // padded row u + 1 ...
        iu[d][1] = u == 1 ? 0 - oy0 : (u == H2 - 2 ? H2 + 1 - oy0 : -1);   // ... and the mirrored border row that reads u
        jv[d][0] = v + 1 - ox0;
        jv[d][1] = v == 1 ? 0 - ox0 : (v == W2 - 2 ? W2 + 1 - ox0 : -1);
      }
      f32x2_t acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = f32x2_t{0.f, 0.f};
#pragma unroll
      for (int du = 0; du < 4; ++du)
#pragma unroll
        for (int dv = 0; dv < 4; ++dv) {
          const float f = cu[du] * cv[dv];
          if (f == 0.f) continue;
          f32x2_t sacc[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[e] = f32x2_t{0.f, 0.f};
#pragma unroll
          for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int rj = 0; rj < 2; ++rj) {
              const int i = iu[du][ri], j = jv[dv][rj];
              if (i < 0 || j < 0) continue;
              const uint4 q = *reinterpret_cast<const uint4*>(gpt + ui_paddr(i, j, ch));
              const uint32_t dd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) sacc[e] += f32x2_t{h16lo_to_f32(dd[e]), h16hi_to_f32(dd[e])};
            }
          const f32x2_t vf = {f, f};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += vf * sacc[e];
        }
      *reinterpret_cast<uint4*>(gx + (((size_t)n * H + h) * W + wq) * 64 + ch * 8) =
          make_uint4(ui_pack2(acc[0][0], acc[0][1]), ui_pack2(acc[1][0], acc[1][1]), ui_pack2(acc[2][0], acc[2][1]),
                     ui_pack2(acc[3][0], acc[3][1]));
    }
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

// data gradient of the same block (gy fp32 NCHW -> gx 16-bit NHWC)
bool rg_upimg_bwd_supported(int N, int H, int W, int Cin, int Cout) {
  return Cout <= 3 && rg_upimg_fwd_supported(N, H, W, Cin, Cout);       // k = 9 Cout <= 32: one MFMA k-step
}
int rg_upimg_bwd_data(const float* gy, const float* w, void* gx, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  RG_REQUIRE(rg_upimg_bwd_supported(N, H, W, Cin, Cout), RG_EUNSUPPORTED, "upimg_bwd_data: shape");
  const int tiles_x = 2 * W / UI_TC, tiles_y = 2 * H / UI_TR;
  const int total = N * tiles_x * tiles_y;
  hipLaunchKernelGGL(upimg_bwd_kernel, dim3((unsigned)upimg_blocks(total)), dim3(256), 0, st, gy, w, (uint16_t*)gx, H, W, Cout,
                     tiles_x, tiles_y, total);
  RG_LAUNCH_CHECK("upimg_bwd_data");
  return RG_OK;
}
