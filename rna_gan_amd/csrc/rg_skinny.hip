// rg_skinny.hip -- the two image-side layers (3 <-> 64 channels at 256x256).  K = 48 (or 3 output
// channels) cannot fill an MFMA tile; these layers are HBM-bound (SURVEY 7 hard part 3), so they
// run on the vector ALUs with scalar-loaded weights and coalesced image-side accesses.
//   first_down : NCHW fp32 image  -> NHWC T, 4x4 s2 p1 conv (+bias, LeakyReLU)
//   last_up    : NHWC T           -> NCHW fp32 image, transposed conv (+bias, tanh)
//   wgrad      : dW[o][3][16] = sum_pix low[pix][o] * patch(pix)
#include "rg_internal.h"
#include <stdlib.h>

namespace {

constexpr int SK_I = 3;
constexpr int SK_K = SK_I * 16;   // 48

// ---------------------------------------------------------------------------------------------
// The image-side layers use the fp32 master weights directly (no bf16 rounding): they are read with
// UNIFORM indices, so hipcc emits scalar loads (s_load_dwordx*) and the FMAs take the weight as an
// SGPR operand -- no LDS or vector-memory traffic for weights.

// first_down: one thread = one output pixel, all O channels in OC-wide register chunks.
// x NCHW fp32; lanes run along wo so every patch load instruction covers a 512-byte span.
template <typename T, int OC>
__global__ __launch_bounds__(256) void first_down_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                         const float* __restrict__ bias, T* __restrict__ y, int N,
                                                         int H, int W, int O, float slope) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long long npix = (long long)N * Ho * Wo;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int wo = (int)(p % Wo);
  const long long tq = p / Wo;
  const int ho = (int)(tq % Ho), n = (int)(tq / Ho);
  float patch[SK_K];
#pragma unroll
  for (int ci = 0; ci < SK_I; ++ci)
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
        bool v = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        patch[ci * 16 + kh * 4 + kw] = v ? x[(((long long)n * SK_I + ci) * H + hi) * W + wi] : 0.f;
      }
  for (int oc = 0; oc < O; oc += OC) {
    float acc[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) {
      const float* wr = wq + (size_t)(oc + j) * SK_K;     // uniform -> scalar loads
      float a = bias ? bias[oc + j] : 0.f;
#pragma unroll
      for (int k = 0; k < SK_K; ++k) a = fmaf(patch[k], wr[k], a);
      acc[j] = lrelu_f(a, slope);
    }
    T* yo = y + p * O + oc;
#pragma unroll
    for (int q = 0; q < OC / 4; ++q) Vec<T, 4>::st(yo + q * 4, acc + q * 4);
  }
}

// ---------------------------------------------------------------------------------------------
// first_down on the matrix cores (bf16 output).  Roles are swapped so that the PIXEL is the MFMA
// column (lane) index:  D[ch][pix] = sum_k W[ch][k] * P[k][pix],  k = ci*16 + tap (48, one 16-wide k-step
// per input channel).  Per 32-pixel group a lane (pixel r = lane&31, half h = lane>>5) builds the B-operand
// fragment of channel ci = the 8 taps kh in {2h, 2h+1} x kw 0..3 of its pixel directly from two NCHW
// image rows (lanes run along wo: every load instruction covers a 256-byte span); A = weights, 6 fragments
// kept in registers.  The 32x32 result has the pixel on the lane and 16 channels in registers;
// v_permlane32_swap pairs the two half-waves so that every store is 16 bytes of consecutive NHWC channels.
typedef __attribute__((ext_vector_type(8))) __bf16 sk_bf16x8;
typedef __attribute__((ext_vector_type(16))) float sk_f32x16;

__device__ __forceinline__ uint32_t sk_pack2(float a, float b) {
  return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}

__global__ __launch_bounds__(256) void first_down_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                              int N, int H, int W, float slope, int ngroups) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Ho = H >> 1, Wo = W >> 1;
  const int gpr = Wo >> 5;                       // 32-pixel groups per output row
  // A fragments: W[ch = 32*i + r][k = ci*16 + 8h .. +7]  (torch layout w[o][ci][16 taps] is k-contiguous)
  sk_bf16x8 wa[2][SK_I];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ci = 0; ci < SK_I; ++ci) {
      const float4* wp = reinterpret_cast<const float4*>(w + ((size_t)(32 * i + r) * SK_I + ci) * 16 + 8 * h);
      float4 a = wp[0], b = wp[1];
      uint4 v = make_uint4(sk_pack2(a.x, a.y), sk_pack2(a.z, a.w), sk_pack2(b.x, b.y), sk_pack2(b.z, b.w));
      wa[i][ci] = __builtin_bit_cast(sk_bf16x8, v);
    }
  // bias of the 16 channels this lane ends up holding: ch = 32*i + 8*g + 4*h + e
  float bs[2][4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int e = 0; e < 4; ++e) bs[i][g4][e] = bias ? bias[32 * i + 8 * g4 + 4 * h + e] : 0.f;

  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int wo = (grp % gpr) * 32 + r;
    const int row = grp / gpr;
    const int ho = row % Ho, n = row / Ho;
    sk_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int wi0 = 2 * wo - 1;
#pragma unroll
    for (int ci = 0; ci < SK_I; ++ci) {
      float pv[8];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int hi = 2 * ho - 1 + 2 * h + a;
        const bool vh = (unsigned)hi < (unsigned)H;
        const float* xr = x + (((size_t)n * SK_I + ci) * H + (vh ? hi : 0)) * W;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const int wi = wi0 + kw;
          const bool v = vh && (unsigned)wi < (unsigned)W;
          pv[a * 4 + kw] = v ? xr[wi] : 0.f;
        }
      }
      uint4 pk = make_uint4(sk_pack2(pv[0], pv[1]), sk_pack2(pv[2], pv[3]), sk_pack2(pv[4], pv[5]),
                            sk_pack2(pv[6], pv[7]));
      sk_bf16x8 pb = __builtin_bit_cast(sk_bf16x8, pk);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][ci], pb, acc[i], 0, 0, 0);
    }
    // acc[i][4*g + e] = D[ch = 32*i + 8*g + 4*h + e][pixel r]
    uint16_t* yo = y + (((size_t)n * Ho + ho) * Wo + wo) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint32_t q[4][2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float v0 = lrelu_f(acc[i][4 * g4 + 0] + bs[i][g4][0], slope), v1 = lrelu_f(acc[i][4 * g4 + 1] + bs[i][g4][1], slope);
        float v2 = lrelu_f(acc[i][4 * g4 + 2] + bs[i][g4][2], slope), v3 = lrelu_f(acc[i][4 * g4 + 3] + bs[i][g4][3], slope);
        q[g4][0] = sk_pack2(v0, v1);
        q[g4][1] = sk_pack2(v2, v3);
      }
      // pair groups (0,1) and (2,3): afterwards the lower half-wave holds [own g | upper's g] = 8 consecutive
      // channels 8g..8g+7 and the upper half-wave [lower's g+1 | own g+1] = channels 8(g+1)..8(g+1)+7
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int ga = 2 * pr, gb = 2 * pr + 1;
        auto s0 = __builtin_amdgcn_permlane32_swap(q[ga][0], q[gb][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(q[ga][1], q[gb][1], false, false);
        uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        const int ch = 32 * i + 8 * (ga + h);
        *reinterpret_cast<uint4*>(yo + ch) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// skinny weight gradient on the matrix cores (bf16 `low`):  dW[o][k] = sum_pix low[pix][o] * P[pix][k].
// The contraction index is the pixel, which is the slow axis of both operands, so per 32-pixel group a
// wave (a) builds the patch values exactly like first_down_mfma (pixel on the lane, 8 taps per lane and
// input channel, coalesced NCHW row reads) and writes them as [pixel][k] rows into its PRIVATE LDS
// region, (b) copies its [32 pixel][64 ch] tile of `low` (4 KB, 16-byte coalesced loads) next to it, and
// (c) reads both back pixel-contiguous with ds_read_b64_tr_b16 as MFMA A/B fragments (2 k-steps x 4
// MFMAs).  Rows are 192 bytes apart: the 4 rows x 64 bytes a half-wave touches per transposed read fall
// into 4 disjoint bank ranges.  No block barrier inside the loop; the 4 waves' accumulators are summed
// through LDS at the end and written as one fp32 slab per block (reduced deterministically afterwards).
typedef __attribute__((ext_vector_type(4))) short sk_s16x4;
typedef __attribute__((ext_vector_type(8))) short sk_s16x8;
constexpr int SKW_ROW = 96;     // uint16 elements per LDS row (192 bytes)

__device__ __forceinline__ sk_s16x4 sk_tr_read(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sk_s16x4*)(p));
}

__global__ __launch_bounds__(256) void skinny_wgrad_mfma_kernel(const uint16_t* __restrict__ low,
                                                                const float* __restrict__ x, float* __restrict__ slab,
                                                                int N, int H, int W, int ngroups) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4 * 2 * 32 * SKW_ROW];     // 48 KB: per wave [patch | low]
  __shared__ float red[64 * SK_K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Ho = H >> 1, Wo = W >> 1;
  const int gpr = Wo >> 5;
  uint16_t* pt = lds + wave * (2 * 32 * SKW_ROW);      // patch tile  [32 pix][96] (k 0..47 valid, 48..63 zero)
  uint16_t* lt = pt + 32 * SKW_ROW;                    // low tile    [32 pix][96] (ch 0..63)
  // zero the k-padding once (columns 48..63 feed output columns that are never written out)
  *reinterpret_cast<uint4*>(pt + r * SKW_ROW + 48 + 8 * h) = make_uint4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < 64 * SK_K; i += 256) red[i] = 0.f;

  sk_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int grp16 = lane >> 4, idx = lane & 15, q = idx >> 2, p4 = idx & 3, fh = grp16 >> 1, cb = grp16 & 1;
  // contiguous range of 32-pixel groups per wave (sequential pixels: L2-friendly)
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int per = (ngroups + nw - 1) / nw;
  const int g0 = wid * per, g1 = min(ngroups, g0 + per);
  for (int grp = g0; grp < g1; ++grp) {
    const int wo = (grp % gpr) * 32 + r;
    const int row = grp / gpr;
    const int ho = row % Ho, n = row / Ho;
    const int wi0 = 2 * wo - 1;
    // (a) patch rows: lane (pixel r, half h) writes taps 8h..8h+7 of each input channel
#pragma unroll
    for (int ci = 0; ci < SK_I; ++ci) {
      float pv[8];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int hi = 2 * ho - 1 + 2 * h + a;
        const bool vh = (unsigned)hi < (unsigned)H;
        const float* xr = x + (((size_t)n * SK_I + ci) * H + (vh ? hi : 0)) * W;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const int wi = wi0 + kw;
          pv[a * 4 + kw] = (vh && (unsigned)wi < (unsigned)W) ? xr[wi] : 0.f;
        }
      }
      *reinterpret_cast<uint4*>(pt + r * SKW_ROW + ci * 16 + 8 * h) =
          make_uint4(sk_pack2(pv[0], pv[1]), sk_pack2(pv[2], pv[3]), sk_pack2(pv[4], pv[5]), sk_pack2(pv[6], pv[7]));
    }
    // (b) low tile: 32 pixels x 128 bytes, 4 x 16 bytes per lane, fully coalesced
    const uint16_t* lsrc = low + (size_t)grp * 32 * 64;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = p * 64 + lane, px = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(lt + px * SKW_ROW + ch * 8) = *reinterpret_cast<const uint4*>(lsrc + px * 64 + ch * 8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private LDS: in-order DS pipe, no barrier needed
    // (c) 2 k-steps of 16 pixels: A = low^T (rows = o), B = patch (cols = k)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int prow = ks * 16 + 8 * fh + q;
      sk_bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = 32 * i + 16 * cb + 4 * p4;
        sk_s16x4 lo = sk_tr_read(lt + prow * SKW_ROW + col), hi = sk_tr_read(lt + (prow + 4) * SKW_ROW + col);
        fa[i] = __builtin_bit_cast(sk_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        lo = sk_tr_read(pt + prow * SKW_ROW + col); hi = sk_tr_read(pt + (prow + 4) * SKW_ROW + col);
        fb[i] = __builtin_bit_cast(sk_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  // block reduction of the 4 waves: acc[i][j][reg] = D[o = 32i + (reg&3)+8(reg>>2)+4h][k = 32j + r]
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = 32 * j + r;
          if (k < SK_K) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * SK_K + k] += acc[i][j][e];
          }
        }
    }
    __syncthreads();
  }
  float* sl = slab + (size_t)blockIdx.x * 64 * SK_K;
  for (int i = threadIdx.x; i < 64 * SK_K; i += 256) sl[i] = red[i];
}

// ---------------------------------------------------------------------------------------------
// last_up: block = 8 x 32 low-res positions; the (8+2) x (32+2) halo tile of the NHWC input is staged
// once in LDS (16-byte coalesced loads, rows padded by 8 bytes -> conflict-free 8-byte reads with one
// lane per pixel); each thread produces its 2x2 output quad x 3 channels and stores float2 pairs that
// are contiguous across the wave (NCHW rows).
constexpr int LU_TW = 32;
template <typename T, int LU_TH>
__global__ __launch_bounds__(LU_TH * 32) void last_up_kernel(const T* __restrict__ x, const float* __restrict__ wq,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N,
                                                      int Ho, int Wo, int O, int apply_tanh) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int rowb = O * (int)sizeof(T) + 8;                 // bytes per staged pixel (padded)
  const int t = threadIdx.x;
  const int tiles_w = (Wo + LU_TW - 1) / LU_TW, tiles_h = (Ho + LU_TH - 1) / LU_TH;
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w;
  const int th = b % tiles_h;
  const int n = b / tiles_h;
  const int h0 = th * LU_TH - 1, w0 = tw * LU_TW - 1;      // halo origin
  const T* xb = x + (long long)n * Ho * Wo * O;
  // stage: (LU_TH+2)*(LU_TW+2) pixels x O channels, 16 bytes per lane
  const int chunks = O * (int)sizeof(T) / 16;
  const int npx = (LU_TH + 2) * (LU_TW + 2);
  for (int i = t; i < npx * chunks; i += LU_TH * 32) {
    int px = i / chunks, ch = i - px * chunks;
    int hh = h0 + px / (LU_TW + 2), ww = w0 + px % (LU_TW + 2);
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((unsigned)hh < (unsigned)Ho && (unsigned)ww < (unsigned)Wo)
      v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(xb + ((long long)hh * Wo + ww) * O) + ch * 16);
    // 8-byte stores keep the padded rows 8-byte aligned
    uint2* d = reinterpret_cast<uint2*>(smem + px * rowb + ch * 16);
    d[0] = make_uint2(v.x, v.y);
    d[1] = make_uint2(v.z, v.w);
  }
  __syncthreads();
  const int lh = t / LU_TW, lw = t % LU_TW;
  const int hq = th * LU_TH + lh, wq_ = tw * LU_TW + lw;
  float acc[2][2][SK_I];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < SK_I; ++i) acc[a][c][i] = bias ? bias[i] : 0.f;
  for (int o0 = 0; o0 < O; o0 += 4) {
    float xv[3][3][4];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const unsigned char* sp = smem + ((lh + dh) * (LU_TW + 2) + (lw + dw)) * rowb + o0 * (int)sizeof(T);
        Vec<T, 4>::ld(reinterpret_cast<const T*>(sp), xv[dh][dw]);
      }
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const float* wr = wq + (size_t)(o0 + oo) * SK_K;      // uniform -> scalar loads
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int kh = ph == 0 ? (a == 0 ? 1 : 3) : (a == 0 ? 0 : 2);
              const int dh = ph == 0 ? (a == 0 ? 1 : 0) : (a == 0 ? 2 : 1);
              const int kw = pw == 0 ? (c == 0 ? 1 : 3) : (c == 0 ? 0 : 2);
              const int dw = pw == 0 ? (c == 0 ? 1 : 0) : (c == 0 ? 2 : 1);
              const float xval = xv[dh][dw][oo];
#pragma unroll
              for (int i = 0; i < SK_I; ++i)
                acc[ph][pw][i] = fmaf(xval, wr[i * 16 + kh * 4 + kw], acc[ph][pw][i]);
            }
    }
  }
  if (hq >= Ho || wq_ >= Wo) return;
  const int H = 2 * Ho, W = 2 * Wo;
#pragma unroll
  for (int i = 0; i < SK_I; ++i)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      float v0 = acc[ph][0][i], v1 = acc[ph][1][i];
      if (apply_tanh) { v0 = tanhf(v0); v1 = tanhf(v1); }
      float2* dst = reinterpret_cast<float2*>(y + (((long long)n * SK_I + i) * H + 2 * hq + ph) * W + 2 * wq_);
      *dst = make_float2(v0, v1);
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad: block = contiguous chunk of pixels, processed in 64-pixel tiles staged in LDS as fp32.
// thread = (pixel quarter ps, o-group og of 4 channels, k-group kg of 12 taps): 48 accumulators,
// per pixel 1 + 3 16-byte LDS reads for 48 FMAs.  The 4 pixel quarters are summed through LDS.
template <typename T>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const T* __restrict__ low, const float* __restrict__ high,
                                                           float* __restrict__ slab, int N, int Ho, int Wo, int O,
                                                           int pix_per_block) {
  __shared__ __attribute__((aligned(16))) float lo_s[64][64 + 4];
  __shared__ __attribute__((aligned(16))) float pa_s[64][SK_K + 4];
  __shared__ __attribute__((aligned(16))) float red[4][64 * SK_K];
  const int t = threadIdx.x;
  const int ps = t >> 6;                 // pixel quarter: pixels ps*16 .. ps*16+15 of each tile
  const int og = (t & 63) >> 2;          // 16 groups of 4 output channels
  const int kg = t & 3;                  // 4 groups of 12 taps
  const int H = 2 * Ho, W = 2 * Wo;
  const long long npix = (long long)N * Ho * Wo;
  const long long pb = (long long)blockIdx.x * pix_per_block;
  long long pe = pb + pix_per_block;
  if (pe > npix) pe = npix;
  for (int oc0 = 0; oc0 < O; oc0 += 64) {
    float acc[4][12];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[a][j] = 0.f;
    for (long long p0 = pb; p0 < pe; p0 += 64) {
      // stage low[64 pix][64 o] (4 consecutive channels per lane) and the 48-value patches
      for (int i = t; i < 64 * 16; i += 256) {
        int pr = i >> 4, c4 = (i & 15) * 4;
        long long p = p0 + pr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (p < pe) Vec<T, 4>::ld(low + p * O + oc0 + c4, v);
        *reinterpret_cast<float4*>(&lo_s[pr][c4]) = make_float4(v[0], v[1], v[2], v[3]);
      }
      for (int i = t; i < 64 * SK_K; i += 256) {
        int k = i >> 6, pr = i & 63;     // lanes run along pixels: NCHW reads are stride-2 contiguous
        long long p = p0 + pr;
        float v = 0.f;
        if (p < pe) {
          int wo = (int)(p % Wo);
          long long tq = p / Wo;
          int ho = (int)(tq % Ho), n = (int)(tq / Ho);
          int ci = k >> 4, kh = (k >> 2) & 3, kw = k & 3;
          int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
          if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
            v = high[(((long long)n * SK_I + ci) * H + hi) * W + wi];
        }
        pa_s[pr][k] = v;
      }
      __syncthreads();
#pragma unroll 4
      for (int pp = 0; pp < 16; ++pp) {
        const int pr = ps * 16 + pp;
        float4 a4 = *reinterpret_cast<const float4*>(&lo_s[pr][og * 4]);
        const float4* pk = reinterpret_cast<const float4*>(&pa_s[pr][kg * 12]);
        float4 k0 = pk[0], k1 = pk[1], k2 = pk[2];
        float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float kv[12] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x, k2.y, k2.z, k2.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int j = 0; j < 12; ++j) acc[a][j] = fmaf(av[a], kv[j], acc[a][j]);
      }
      __syncthreads();
    }
    // sum the 4 pixel quarters, write slab[block][o][k]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 12; ++j) red[ps][(og * 4 + a) * SK_K + kg * 12 + j] = acc[a][j];
    __syncthreads();
    float* sl = slab + (long long)blockIdx.x * O * SK_K + (long long)oc0 * SK_K;
    for (int i = t; i < 64 * SK_K; i += 256) sl[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    __syncthreads();
  }
}

int skinny_wgrad_blocks(long long npix, int* ppb) {
  long long want = 512;
  long long per = (npix + want - 1) / want;
  per = (per + 63) / 64 * 64;
  if (per < 64) per = 64;
  *ppb = (int)per;
  return (int)((npix + per - 1) / per);
}

}  // namespace

bool rg_skinny_supported(int I, int O) { return I == SK_I && O % 64 == 0 && O <= 128; }

int rg_skinny_first_down(const float* x, const float* w, const float* bias, void* y, int N, int H, int W, int I,
                         int O, float slope, int dtype, hipStream_t st) {
  (void)I;
  long long npix = (long long)N * (H / 2) * (W / 2);
  static int no_mfma = -1;
  if (no_mfma < 0) { const char* e = getenv("RNAGAN_SKINNY_VALU"); no_mfma = (e && e[0] == '1') ? 1 : 0; }
  if (dtype == RG_BF16 && O == 64 && (W / 2) % 32 == 0 && npix / 32 < 0x7fffffff && !no_mfma) {
    int ngroups = (int)(npix / 32);
    int blocks = (ngroups + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(first_down_mfma_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, (uint16_t*)y, N, H, W, slope,
                       ngroups);
    RG_LAUNCH_CHECK("first_down(mfma)");
    return RG_OK;
  }
  unsigned blocks = (unsigned)((npix + 255) / 256);
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((first_down_kernel<T, 16>), dim3(blocks), dim3(256), 0, st, x, w, bias, (T*)y, N, H, W, O, slope);
    RG_LAUNCH_CHECK("first_down");
    return RG_OK;
  })
}

int rg_skinny_last_up(const void* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo, int O, int I,
                      int apply_tanh, int dtype, hipStream_t st) {
  (void)I;
  if (dtype == RG_BF16) {
    constexpr int TH = 8;
    int tiles = ((Wo + LU_TW - 1) / LU_TW) * ((Ho + TH - 1) / TH);
    size_t sh = (size_t)(TH + 2) * (LU_TW + 2) * (O * 2 + 8);
    hipLaunchKernelGGL((last_up_kernel<bf16_t, TH>), dim3((unsigned)((long long)N * tiles)), dim3(TH * 32), sh, st,
                       (const bf16_t*)x, w, bias, y, N, Ho, Wo, O, apply_tanh);
  } else if (dtype == RG_F32) {
    constexpr int TH = 2;
    int tiles = ((Wo + LU_TW - 1) / LU_TW) * ((Ho + TH - 1) / TH);
    size_t sh = (size_t)(TH + 2) * (LU_TW + 2) * (O * 4 + 8);
    hipLaunchKernelGGL((last_up_kernel<float, TH>), dim3((unsigned)((long long)N * tiles)), dim3(TH * 32), sh, st,
                       (const float*)x, w, bias, y, N, Ho, Wo, O, apply_tanh);
  } else {
    rg_set_error("bad dtype %d", dtype);
    return RG_EINVAL;
  }
  RG_LAUNCH_CHECK("last_up");
  return RG_OK;
}

constexpr int SKW_BLOCKS = 1024;

size_t rg_skinny_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I) {
  (void)I;
  int ppb;
  int nb = skinny_wgrad_blocks((long long)N * Ho * Wo, &ppb);
  if (nb < SKW_BLOCKS) nb = SKW_BLOCKS;
  return (size_t)nb * O * SK_K * sizeof(float);
}

int rg_skinny_wgrad_impl(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                         int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  (void)I;
  size_t elems = (size_t)O * SK_K;
  long long npix = (long long)N * Ho * Wo;
  static int no_mfma = -1;
  if (no_mfma < 0) { const char* e = getenv("RNAGAN_SKINNY_VALU"); no_mfma = (e && e[0] == '1') ? 1 : 0; }
  if (dtype == RG_BF16 && O == 64 && Wo % 32 == 0 && npix / 32 < 0x7fffffff && !no_mfma) {
    int ngroups = (int)(npix / 32);
    int nbm = (ngroups + 3) / 4;
    if (nbm > SKW_BLOCKS) nbm = SKW_BLOCKS;
    RG_REQUIRE(ws && ws_bytes >= (size_t)nbm * elems * sizeof(float), RG_EWORKSPACE, "skinny_wgrad: workspace too small");
    hipLaunchKernelGGL(skinny_wgrad_mfma_kernel, dim3(nbm), dim3(256), 0, st, (const uint16_t*)low, high_nchw, (float*)ws,
                       N, 2 * Ho, 2 * Wo, ngroups);
    RG_LAUNCH_CHECK("skinny_wgrad(mfma)");
    return rg_reduce_slabs((const float*)ws, dw, elems, nbm, accumulate, 0, 0, st);
  }
  int ppb;
  int nb = skinny_wgrad_blocks(npix, &ppb);
  RG_REQUIRE(ws && ws_bytes >= (size_t)nb * elems * sizeof(float), RG_EWORKSPACE, "skinny_wgrad: workspace too small");
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((skinny_wgrad_kernel<T>), dim3(nb), dim3(256), 0, st, (const T*)low, high_nchw, (float*)ws, N,
                       Ho, Wo, O, ppb);
    RG_LAUNCH_CHECK("skinny_wgrad");
  })
  return rg_reduce_slabs((const float*)ws, dw, elems, nb, accumulate, 0, 0, st);
}
