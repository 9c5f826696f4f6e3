// rg_skinny.hip -- the two image-side layers (3 <-> 64 channels at 256x256).  K = 48 (or 3 output
// channels) cannot fill an MFMA tile; these layers are HBM-bound (SURVEY 7 hard part 3), so they
// run on the vector ALUs with scalar-loaded weights and coalesced image-side accesses.
//   first_down : NCHW fp32 image  -> NHWC T, 4x4 s2 p1 conv (+bias, LeakyReLU)
//   last_up    : NHWC T           -> NCHW fp32 image, transposed conv (+bias, tanh)
//   wgrad      : dW[o][3][16] = sum_pix low[pix][o] * patch(pix)
#include "rg_internal.h"

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

size_t rg_skinny_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I) {
  (void)I;
  int ppb;
  int nb = skinny_wgrad_blocks((long long)N * Ho * Wo, &ppb);
  return (size_t)nb * O * SK_K * sizeof(float);
}

int rg_skinny_wgrad_impl(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                         int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  (void)I;
  int ppb;
  int nb = skinny_wgrad_blocks((long long)N * Ho * Wo, &ppb);
  size_t elems = (size_t)O * SK_K;
  RG_REQUIRE(ws && ws_bytes >= (size_t)nb * elems * sizeof(float), RG_EWORKSPACE, "skinny_wgrad: workspace too small");
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((skinny_wgrad_kernel<T>), dim3(nb), dim3(256), 0, st, (const T*)low, high_nchw, (float*)ws, N,
                       Ho, Wo, O, ppb);
    RG_LAUNCH_CHECK("skinny_wgrad");
  })
  return rg_reduce_slabs((const float*)ws, dw, elems, nb, accumulate, 0, 0, st);
}
