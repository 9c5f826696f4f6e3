// rg_skinny.hip -- the two image-side layers (3 <-> 64 channels at 256x256).  K = 48 (or 3 output
// channels) cannot fill an MFMA tile; these layers are HBM-bound (SURVEY 7 hard part 3), so they
// run on the vector ALUs with the weights broadcast from LDS and coalesced image-side accesses.
//   first_down : NCHW fp32 image  -> NHWC T, 4x4 s2 p1 conv (+bias, LeakyReLU)
//   last_up    : NHWC T           -> NCHW fp32 image, transposed conv (+bias, tanh)
//   wgrad      : dW[o][3][16] = sum_pix low[pix][o] * patch(pix)
#include "rg_internal.h"

namespace {

constexpr int SK_I = 3;
constexpr int SK_K = SK_I * 16;   // 48

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void first_down_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ y, int N,
                                                         int H, int W, int O, float slope) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [48][O]
  const int t = threadIdx.x;
  for (int i = t; i < SK_K * O; i += 256) {
    int k = i / O, o = i - k * O;
    wl[i] = Elem<T>::round(w[o * SK_K + k]);
  }
  __syncthreads();
  const int Ho = H >> 1, Wo = W >> 1;
  const long long npix = (long long)N * Ho * Wo;
  const long long p = (long long)blockIdx.x * 64 + (t & 63);
  const int og = t >> 6;
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
  for (int oc = og * 16; oc < O; oc += 64) {
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[oc + j] : 0.f;
#pragma unroll
    for (int k = 0; k < SK_K; ++k) {
      const float4* wr = reinterpret_cast<const float4*>(wl + k * O + oc);
      float xv = patch[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 wv = wr[q];
        acc[q * 4 + 0] = fmaf(xv, wv.x, acc[q * 4 + 0]);
        acc[q * 4 + 1] = fmaf(xv, wv.y, acc[q * 4 + 1]);
        acc[q * 4 + 2] = fmaf(xv, wv.z, acc[q * 4 + 2]);
        acc[q * 4 + 3] = fmaf(xv, wv.w, acc[q * 4 + 3]);
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = lrelu_f(acc[j], slope);
    T* yo = y + p * O + oc;
#pragma unroll
    for (int q = 0; q < 4; ++q) Vec<T, 4>::st(yo + q * 4, acc + q * 4);
  }
}

// ---------------------------------------------------------------------------------------------
// one thread per low-res position (hq,wq): its 2x2 output quad x 3 channels
template <typename T>
__global__ __launch_bounds__(256) void last_up_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N,
                                                      int Ho, int Wo, int O, int apply_tanh) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [O][48]  (torch order: i*16 + kh*4 + kw)
  const int t = threadIdx.x;
  for (int i = t; i < O * SK_K; i += 256) wl[i] = Elem<T>::round(w[i]);
  __syncthreads();
  const long long npix = (long long)N * Ho * Wo;
  const long long p = (long long)blockIdx.x * 256 + t;
  if (p >= npix) return;
  const int wq = (int)(p % Wo);
  const long long tq = p / Wo;
  const int hq = (int)(tq % Ho), n = (int)(tq / Ho);
  float acc[2][2][SK_I];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < SK_I; ++i) acc[a][b][i] = bias ? bias[i] : 0.f;
  const T* xb = x + (long long)n * Ho * Wo * O;
  bool vh[3], vw[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    vh[d] = (unsigned)(hq + d - 1) < (unsigned)Ho;
    vw[d] = (unsigned)(wq + d - 1) < (unsigned)Wo;
  }
  for (int o0 = 0; o0 < O; o0 += 4) {
    float xv[3][3][4];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        if (vh[dh] && vw[dw]) {
          Vec<T, 4>::ld(xb + ((long long)(hq + dh - 1) * Wo + (wq + dw - 1)) * O + o0, xv[dh][dw]);
        } else {
          xv[dh][dw][0] = xv[dh][dw][1] = xv[dh][dw][2] = xv[dh][dw][3] = 0.f;
        }
      }
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const float* wr = wl + (o0 + oo) * SK_K;
      // output row parity ph: ph=0 uses (kh=1, ho=hq), (kh=3, ho=hq-1); ph=1 uses (kh=0, ho=hq+1), (kh=2, ho=hq)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int kh = ph == 0 ? (a == 0 ? 1 : 3) : (a == 0 ? 0 : 2);
              const int dh = ph == 0 ? (a == 0 ? 1 : 0) : (a == 0 ? 2 : 1);   // index into xv: ho-hq+1
              const int kw = pw == 0 ? (b == 0 ? 1 : 3) : (b == 0 ? 0 : 2);
              const int dw = pw == 0 ? (b == 0 ? 1 : 0) : (b == 0 ? 2 : 1);
              const float xval = xv[dh][dw][oo];
#pragma unroll
              for (int i = 0; i < SK_I; ++i)
                acc[ph][pw][i] = fmaf(xval, wr[i * 16 + kh * 4 + kw], acc[ph][pw][i]);
            }
    }
  }
  const int H = 2 * Ho, W = 2 * Wo;
#pragma unroll
  for (int i = 0; i < SK_I; ++i)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      float v0 = acc[ph][0][i], v1 = acc[ph][1][i];
      if (apply_tanh) { v0 = tanhf(v0); v1 = tanhf(v1); }
      float2* dst = reinterpret_cast<float2*>(y + (((long long)n * SK_I + i) * H + 2 * hq + ph) * W + 2 * wq);
      *dst = make_float2(v0, v1);
    }
}

// ---------------------------------------------------------------------------------------------
// block: a contiguous chunk of pixels; thread (o = t&63 [+64*oc], kg = t>>6) owns 12 k's.
template <typename T>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const T* __restrict__ low, const float* __restrict__ high,
                                                           float* __restrict__ slab, int N, int Ho, int Wo, int O,
                                                           int pix_per_block) {
  __shared__ __attribute__((aligned(16))) float lo_s[64][128 + 1];
  __shared__ __attribute__((aligned(16))) float pa_s[64][SK_K];
  const int t = threadIdx.x;
  const int o = t & 63, kg = t >> 6;
  const int H = 2 * Ho, W = 2 * Wo;
  const long long npix = (long long)N * Ho * Wo;
  const long long pb = (long long)blockIdx.x * pix_per_block;
  long long pe = pb + pix_per_block;
  if (pe > npix) pe = npix;
  const int noc = O >> 6;   // 1 or 2
  float acc[2][12];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[c][j] = 0.f;

  for (long long p0 = pb; p0 < pe; p0 += 64) {
    // stage low tile [64 pix][O] and the 48-value patches
    for (int i = t; i < 64 * O; i += 256) {
      int pr = i / O, oc = i - pr * O;
      long long p = p0 + pr;
      lo_s[pr][oc] = p < pe ? Elem<T>::ld(low + p * O + oc) : 0.f;
    }
    for (int i = t; i < 64 * SK_K; i += 256) {
      int pr = i / SK_K, k = i - pr * SK_K;
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
    for (int pr = 0; pr < 64; ++pr) {
      const float4* pk = reinterpret_cast<const float4*>(&pa_s[pr][kg * 12]);
      float4 k0 = pk[0], k1 = pk[1], k2 = pk[2];
      float kv[12] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x, k2.y, k2.z, k2.w};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c < noc) {
          float a = lo_s[pr][o + 64 * c];
#pragma unroll
          for (int j = 0; j < 12; ++j) acc[c][j] = fmaf(a, kv[j], acc[c][j]);
        }
      }
    }
    __syncthreads();
  }
  float* sl = slab + (long long)blockIdx.x * O * SK_K;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c < noc) {
#pragma unroll
      for (int j = 0; j < 12; ++j) sl[(o + 64 * c) * SK_K + kg * 12 + j] = acc[c][j];
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
  unsigned blocks = (unsigned)((npix + 63) / 64);
  size_t sh = (size_t)SK_K * O * sizeof(float);
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((first_down_kernel<T>), dim3(blocks), dim3(256), sh, st, x, w, bias, (T*)y, N, H, W, O, slope);
    RG_LAUNCH_CHECK("first_down");
    return RG_OK;
  })
}

int rg_skinny_last_up(const void* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo, int O, int I,
                      int apply_tanh, int dtype, hipStream_t st) {
  (void)I;
  long long npix = (long long)N * Ho * Wo;
  unsigned blocks = (unsigned)((npix + 255) / 256);
  size_t sh = (size_t)SK_K * O * sizeof(float);
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((last_up_kernel<T>), dim3(blocks), dim3(256), sh, st, (const T*)x, w, bias, y, N, Ho, Wo, O,
                       apply_tanh);
    RG_LAUNCH_CHECK("last_up");
    return RG_OK;
  })
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
