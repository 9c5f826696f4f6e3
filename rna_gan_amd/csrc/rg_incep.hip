// rg_incep.hip -- the data-movement kernels the Inception-v3 feature extractor of the FID metric needs around the GEMM
// (src/fid.py:33-94: torchvision inception_v3 up to Mixed_7c, spatially averaged).  NHWC fp32 activations; every
// BasicConv2d (Conv2d without bias + eval-mode BatchNorm2d(eps 1e-3) + ReLU) is ONE GEMM with the folded BatchNorm affine
// and the ReLU in its epilogue (rg_linear_affine_act, slope 0) over the patch matrix rg_im2col_nhwc writes (1x1
// convolutions read the activation in place); the branches of an Inception block write straight into their channel slice
// of the block's output (row stride = total channels), so torch.cat costs nothing.  HBM-bound passes, 16 bytes per thread
// where the channel count allows.
#include "rg_internal.h"

namespace {

__global__ void im2col_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int ldx,
                                   int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int vec) {
  // one thread per (output pixel, tap, channel group of `vec`)
  const int cg = C / vec;
  const size_t total = (size_t)N * Ho * Wo * kh * kw * cg;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cg) * vec;
    size_t r = i / cg;
    const int j = (int)(r % kw); r /= kw;
    const int ii = (int)(r % kh); r /= kh;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int hi = ho * sh - ph + ii, wi = wo * sw - pw + j;
    const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
    float* d = y + (((size_t)(n * Ho + ho) * Wo + wo) * kh * kw + (size_t)ii * kw + j) * C + c;
    const float* s = x + ((size_t)(n * H + hi) * W + wi) * ldx + c;
    if (vec == 4) {
      *reinterpret_cast<float4*>(d) = ok ? *reinterpret_cast<const float4*>(s) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      d[0] = ok ? s[0] : 0.f;
    }
  }
}

// mode 0: max over the window's IN-RANGE taps (F.max_pool2d, no padding used by Inception); mode 1: mean with the padding
// counted (F.avg_pool2d(kernel, stride, padding), count_include_pad = True: divisor k*k everywhere)
__global__ void pool2d_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int ldx,
                                   int ldy, int k, int stride_, int pad, int mode, int Ho, int Wo) {
  const size_t total = (size_t)N * Ho * Wo * C;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float acc = mode == 0 ? -3.402823466e38f : 0.f;
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const int hi = ho * stride_ - pad + a, wi = wo * stride_ - pad + b;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
          const float v = x[((size_t)(n * H + hi) * W + wi) * ldx + c];
          acc = mode == 0 ? fmaxf(acc, v) : acc + v;
        }
      }
    if (mode == 1) acc /= (float)(k * k);
    y[((size_t)(n * Ho + ho) * Wo + wo) * ldy + c] = acc;
  }
}

__global__ void nchw_to_nhwc_affine_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW,
                                           const float* __restrict__ scale, const float* __restrict__ shift) {
  const size_t total = (size_t)N * C * HW;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {     // i indexes the OUTPUT (n, p, c)
    const int c = (int)(i % C);
    const size_t r = i / C;
    const int p = (int)(r % HW);
    const int n = (int)(r / HW);
    y[i] = x[((size_t)n * C + c) * HW + p] * scale[c] + shift[c];
  }
}

// y[n][c] = mean over the HW positions of x[n][p][c] (adaptive_avg_pool2d to 1 x 1), rows summed in order (deterministic)
__global__ void spatial_mean_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int c = i % C, n = i / C;
  float acc = 0.f;
  for (int p = 0; p < HW; ++p) acc += x[((size_t)n * HW + p) * C + c];
  y[i] = acc / (float)HW;
}

static unsigned grid_of(size_t total) {
  size_t b = (total + 255) / 256;
  return (unsigned)(b > 65535 * 4 ? 65535 * 4 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int rg_im2col_nhwc(const float* x, int ldx, float* cols, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                              int ph, int pw, void* stream) {
  RG_REQUIRE(x && cols && N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 &&
                 pw >= 0, RG_EINVAL, "im2col_nhwc: bad args");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  RG_REQUIRE(Ho > 0 && Wo > 0, RG_EINVAL, "im2col_nhwc: empty output");
  const int vec = (C % 4 == 0 && ldx % 4 == 0) ? 4 : 1;
  const size_t total = (size_t)N * Ho * Wo * kh * kw * (C / vec);
  hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(grid_of(total)), dim3(256), 0, rg_stream(stream), x, cols, N, H, W, C, ldx, kh, kw,
                     sh, sw, ph, pw, Ho, Wo, vec);
  RG_LAUNCH_CHECK("im2col_nhwc");
  return RG_OK;
}

extern "C" int rg_pool2d_nhwc(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int k, int stride_, int pad,
                              int mode, void* stream) {
  RG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C && k > 0 && stride_ > 0 && pad >= 0 &&
                 (mode == 0 || mode == 1), RG_EINVAL, "pool2d_nhwc: bad args");
  const int Ho = (H + 2 * pad - k) / stride_ + 1, Wo = (W + 2 * pad - k) / stride_ + 1;
  RG_REQUIRE(Ho > 0 && Wo > 0, RG_EINVAL, "pool2d_nhwc: empty output");
  hipLaunchKernelGGL(pool2d_nhwc_kernel, dim3(grid_of((size_t)N * Ho * Wo * C)), dim3(256), 0, rg_stream(stream), x, y, N, H, W, C,
                     ldx, ldy, k, stride_, pad, mode, Ho, Wo);
  RG_LAUNCH_CHECK("pool2d_nhwc");
  return RG_OK;
}

extern "C" int rg_nchw_to_nhwc_affine(const float* x_nchw, float* y_nhwc, int N, int C, int H, int W, const float* scale,
                                      const float* shift, void* stream) {
  RG_REQUIRE(x_nchw && y_nhwc && scale && shift && N > 0 && C > 0 && H > 0 && W > 0, RG_EINVAL, "nchw_to_nhwc_affine: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_affine_kernel, dim3(grid_of((size_t)N * C * H * W)), dim3(256), 0, rg_stream(stream), x_nchw,
                     y_nhwc, N, C, H * W, scale, shift);
  RG_LAUNCH_CHECK("nchw_to_nhwc_affine");
  return RG_OK;
}

extern "C" int rg_spatial_mean_nhwc(const float* x, float* y, int N, int HW, int C, void* stream) {
  RG_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, RG_EINVAL, "spatial_mean_nhwc: bad args");
  hipLaunchKernelGGL(spatial_mean_nhwc_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, rg_stream(stream), x, y, N, HW,
                     C);
  RG_LAUNCH_CHECK("spatial_mean_nhwc");
  return RG_OK;
}
