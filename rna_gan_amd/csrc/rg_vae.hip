// rg_vae.hip -- betaVAE TRAINING path: the pieces around the GEMMs.
//   reference: src/betaVAE.py:18-42 (RNAEncoder), :63-107 (betaVAE: encode / reparametrize / decoder),
//              :145-163 (betaVAEloss), :166-284 (train_betaVAE).
// A Linear layer's three GEMMs (y = x W^T, dx = dy W, dW = dy^T x) all run as the "NT" GEMM of the encoder path
// (rg_gemm_nt_bf16 = gather_gemm_dma_kernel MODE_PLAIN / fp32 epilogue) on operands this file re-lays:
//   * rg_transpose_pack_bf16 : fp32 [R][C] -> bf16 [C_pad][R_pad] (zero padded) -- W^T for dx, dy^T and x^T for dW
//   * rg_transpose_f32       : fp32 parity mode (operands of the functor GEMM)
// plus the elementwise / reduction kernels of the VAE: input dropout, reparametrisation (+ backward), the
// beta-VAE loss with its three gradients, tanh.  Everything here is HBM-bound; the whole training step is bound by
// the optimizer's 30 B/parameter and the fp32 weight-gradient write (303 M parameters).
#include "rg_common.h"
#include "rg_internal.h"

namespace {

inline unsigned vgrid(size_t n, int per_block = 256) {
  size_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// dst[c][r] = src[r][c]; 32x32 tiles through LDS (padded: no bank conflicts), both sides coalesced
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * R + r] = tile[tx][ty + 8 * k];
  }
}
// dst[c][r] (bf16, [Cp][Rp], zero padded) = src[r][c] (fp32 [R][C]); 64x64 tiles, 2 bf16 per store
__global__ __launch_bounds__(256) void transpose_pack_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                                  int R, int C, int Rp, int Cp) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = r0 + ty + 4 * k, c = c0 + tx;
    tile[ty + 4 * k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  const int px = threadIdx.x & 31, py = threadIdx.x >> 5;       // 32 pairs x 8
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + py + 8 * k, r = r0 + 2 * px;
    if (c < Cp && r < Rp) {                                     // Rp is even (a multiple of 64)
      const uint32_t v = (uint32_t)f32_to_h16(tile[2 * px][py + 8 * k]) |
                         ((uint32_t)f32_to_h16(tile[2 * px + 1][py + 8 * k]) << 16);
      *reinterpret_cast<uint32_t*>(dst + (size_t)c * Rp + r) = v;
    }
  }
}

// y[n][j] = mask[n][j] ? x[n][j] * scale : 0 for j < F, 0 for the pad columns F <= j < ld  (nn.Dropout, train mode)
__global__ void dropout_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float* __restrict__ y, int N,
                               int F, int ld, float scale) {
  const size_t tot = (size_t)N * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / ld;
    const int j = (int)(i - n * ld);
    float v = 0.f;
    if (j < F) {
      const size_t s = n * F + j;
      v = (!mask || mask[s]) ? x[s] * scale : 0.f;
    }
    y[i] = v;
  }
}
// z = mu + eps * exp(0.5 * logvar)    (src/betaVAE.py:96-100)
__global__ void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                               float* __restrict__ z, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    z[i] = mu[i] + eps[i] * expf(0.5f * lv[i]);
}
// gmu = gmu_loss + gz ; glv = glv_loss + gz * eps * 0.5 * exp(0.5 * logvar)
__global__ void reparam_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ lv, const float* __restrict__ eps,
                                   const float* __restrict__ gmu_loss, const float* __restrict__ glv_loss,
                                   float* __restrict__ gmu, float* __restrict__ glv, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = gz[i];
    gmu[i] = (gmu_loss ? gmu_loss[i] : 0.f) + g;
    glv[i] = (glv_loss ? glv_loss[i] : 0.f) + g * eps[i] * 0.5f * expf(0.5f * lv[i]);
  }
}
__global__ void add_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] += x[i];
}
__global__ void tanh_kernel(float* __restrict__ x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = tanhf(x[i]);
}

// loss, stage 1: block partials of sum (xr - x)^2 (pad columns hold zeros on both sides) and of
// sum (1 + lv - mu^2 - exp(lv)); the three gradients are written on the way.
__global__ __launch_bounds__(256) void vae_loss_partial_kernel(const float* __restrict__ x, const float* __restrict__ xr,
                                                               float* __restrict__ gxr, size_t nx, float gx_scale,
                                                               const float* __restrict__ mu, const float* __restrict__ lv,
                                                               float* __restrict__ gmu, float* __restrict__ glv, size_t nz,
                                                               float gz_scale, float* __restrict__ partial) {
  __shared__ float sm[4];
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (size_t i = t0; i < nx; i += stride) {
    const float d = xr[i] - x[i];
    s1 += d * d;
    gxr[i] = gx_scale * d;                          // d recons / d xr = 2 (xr - x) / (N F)
  }
  for (size_t i = t0; i < nz; i += stride) {
    const float m = mu[i], l = lv[i], e = expf(l);
    s2 += 1.f + l - m * m - e;
    gmu[i] = gz_scale * m;                          // beta * d kld / d mu = beta * mu / N
    glv[i] = gz_scale * 0.5f * (e - 1.f);           // beta * d kld / d lv = beta * 0.5 (exp(lv) - 1) / N
  }
  const float a = block_sum_256(s1, sm);
  const float b = block_sum_256(s2, sm);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b; }
}
// stage 2 (one block): out = {total, reconstruction, kl}   (src/betaVAE.py:146-161)
__global__ __launch_bounds__(256) void vae_loss_final_kernel(const float* __restrict__ partial, int nb, float inv_nf, float inv_n,
                                                             float beta, int training, float* __restrict__ out) {
  __shared__ float sm[4];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { s1 += partial[2 * i]; s2 += partial[2 * i + 1]; }
  const float a = block_sum_256(s1, sm);
  const float b = block_sum_256(s2, sm);
  if (threadIdx.x == 0) {
    const float recons = a * inv_nf, kld = -0.5f * b * inv_n;
    out[0] = training ? recons + beta * kld : recons;
    out[1] = recons;
    out[2] = kld;
  }
}

}  // namespace

extern "C" int rg_transpose_f32(const float* src, float* dst, int R, int C, void* stream) {
  RG_REQUIRE(src && dst && R > 0 && C > 0, RG_EINVAL, "transpose_f32: bad args");
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, rg_stream(stream), src, dst, R, C);
  RG_LAUNCH_CHECK("transpose_f32");
  return RG_OK;
}
extern "C" int rg_transpose_pack_bf16(const float* src, void* dst, int R, int C, int R_pad, int C_pad, void* stream) {
  RG_REQUIRE(src && dst && R > 0 && C > 0 && R_pad >= R && C_pad >= C && R_pad % 64 == 0, RG_EINVAL,
             "transpose_pack_bf16: bad args (R_pad must be a multiple of 64)");
  hipLaunchKernelGGL(transpose_pack_bf16_kernel, dim3((C_pad + 63) / 64, R_pad / 64), dim3(256), 0, rg_stream(stream), src,
                     (uint16_t*)dst, R, C, R_pad, C_pad);
  RG_LAUNCH_CHECK("transpose_pack_bf16");
  return RG_OK;
}
extern "C" size_t rg_gemm_nt_bf16_workspace_bytes(int M, int K_pad, int Nout) {
  if (M <= 0 || K_pad <= 0 || Nout <= 0) return 0;
  return rg_mfma_linear_ws_bytes(M, K_pad, (Nout + 7) / 8 * 8) + 256;
}
extern "C" int rg_gemm_nt_bf16(const void* a, const void* b, const float* scale, const float* shift, float* y, int ldy, int M,
                               int K_pad, int Nout, float slope, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(a && b && y && M > 0 && K_pad > 0 && K_pad % 64 == 0 && Nout > 0 && ldy >= Nout, RG_EINVAL,
             "gemm_nt_bf16: bad args (K_pad must be a multiple of 64)");
  return rg_mfma_linear(a, b, scale, shift, y, ldy, M, K_pad, Nout, slope, ws, ws_bytes, rg_stream(stream));
}

extern "C" int rg_vae_dropout(const float* x, const unsigned char* mask, float* y, int N, int F, int ld, float scale,
                              void* stream) {
  RG_REQUIRE(x && y && N > 0 && F > 0 && ld >= F, RG_EINVAL, "vae_dropout: bad args");
  hipLaunchKernelGGL(dropout_kernel, dim3(vgrid((size_t)N * ld)), dim3(256), 0, rg_stream(stream), x, mask, y, N, F, ld, scale);
  RG_LAUNCH_CHECK("vae_dropout");
  return RG_OK;
}
extern "C" int rg_vae_reparam(const float* mu, const float* logvar, const float* eps, float* z, size_t n, void* stream) {
  RG_REQUIRE(mu && logvar && eps && z && n > 0, RG_EINVAL, "vae_reparam: bad args");
  hipLaunchKernelGGL(reparam_kernel, dim3(vgrid(n)), dim3(256), 0, rg_stream(stream), mu, logvar, eps, z, n);
  RG_LAUNCH_CHECK("vae_reparam");
  return RG_OK;
}
extern "C" int rg_vae_reparam_bwd(const float* gz, const float* logvar, const float* eps, const float* gmu_loss,
                                  const float* glv_loss, float* gmu, float* glv, size_t n, void* stream) {
  RG_REQUIRE(gz && logvar && eps && gmu && glv && n > 0, RG_EINVAL, "vae_reparam_bwd: bad args");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(vgrid(n)), dim3(256), 0, rg_stream(stream), gz, logvar, eps, gmu_loss, glv_loss,
                     gmu, glv, n);
  RG_LAUNCH_CHECK("vae_reparam_bwd");
  return RG_OK;
}
extern "C" int rg_add_inplace(float* y, const float* x, size_t n, void* stream) {
  RG_REQUIRE(x && y && n > 0, RG_EINVAL, "add_inplace: bad args");
  hipLaunchKernelGGL(add_kernel, dim3(vgrid(n)), dim3(256), 0, rg_stream(stream), y, x, n);
  RG_LAUNCH_CHECK("add_inplace");
  return RG_OK;
}
extern "C" int rg_tanh_inplace(float* x, size_t n, void* stream) {
  RG_REQUIRE(x && n > 0, RG_EINVAL, "tanh_inplace: bad args");
  hipLaunchKernelGGL(tanh_kernel, dim3(vgrid(n)), dim3(256), 0, rg_stream(stream), x, n);
  RG_LAUNCH_CHECK("tanh_inplace");
  return RG_OK;
}
extern "C" size_t rg_vae_loss_workspace_bytes(void) { return 2 * 1024 * sizeof(float); }
extern "C" int rg_vae_loss(const float* x, const float* x_recons, int N, int F, int ld, const float* z_mean,
                           const float* z_logvar, int Z, float beta, int training, float* losses, float* g_recons,
                           float* g_mean, float* g_logvar, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && x_recons && z_mean && z_logvar && losses && g_recons && g_mean && g_logvar && N > 0 && F > 0 && ld >= F &&
                 Z > 0, RG_EINVAL, "vae_loss: bad args");
  RG_REQUIRE(ws && ws_bytes >= rg_vae_loss_workspace_bytes(), RG_EWORKSPACE, "vae_loss: workspace too small");
  const size_t nx = (size_t)N * ld, nz = (size_t)N * Z;
  unsigned nb = vgrid(nx > nz ? nx : nz, 1024);
  if (nb > 1024) nb = 1024;
  const float inv_nf = 1.0f / ((float)N * (float)F), inv_n = 1.0f / (float)N;
  // the KL gradients only flow when training (total = recons + beta * kld); in evaluation total = recons
  hipLaunchKernelGGL(vae_loss_partial_kernel, dim3(nb), dim3(256), 0, rg_stream(stream), x, x_recons, g_recons, nx,
                     2.0f * inv_nf, z_mean, z_logvar, g_mean, g_logvar, nz, training ? beta * inv_n : 0.f, (float*)ws);
  RG_LAUNCH_CHECK("vae_loss");
  hipLaunchKernelGGL(vae_loss_final_kernel, dim3(1), dim3(256), 0, rg_stream(stream), (const float*)ws, (int)nb, inv_nf, inv_n,
                     beta, training, losses);
  RG_LAUNCH_CHECK("vae_loss");
  return RG_OK;
}
