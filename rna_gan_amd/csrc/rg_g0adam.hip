// rg_g0adam.hip -- the generator's first layer: weight gradient + Adam step in ONE streaming pass.
//
// G.0 = ConvTranspose2d(E, C, 4, 1, 0) on a 1 x 1 input is a plain GEMM; its weight w[E][C][4][4] holds 67 M of the
// generator's 112 M parameters (E = C = 2048).  Its gradient dw[e][c][tap] = sum_n z[n][e] * gz0[n][tap][c] contracts over
// the BATCH only (K = 64): producing it is pure output streaming (268 MB of fp32 written by rg_g0_wgrad: 93 us), and the
// optimizer step that follows reads it straight back (rg_adam_step_dev: 30 B per parameter).  Here the gradient of a tile is
// formed in registers (64 FMAs per element from LDS-resident bf16 operand tiles -- the same bf16 operands / fp32 accumulation
// as the MFMA path; ~80 us of VALU work chip-wide, hidden under the stream) and the Adam update of torch.optim.Adam is
// applied in place: 12 B read (p, m, v) + 14 B written (p, m, v, bf16 shadow) per parameter, the 8 B gradient round trip
// and one launch are gone.  Used by rna_gan_amd.optim.Adam when the generator-loss train_op runs in a single process (a
// data-parallel run needs the gradient in memory for the all-reduce).  Bound: HBM, 26 B per parameter.
//
// Tile: 64 e-rows x 256 columns j = (c, tap) (16 channels x 16 taps of the master's [E][C*16] row), 256 threads:
// thread (tj = t % 64, te = t / 64 = its wave) owns columns 4 tj .. 4 tj + 3 of rows 16 te .. 16 te + 15 (64 accumulators).
#include "rg_internal.h"

namespace {

constexpr int GA_E = 64, GA_J = 256, GA_N = 64;

struct AdamC { float b1, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2, wd; };
__device__ __forceinline__ void adam_upd(const AdamC& a, float& pp, float gg, float& mm, float& vv) {
  // rg_misc.hip Adam::upd (torch.optim.Adam single-tensor arithmetic)
  if (a.wd != 0.f) gg += a.wd * pp;
  mm = mm + a.omb1 * (gg - mm);
  vv = a.b2 * vv + a.omb2 * gg * gg;
  const float denom = sqrtf(vv) * a.inv_sqrt_bc2 + a.eps;
  pp -= a.step_size * (mm / denom);
}

template <typename TG>      // TG: element type of gz0 (bf16_t or float)
__global__ __launch_bounds__(256) void g0_wgrad_adam_kernel(const float* __restrict__ z, const TG* __restrict__ gz0,
                                                            float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ hyper, uint16_t* __restrict__ shadow,
                                                            int N, int E, int C) {
  __shared__ __attribute__((aligned(16))) uint16_t zs[GA_N][GA_E];      // 8 KB   z tile, bf16, [n][e]
  __shared__ __attribute__((aligned(16))) uint16_t gs[GA_N][GA_J];      // 32 KB  gz0 tile, bf16, [n][c_local * 16 + tap]
  const int t = threadIdx.x, tj = t & 63, te = t >> 6;
  const int e0 = blockIdx.y * GA_E, c0 = blockIdx.x * 16;               // 16 channels = 256 columns
  float acc[16][4];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;

  for (int n0 = 0; n0 < N; n0 += GA_N) {
    if (n0) __syncthreads();
    // ---- stage z[n0 .. n0+63][e0 .. e0+63] (fp32 -> bf16): 4096 elements, 16 per thread (4 x float4)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = t + 256 * k;                 // float4 index: n = idx / 16, e4 = idx % 16
      const int n = idx >> 4, e4 = (idx & 15) * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + n < N) x = *reinterpret_cast<const float4*>(z + (size_t)(n0 + n) * E + e0 + e4);
      *reinterpret_cast<uint2*>(&zs[n][e4]) = make_uint2((uint32_t)f32_to_bf16(x.x) | ((uint32_t)f32_to_bf16(x.y) << 16),
                                                         (uint32_t)f32_to_bf16(x.z) | ((uint32_t)f32_to_bf16(x.w) << 16));
    }
    // ---- stage gz0[n][tap][c0 .. c0+15] -> gs[n][cl * 16 + tap]: 64 n x 16 taps x 16 channels; thread = (n, 4 taps)
    {
      const int n = t >> 2, tq = (t & 3) * 4;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int tap = tq + a;
        float x[16];
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) x[cl] = 0.f;
        if (n0 + n < N) {
          const TG* src = gz0 + ((size_t)(n0 + n) * 16 + tap) * C + c0;
          Vec<TG, 8>::ld(src, x);
          Vec<TG, 8>::ld(src + 8, x + 8);
        }
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) gs[n][cl * 16 + tap] = f32_to_bf16(x[cl]);
      }
    }
    __syncthreads();
    // ---- 64 n x (16 e x 4 j) FMAs per thread
#pragma unroll 4
    for (int n = 0; n < GA_N; ++n) {
      const uint2 gq = *reinterpret_cast<const uint2*>(&gs[n][4 * tj]);
      const float g0 = __uint_as_float(gq.x << 16), g1 = __uint_as_float(gq.x & 0xffff0000u);
      const float g2 = __uint_as_float(gq.y << 16), g3 = __uint_as_float(gq.y & 0xffff0000u);
      const uint4 za = *reinterpret_cast<const uint4*>(&zs[n][16 * te]), zb = *reinterpret_cast<const uint4*>(&zs[n][16 * te + 8]);
      const uint32_t zw[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float zv = __uint_as_float((i & 1) ? (zw[i >> 1] & 0xffff0000u) : (zw[i >> 1] << 16));
        acc[i][0] += zv * g0; acc[i][1] += zv * g1; acc[i][2] += zv * g2; acc[i][3] += zv * g3;
      }
    }
  }

  // ---- Adam on the tile: rows e0 + 16 te + i, columns c0 * 16 + 4 tj .. + 3 (a wave streams 1 KB of each row)
  const AdamC a{hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7]};
  const size_t ld = (size_t)C * 16;
  const size_t col = (size_t)c0 * 16 + 4 * tj;
#pragma unroll
  for (int i0 = 0; i0 < 16; i0 += 4) {
    float4 P[4], M[4], V[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t idx = (size_t)(e0 + 16 * te + i0 + k) * ld + col;
      P[k] = *reinterpret_cast<const float4*>(p + idx);
      M[k] = *reinterpret_cast<const float4*>(m + idx);
      V[k] = *reinterpret_cast<const float4*>(v + idx);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t idx = (size_t)(e0 + 16 * te + i0 + k) * ld + col;
      adam_upd(a, P[k].x, acc[i0 + k][0], M[k].x, V[k].x);
      adam_upd(a, P[k].y, acc[i0 + k][1], M[k].y, V[k].y);
      adam_upd(a, P[k].z, acc[i0 + k][2], M[k].z, V[k].z);
      adam_upd(a, P[k].w, acc[i0 + k][3], M[k].w, V[k].w);
      *reinterpret_cast<float4*>(p + idx) = P[k];
      *reinterpret_cast<float4*>(m + idx) = M[k];
      *reinterpret_cast<float4*>(v + idx) = V[k];
      if (shadow)
        *reinterpret_cast<uint2*>(shadow + idx) = make_uint2((uint32_t)f32_to_bf16(P[k].x) | ((uint32_t)f32_to_bf16(P[k].y) << 16),
                                                             (uint32_t)f32_to_bf16(P[k].z) | ((uint32_t)f32_to_bf16(P[k].w) << 16));
    }
  }
}

}  // namespace

extern "C" int rg_g0_wgrad_adam_supported(int N, int E, int C, int dtype) {
  return N > 0 && E % GA_E == 0 && C % 16 == 0 && (dtype == RG_BF16 || dtype == RG_F32) ? 1 : 0;
}

extern "C" int rg_g0_wgrad_adam(const float* z, const void* gz0, float* p, float* m, float* v, const float* hyper,
                                void* shadow_bf16, int N, int E, int C, int dtype, void* stream) {
  RG_REQUIRE(z && gz0 && p && m && v && hyper, RG_EINVAL, "g0_wgrad_adam: null");
  RG_REQUIRE(rg_g0_wgrad_adam_supported(N, E, C, dtype), RG_EUNSUPPORTED, "g0_wgrad_adam: E %% 64 == 0 and C %% 16 == 0 required");
  RG_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)z | (uintptr_t)gz0) & 15) == 0 &&
                 (((uintptr_t)shadow_bf16) & 7) == 0, RG_EINVAL, "g0_wgrad_adam: 16-byte aligned buffers required");
  const dim3 grid((unsigned)(C / 16), (unsigned)(E / GA_E));
  hipStream_t st = rg_stream(stream);
  if (dtype == RG_BF16)
    hipLaunchKernelGGL(g0_wgrad_adam_kernel<bf16_t>, grid, dim3(256), 0, st, z, (const bf16_t*)gz0, p, m, v, hyper,
                       (uint16_t*)shadow_bf16, N, E, C);
  else
    hipLaunchKernelGGL(g0_wgrad_adam_kernel<float>, grid, dim3(256), 0, st, z, (const float*)gz0, p, m, v, hyper,
                       (uint16_t*)shadow_bf16, N, E, C);
  RG_LAUNCH_CHECK("g0_wgrad_adam");
  return RG_OK;
}
