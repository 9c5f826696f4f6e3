// rg_g0adam.hip -- the generator's first layer: weight gradient + Adam step in ONE streaming pass.
//
// G.0 = ConvTranspose2d(E, C, 4, 1, 0) on a 1 x 1 input is a plain GEMM; its weight w[E][C][4][4] holds 67 M of the
// generator's 112 M parameters (E = C = 2048).  Its gradient dw[e][c][tap] = sum_n z[n][e] * gz0[n][tap][c] contracts over
// the BATCH only (K = 64): producing it is pure output streaming (268 MB of fp32 written by rg_g0_wgrad: 93 us), and the
// optimizer step that follows reads it straight back (rg_adam_step_dev: 30 B per parameter).  Here the gradient of a tile is
// formed on the matrix cores from LDS-resident bf16 operand tiles (v_mfma_f32_32x32x16_bf16, fp32 accumulation: K is the
// batch, or the batches of all ranks when the caller hands over gathered factors) and the Adam update of torch.optim.Adam is
// applied in place: 12 B read (p, m, v) + 14 B written (p, m, v, bf16 shadow) per parameter, the 8 B gradient round trip
// and one launch are gone.  Used by rna_gan_amd.optim.Adam when the generator-loss train_op runs in a single process (a
// data-parallel run needs the gradient in memory for the all-reduce).  Bound: HBM, 26 B per parameter.
//
// Tile: 64 e-rows x 256 columns j = (c, tap) (16 channels x 16 taps of the master's [E][C*16] row), 256 threads = 4 waves.
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

typedef __attribute__((ext_vector_type(8))) __bf16 ga_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ga_f32x16;
constexpr int GA_CP = GA_J + 4;  // fp32 pitch of the accumulator tile: rows 4 apart (the two lane halves) land 16 banks apart
constexpr int GA_P = 72;       // LDS row pitch (bf16) of the k-contiguous operand images: 144 B, 16-byte fragment reads spread over the banks

// Operand images are stored TRANSPOSED (contraction index n contiguous): zsT[e][n], gsT[j][n], so that a lane's MFMA
// fragment -- 8 consecutive n of one row e (A) or one column j (B), v_mfma_f32_32x32x16_bf16 -- is one 16-byte LDS read.
// Wave w owns columns 64 w .. 64 w + 63 of the 64 x 256 tile: 2 x 2 MFMA tiles, 4 k-steps per 64 samples (16 MFMAs where the
// VALU form issued 4096 FMAs per thread).  The accumulators go through an LDS tile, one 32-row half at a time, into the
// thread -> (8 rows x 4 consecutive columns) map of the streaming Adam pass (16-byte accesses, 1 KB per wave and row).
template <typename TG>      // TG: element type of gz0 (bf16_t or float)
__global__ __launch_bounds__(256) void g0_wgrad_adam_kernel(const float* __restrict__ z, const TG* __restrict__ gz0,
                                                            float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ hyper, uint16_t* __restrict__ shadow,
                                                            int N, int E, int C) {
  constexpr int OPS = (GA_E + GA_J) * GA_P * 2;                       // 46080 B of operand images
  static_assert(32 * GA_CP * 4 <= OPS, "accumulator tile must fit into the operand area");
  __shared__ __attribute__((aligned(16))) unsigned char smem[OPS];
  uint16_t* zsT = reinterpret_cast<uint16_t*>(smem);                  // [64 e][GA_P]
  uint16_t* gsT = zsT + GA_E * GA_P;                                  // [256 j][GA_P]
  float* ct = reinterpret_cast<float*>(smem);                         // [32 e][GA_CP] fp32 = 33 KB, after the k loop
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int e0 = blockIdx.y * GA_E, c0 = blockIdx.x * 16;             // 16 channels = 256 columns
  ga_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int n0 = 0; n0 < N; n0 += GA_N) {
    if (n0) __syncthreads();
    // ---- stage z[n0 .. n0+63][e0 .. e0+63] (fp32 -> bf16) transposed: zsT[e][n]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = t + 256 * k;                 // float4 index: n = idx / 16, e4 = idx % 16
      const int n = idx >> 4, e4 = (idx & 15) * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + n < N) x = *reinterpret_cast<const float4*>(z + (size_t)(n0 + n) * E + e0 + e4);
      zsT[(e4 + 0) * GA_P + n] = f32_to_bf16(x.x);
      zsT[(e4 + 1) * GA_P + n] = f32_to_bf16(x.y);
      zsT[(e4 + 2) * GA_P + n] = f32_to_bf16(x.z);
      zsT[(e4 + 3) * GA_P + n] = f32_to_bf16(x.w);
    }
    // ---- stage gz0[n][tap][c0 .. c0+15] -> gsT[cl * 16 + tap][n]; thread = (n, 4 taps)
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
        for (int cl = 0; cl < 16; ++cl) gsT[(cl * 16 + tap) * GA_P + n] = f32_to_bf16(x[cl]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ga_bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = __builtin_bit_cast(ga_bf16x8, *reinterpret_cast<const uint4*>(zsT + (32 * i + fr) * GA_P + 16 * ks + 8 * fh));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j] = __builtin_bit_cast(ga_bf16x8,
                                   *reinterpret_cast<const uint4*>(gsT + (64 * wave + 32 * j + fr) * GA_P + 16 * ks + 8 * fh));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- Adam on the tile, one 32-row half at a time: acc[i][j][4 g + q] = dW[e = 32 i + 8 g + 4 fh + q][j = 64 wave + 32 j + fr]
  const AdamC a{hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7]};
  const size_t ld = (size_t)C * 16;
  const int tj = t & 63, te = t >> 6;
  const size_t col = (size_t)c0 * 16 + 4 * tj;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    __syncthreads();                               // operand images (i = 0) / the previous half's tile (i = 1) are done with
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[(8 * (r >> 2) + 4 * fh + (r & 3)) * GA_CP + 64 * wave + 32 * j + fr] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 4) {
      float4 P[4], M[4], V[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t idx = (size_t)(e0 + 32 * i + 8 * te + k0 + k) * ld + col;
        P[k] = *reinterpret_cast<const float4*>(p + idx);
        M[k] = *reinterpret_cast<const float4*>(m + idx);
        V[k] = *reinterpret_cast<const float4*>(v + idx);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t idx = (size_t)(e0 + 32 * i + 8 * te + k0 + k) * ld + col;
        const float4 g4 = *reinterpret_cast<const float4*>(ct + (8 * te + k0 + k) * GA_CP + 4 * tj);
        adam_upd(a, P[k].x, g4.x, M[k].x, V[k].x);
        adam_upd(a, P[k].y, g4.y, M[k].y, V[k].y);
        adam_upd(a, P[k].z, g4.z, M[k].z, V[k].z);
        adam_upd(a, P[k].w, g4.w, M[k].w, V[k].w);
        *reinterpret_cast<float4*>(p + idx) = P[k];
        *reinterpret_cast<float4*>(m + idx) = M[k];
        *reinterpret_cast<float4*>(v + idx) = V[k];
        if (shadow)
          *reinterpret_cast<uint2*>(shadow + idx) = make_uint2((uint32_t)f32_to_bf16(P[k].x) | ((uint32_t)f32_to_bf16(P[k].y) << 16),
                                                               (uint32_t)f32_to_bf16(P[k].z) | ((uint32_t)f32_to_bf16(P[k].w) << 16));
      }
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
