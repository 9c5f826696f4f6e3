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
#include <stdlib.h>

namespace {

constexpr int GA_E = 64, GA_J = 256, GA_N = 64;

struct AdamC { float b1, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2, wd, ginv; };
__device__ __forceinline__ void adam_upd(const AdamC& a, float& pp, float gg, float& mm, float& vv) {
  rg_adam_upd(pp, gg * a.ginv, mm, vv, a.b2, a.omb1, a.omb2, a.eps, a.step_size, a.inv_sqrt_bc2, a.wd);     // rg_common.h
}

typedef rg_h16x8 ga_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ga_f32x16;
constexpr int GA_CP = GA_J + 4;  // fp32 pitch of the accumulator tile: rows 4 apart (the two lane halves) land 16 banks apart
constexpr int GA_P = 72;       // LDS row pitch (bf16) of the k-contiguous operand images: 144 B, 16-byte fragment reads spread over the banks

// Operand images are stored TRANSPOSED (contraction index n contiguous): zsT[e][n], gsT[j][n], so that a lane's MFMA
// fragment -- 8 consecutive n of one row e (A) or one column j (B), v_mfma_f32_32x32x16_bf16 -- is one 16-byte LDS read.
// Wave w owns columns 64 w .. 64 w + 63 of the 64 x 256 tile: 2 x 2 MFMA tiles, 4 k-steps per 64 samples (16 MFMAs where the
// VALU form issued 4096 FMAs per thread).  The accumulators go through an LDS tile, one 32-row half at a time, into the
// thread -> (8 rows x 4 consecutive columns) map of the streaming Adam pass (16-byte accesses, 1 KB per wave and row).
template <typename TG>      // TG: element type of gz0 (h16_t or float)
__global__ __launch_bounds__(256) void g0_wgrad_adam_kernel(const float* __restrict__ z, const TG* __restrict__ gz0,
                                                            float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ hyper, uint16_t* __restrict__ shadow,
                                                            int N, int E, int C, int e_fastest) {
  constexpr int OPS = (GA_E + GA_J) * GA_P * 2;                       // 46080 B of operand images
  static_assert(32 * GA_CP * 4 <= OPS, "accumulator tile must fit into the operand area");
  __shared__ __attribute__((aligned(16))) unsigned char smem[OPS];
  uint16_t* zsT = reinterpret_cast<uint16_t*>(smem);                  // [64 e][GA_P]
  uint16_t* gsT = zsT + GA_E * GA_P;                                  // [256 j][GA_P]
  float* ct = reinterpret_cast<float*>(smem);                         // [32 e][GA_CP] fp32 = 33 KB, after the k loop
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // tile order: e_fastest (default) -- the row tiles of one 256-column strip run back to back and share that strip's gz0 slab
  // (K x 512 B, the large operand) in L2 instead of re-reading it from memory; the other order (column tiles fastest) streams
  // adjacent 1 KB pieces of the same 64 rows and is slower at every K
  const int e0 = (e_fastest ? blockIdx.x : blockIdx.y) * GA_E, c0 = (e_fastest ? blockIdx.y : blockIdx.x) * 16;
  ga_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- operand staging, software-pipelined over the 64-sample chunks: the global loads of chunk k + 1 are issued before
  // the MFMAs of chunk k (K = world x batch in a data-parallel run: up to 8 chunks); two consecutive samples go into one
  // 4-byte LDS word (half the LDS write instructions of a per-sample scatter).
  //   z:   thread = (e = t & 63, sample pairs q = (t >> 6) + 4 i, i < 8): 2 x 8 scalar loads, 8 word writes
  //   gz0: thread = (sample pair np = t & 31, taps 2 (t >> 5), + 1): per tap 2 samples x 32 B, 16 word writes
  const int ze = t & 63, zq = t >> 6;
  const int gnp = t & 31, gtap = (t >> 5) * 2;
  float zr[8][2];
  uint4 gr[2][2][2];                               // [tap][sample of the pair][8-channel half]
  auto load_chunk = [&](int n0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + 2 * (zq + 4 * i);
      zr[i][0] = n < N ? z[(size_t)n * E + e0 + ze] : 0.f;
      zr[i][1] = n + 1 < N ? z[(size_t)(n + 1) * E + e0 + ze] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int n = n0 + 2 * gnp + k;
        gr[a][k][0] = gr[a][k][1] = make_uint4(0, 0, 0, 0);
        if (n < N) {
          const TG* src = gz0 + ((size_t)n * 16 + gtap + a) * C + c0;
          if constexpr (sizeof(TG) == 2) {
            gr[a][k][0] = *reinterpret_cast<const uint4*>(src);
            gr[a][k][1] = *reinterpret_cast<const uint4*>(src + 8);
          } else {                                 // fp32 source (fp32 activations): convert on the way in
            float x[16];
            Vec<TG, 8>::ld(src, x);
            Vec<TG, 8>::ld(src + 8, x + 8);
            uint32_t w[8];
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) w[c2] = (uint32_t)f32_to_h16(x[2 * c2]) | ((uint32_t)f32_to_h16(x[2 * c2 + 1]) << 16);
            gr[a][k][0] = make_uint4(w[0], w[1], w[2], w[3]);
            gr[a][k][1] = make_uint4(w[4], w[5], w[6], w[7]);
          }
        }
      }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<uint32_t*>(zsT + ze * GA_P + 2 * (zq + 4 * i)) =
          (uint32_t)f32_to_h16(zr[i][0]) | ((uint32_t)f32_to_h16(zr[i][1]) << 16);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t lo[4] = {gr[a][0][h].x, gr[a][0][h].y, gr[a][0][h].z, gr[a][0][h].w};   // sample 2 np: channels 8 h .. + 7
        const uint32_t hi[4] = {gr[a][1][h].x, gr[a][1][h].y, gr[a][1][h].z, gr[a][1][h].w};   // sample 2 np + 1
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          const int cl = 8 * h + 2 * c2;
          *reinterpret_cast<uint32_t*>(gsT + (cl * 16 + gtap + a) * GA_P + 2 * gnp) = (lo[c2] & 0xffffu) | (hi[c2] << 16);
          *reinterpret_cast<uint32_t*>(gsT + ((cl + 1) * 16 + gtap + a) * GA_P + 2 * gnp) = (lo[c2] >> 16) | (hi[c2] & 0xffff0000u);
        }
      }
  };
  load_chunk(0);
  for (int n0 = 0; n0 < N; n0 += GA_N) {
    if (n0) __syncthreads();                       // the previous chunk's fragment reads are done
    store_chunk();
    __syncthreads();
    if (n0 + GA_N < N) load_chunk(n0 + GA_N);      // in flight under this chunk's MFMAs
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
        for (int j = 0; j < 2; ++j) acc[i][j] = rg_mfma_h16_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- Adam on the tile, one 32-row half at a time: acc[i][j][4 g + q] = dW[e = 32 i + 8 g + 4 fh + q][j = 64 wave + 32 j + fr]
  const AdamC a{hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7], hyper[8]};
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
          *reinterpret_cast<uint2*>(shadow + idx) = make_uint2((uint32_t)f32_to_h16(P[k].x) | ((uint32_t)f32_to_h16(P[k].y) << 16),
                                                               (uint32_t)f32_to_h16(P[k].z) | ((uint32_t)f32_to_h16(P[k].w) << 16));
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same fusion for an nn.Linear weight W[O][I] (row pitch I, any O, I): dW[o][i] = sum_n g[n][o] * x[n][i] with the batch as
// the only contraction -- the betaVAE's layers (19198 x 6000 ... 2048 x 2048, batch 64: src/betaVAE_training.py) -- formed on
// MFMA and consumed by torch.optim.Adam's update in the same pass (24 B per parameter instead of 4 + 28 and a GEMM launch).
// Operands arrive TRANSPOSED and bf16: gT[O' >= O][ldn], xT[I' >= I][ldn] with the samples contiguous and zero padded to a
// multiple of 64 (rg_transpose_pack_bf16, which the unfused weight-gradient GEMM needs as well), i.e. exactly the k-contiguous
// LDS images of g0_wgrad_adam_kernel: staging is a straight 16-byte copy.  VEC: floats per access of the Adam stream (4: I % 4
// == 0 and 16-byte aligned segment; 2: even I, 8-byte aligned; 1 otherwise); ragged tile edges are guarded.
template <int VEC>
__global__ __launch_bounds__(256) void lin_wgrad_adam_kernel(const uint16_t* __restrict__ gT, const uint16_t* __restrict__ xT,
                                                             int ldn, int N, float* __restrict__ p, float* __restrict__ m,
                                                             float* __restrict__ v, const float* __restrict__ hyper, int O,
                                                             int I, uint16_t* __restrict__ wpack, int Kp) {
  constexpr int OPS = (GA_E + GA_J) * GA_P * 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OPS];
  uint16_t* zsT = reinterpret_cast<uint16_t*>(smem);                  // [64 o][GA_P]
  uint16_t* gsT = zsT + GA_E * GA_P;                                  // [256 i][GA_P]
  float* ct = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int o0 = blockIdx.x * GA_E, i0 = blockIdx.y * GA_J;           // row tiles fastest (see rg_g0_wgrad_adam)
  ga_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  for (int n0 = 0; n0 < N; n0 += GA_N) {
    if (n0) __syncthreads();
    // 64 samples = 128 B = eight 16-byte pieces per operand row: 512 pieces of gT, 2048 of xT
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = t + 256 * k, row = idx >> 3, pc = idx & 7;
      uint4 w = make_uint4(0, 0, 0, 0);
      if (o0 + row < O) w = *reinterpret_cast<const uint4*>(gT + (size_t)(o0 + row) * ldn + n0 + 8 * pc);
      *reinterpret_cast<uint4*>(zsT + row * GA_P + 8 * pc) = w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = t + 256 * k, row = idx >> 3, pc = idx & 7;
      uint4 w = make_uint4(0, 0, 0, 0);
      if (i0 + row < I) w = *reinterpret_cast<const uint4*>(xT + (size_t)(i0 + row) * ldn + n0 + 8 * pc);
      *reinterpret_cast<uint4*>(gsT + row * GA_P + 8 * pc) = w;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ga_bf16x8 fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[a] = __builtin_bit_cast(ga_bf16x8, *reinterpret_cast<const uint4*>(zsT + (32 * a + fr) * GA_P + 16 * ks + 8 * fh));
#pragma unroll
      for (int b = 0; b < 2; ++b)
        fb[b] = __builtin_bit_cast(ga_bf16x8,
                                   *reinterpret_cast<const uint4*>(gsT + (64 * wave + 32 * b + fr) * GA_P + 16 * ks + 8 * fh));
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = rg_mfma_h16_32x32x16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }

  const AdamC hy{hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7], hyper[8]};
  const int tj = t & 63, te = t >> 6;
  const int col = i0 + 4 * tj;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[(8 * (r >> 2) + 4 * fh + (r & 3)) * GA_CP + 64 * wave + 32 * b + fr] = acc[a][b][r];
    __syncthreads();
    // four rows at a time: their p / m / v are requested before the first update is computed (4 x 3 x 16 B in flight per
    // thread, as in g0_wgrad_adam_kernel), VPR = vectors of VEC floats per row
    constexpr int VPR = 4 / VEC;
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 4) {
      float P[4][4], M[4][4], V[4][4];
      bool ok[4][VPR];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = o0 + 32 * a + 8 * te + k0 + k;
        const size_t base = (size_t)row * I + col;
#pragma unroll
        for (int q = 0; q < VPR; ++q) {
          ok[k][q] = row < O && col + q * VEC < I;
          if (!ok[k][q]) continue;
          if constexpr (VEC == 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(p + base), b4 = *reinterpret_cast<const float4*>(m + base),
                         c4 = *reinterpret_cast<const float4*>(v + base);
            P[k][0] = a4.x; P[k][1] = a4.y; P[k][2] = a4.z; P[k][3] = a4.w;
            M[k][0] = b4.x; M[k][1] = b4.y; M[k][2] = b4.z; M[k][3] = b4.w;
            V[k][0] = c4.x; V[k][1] = c4.y; V[k][2] = c4.z; V[k][3] = c4.w;
          } else if constexpr (VEC == 2) {
            const float2 a2 = *reinterpret_cast<const float2*>(p + base + 2 * q), b2 = *reinterpret_cast<const float2*>(m + base + 2 * q),
                         c2 = *reinterpret_cast<const float2*>(v + base + 2 * q);
            P[k][2 * q] = a2.x; P[k][2 * q + 1] = a2.y; M[k][2 * q] = b2.x; M[k][2 * q + 1] = b2.y;
            V[k][2 * q] = c2.x; V[k][2 * q + 1] = c2.y;
          } else {
            P[k][q] = p[base + q]; M[k][q] = m[base + q]; V[k][q] = v[base + q];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = o0 + 32 * a + 8 * te + k0 + k;
        const size_t base = (size_t)row * I + col;
        const float* gsrc = ct + (8 * te + k0 + k) * GA_CP + 4 * tj;
#pragma unroll
        for (int q = 0; q < VPR; ++q) {
          if (!ok[k][q]) continue;
#pragma unroll
          for (int e = 0; e < VEC; ++e) adam_upd(hy, P[k][q * VEC + e], gsrc[q * VEC + e], M[k][q * VEC + e], V[k][q * VEC + e]);
          if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(p + base) = make_float4(P[k][0], P[k][1], P[k][2], P[k][3]);
            *reinterpret_cast<float4*>(m + base) = make_float4(M[k][0], M[k][1], M[k][2], M[k][3]);
            *reinterpret_cast<float4*>(v + base) = make_float4(V[k][0], V[k][1], V[k][2], V[k][3]);
            if (wpack)      // the next forward's bf16 operand image [.][Kp] of the UPDATED weight (rg_pack_linear_weight's layout)
              *reinterpret_cast<uint2*>(wpack + (size_t)row * Kp + col) =
                  make_uint2((uint32_t)f32_to_h16(P[k][0]) | ((uint32_t)f32_to_h16(P[k][1]) << 16),
                             (uint32_t)f32_to_h16(P[k][2]) | ((uint32_t)f32_to_h16(P[k][3]) << 16));
          } else if constexpr (VEC == 2) {
            *reinterpret_cast<float2*>(p + base + 2 * q) = make_float2(P[k][2 * q], P[k][2 * q + 1]);
            *reinterpret_cast<float2*>(m + base + 2 * q) = make_float2(M[k][2 * q], M[k][2 * q + 1]);
            *reinterpret_cast<float2*>(v + base + 2 * q) = make_float2(V[k][2 * q], V[k][2 * q + 1]);
            if (wpack)
              *reinterpret_cast<uint32_t*>(wpack + (size_t)row * Kp + col + 2 * q) =
                  (uint32_t)f32_to_h16(P[k][2 * q]) | ((uint32_t)f32_to_h16(P[k][2 * q + 1]) << 16);
          } else {
            p[base + q] = P[k][q]; m[base + q] = M[k][q]; v[base + q] = V[k][q];
            if (wpack) wpack[(size_t)row * Kp + col + q] = f32_to_h16(P[k][q]);
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int rg_g0_wgrad_adam_supported(int N, int E, int C, int dtype) {
  return N > 0 && E % GA_E == 0 && C % 16 == 0 && (dtype == RG_H16 || dtype == RG_F32) ? 1 : 0;
}

extern "C" int rg_g0_wgrad_adam(const float* z, const void* gz0, float* p, float* m, float* v, const float* hyper,
                                void* shadow_bf16, int N, int E, int C, int dtype, void* stream) {
  RG_REQUIRE(z && gz0 && p && m && v && hyper, RG_EINVAL, "g0_wgrad_adam: null");
  RG_REQUIRE(rg_g0_wgrad_adam_supported(N, E, C, dtype), RG_EUNSUPPORTED, "g0_wgrad_adam: E %% 64 == 0 and C %% 16 == 0 required");
  RG_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)z | (uintptr_t)gz0) & 15) == 0 &&
                 (((uintptr_t)shadow_bf16) & 7) == 0, RG_EINVAL, "g0_wgrad_adam: 16-byte aligned buffers required");
  static int order = -1;
  // measured at E = C = 2048 (tools/scratch/bench_g0adam.py), column tiles fastest / row tiles fastest: K = 64: 345 / 325 us,
  // 128: 375 / 347, 256: 490 / 397, 512: 779 / 514 -- row tiles fastest at every K (RNAGAN_G0ADAM_ORDER=0: the other order)
  if (order < 0) { const char* e = getenv("RNAGAN_G0ADAM_ORDER"); order = e ? atoi(e) : 1; }
  const int e_fastest = order != 0;
  const dim3 grid(e_fastest ? (unsigned)(E / GA_E) : (unsigned)(C / 16), e_fastest ? (unsigned)(C / 16) : (unsigned)(E / GA_E));
  hipStream_t st = rg_stream(stream);
  if (dtype == RG_H16)
    hipLaunchKernelGGL(g0_wgrad_adam_kernel<h16_t>, grid, dim3(256), 0, st, z, (const h16_t*)gz0, p, m, v, hyper,
                       (uint16_t*)shadow_bf16, N, E, C, e_fastest);
  else
    hipLaunchKernelGGL(g0_wgrad_adam_kernel<float>, grid, dim3(256), 0, st, z, (const float*)gz0, p, m, v, hyper,
                       (uint16_t*)shadow_bf16, N, E, C, e_fastest);
  RG_LAUNCH_CHECK("g0_wgrad_adam");
  return RG_OK;
}

extern "C" int rg_linear_wgrad_adam(const void* gT, const void* xT, int ldn, int N, float* p, float* m, float* v,
                                    const float* hyper, int O, int I, void* wpack_bf16, int Kp, void* stream) {
  RG_REQUIRE(gT && xT && p && m && v && hyper && N > 0 && O > 0 && I > 0, RG_EINVAL, "linear_wgrad_adam: bad args");
  RG_REQUIRE(ldn % 64 == 0 && ldn >= N && (((uintptr_t)gT | (uintptr_t)xT) & 15) == 0, RG_EINVAL,
             "linear_wgrad_adam: operands must be sample-contiguous, zero padded to a multiple of 64 samples, 16-byte aligned");
  RG_REQUIRE(!wpack_bf16 || (Kp >= I && Kp % 8 == 0 && ((uintptr_t)wpack_bf16 & 15) == 0), RG_EINVAL,
             "linear_wgrad_adam: packed image needs Kp >= I, Kp %% 8 == 0, 16-byte alignment");
  uint16_t* wpk = (uint16_t*)wpack_bf16;
  const dim3 grid((unsigned)((O + GA_E - 1) / GA_E), (unsigned)((I + GA_J - 1) / GA_J));
  hipStream_t st = rg_stream(stream);
  const uintptr_t al = (uintptr_t)p | (uintptr_t)m | (uintptr_t)v;
  const uint16_t* a = (const uint16_t*)gT;
  const uint16_t* b = (const uint16_t*)xT;
  if (I % 4 == 0 && (al & 15) == 0)
    hipLaunchKernelGGL(lin_wgrad_adam_kernel<4>, grid, dim3(256), 0, st, a, b, ldn, N, p, m, v, hyper, O, I, wpk, Kp);
  else if (I % 2 == 0 && (al & 7) == 0)
    hipLaunchKernelGGL(lin_wgrad_adam_kernel<2>, grid, dim3(256), 0, st, a, b, ldn, N, p, m, v, hyper, O, I, wpk, Kp);
  else
    hipLaunchKernelGGL(lin_wgrad_adam_kernel<1>, grid, dim3(256), 0, st, a, b, ldn, N, p, m, v, hyper, O, I, wpk, Kp);
  RG_LAUNCH_CHECK("linear_wgrad_adam");
  return RG_OK;
}
