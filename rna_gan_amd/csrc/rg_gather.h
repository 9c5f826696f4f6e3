// rg_gather.h -- argument blocks and small device helpers shared by the bf16 MFMA implicit-GEMM kernels
// (rg_mfma.hip: 2-stage LDS-DMA gather GEMM; rg_conv8.hip: 8-wave ping-pong gather GEMM).
#pragma once
#include "rg_internal.h"

namespace {

typedef rg_h16x8 h16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int MODE_DOWN = 0, MODE_UP = 1, MODE_PLAIN = 2;
// MODE_C3: 3x3 stride-1 VALID conv over a pre-padded image (the resize-convolution block of DCGANUpGenerator:
// A = materialised bilinear-x2 + reflection-pad image [N][Hs][Ws][Cin], output grid (Hs-2) x (Ws-2), 9 taps)
constexpr int MODE_C3 = 3;
// MODE_C3T: its data gradient -- full 3x3 correlation of gy[N][Hs][Ws][Cin=Cout] onto the padded grid (Hs+2) x (Ws+2)
// (row decode by division: the padded grid is not a power of two), B = w transposed to [c][tap][o]
constexpr int MODE_C3T = 4;
constexpr int EPI_BF16 = 0, EPI_LINEAR = 1;

struct GArgs {
  const uint16_t* A;
  const uint16_t* B;
  void* C;
  int M, Ncols, Cin, taps;
  int lgW, lgH;     // row m -> (n, hq, wq): wq = m & (2^lgW-1), hq = (m>>lgW) & (2^lgH-1)
  int Hs, Ws;       // spatial dims of the tensor A rows are gathered from
  int ldc;          // output row stride in elements
  int b_col, b_tap; // B operand: elements between consecutive output columns / between consecutive taps
  int tiles_n;      // number of 128-wide column tiles
  const float* scale;
  const float* shift;
  float slope;
  float* stats;           // EPI_BF16, optional: per-tile column sums of the (bf16-rounded) output and of its square,
                          // [partial rows][2][Ncols] (BatchNorm statistics straight from the conv epilogue)
  const uint16_t* mask;   // EPI_BF16, optional: activation with the output's shape; out *= (mask > 0 ? 1 : mslope)
  float mslope;           // (LeakyReLU backward of the consumer fused into the data-gradient conv)
  int in_fp8, out_fp8;    // conv8_kernel<.., EB = 1>: fp8 e4m3 operands (A, B) / fp8 output (row stride ldc in elements)
  int mask_packed;        // mask holds packed sign bits (one 64-bit word per output pixel: convp_kernel only)
  // BatchNorm-backward sums of the CONSUMER in this launch's epilogue (conv8_kernel, bf16 output without split-K): the output
  // is the data gradient ga arriving at a [BatchNorm + LeakyReLU] block whose stored pre-activation is bwd_z (the output's
  // shape); per block tile the column sums of gy = ga * lrelu'(y) and gy * xhat go to bwd_sums [rows][2][Ncols] (rows =
  // parity classes x row tiles), which rg_bn_act_bwd_partials finishes -- the separate reduction pass over (z, ga) disappears.
  const uint16_t* bwd_z;
  const float* bwd_mean; const float* bwd_invstd; const float* bwd_gamma; const float* bwd_beta;
  float bwd_slope;
  float* bwd_sums;
  int bwd_half_m;         // > 0: two batch groups -- rows m >= bwd_half_m take mean / invstd + Ncols (second group)
  int defer_reduce;       // split-K launches: leave the fp32 slabs in the workspace, do not launch the slab reduction (the consumer
                          // reduces them itself: rg_splitbn.hip fuses the reduction into the BatchNorm pass that follows)
  const float* maskf;     // EPI_LINEAR (fp32 result), optional: fp32 activation with the output's shape; out *= (maskf > 0 ? 1 : mslope)
                          // (the fp32 mode's 64-column transposed conv on bf16 planes: the consumer's LeakyReLU backward)
  int affine;             // EPI_BF16, bf16 output without split-K: out = lrelu(acc * scale[col] + shift[col], slope) (eval-mode
                          // BatchNorm folded into the conv epilogue: generator-only inference)
};

// 8 bf16 outputs (packed in o) times the LeakyReLU derivative at 8 bf16 activations (packed in a)
__device__ __forceinline__ float rg_lmask(uint32_t abits, float slope) {
  return (abits & 0x8000u) || !(abits & 0x7fffu) ? slope : 1.f;      // a <= 0 (incl. -0): slope
}

// 8 consecutive output columns: v = lrelu(v * scale[c] + shift[c], slope)
__device__ __forceinline__ void rg_affine8(float4& v0, float4& v1, const float* sc, const float* sh, float slope) {
  const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
  const float4 h0 = *reinterpret_cast<const float4*>(sh), h1 = *reinterpret_cast<const float4*>(sh + 4);
  v0.x = lrelu_f(v0.x * s0.x + h0.x, slope); v0.y = lrelu_f(v0.y * s0.y + h0.y, slope);
  v0.z = lrelu_f(v0.z * s0.z + h0.z, slope); v0.w = lrelu_f(v0.w * s0.w + h0.w, slope);
  v1.x = lrelu_f(v1.x * s1.x + h1.x, slope); v1.y = lrelu_f(v1.y * s1.y + h1.y, slope);
  v1.z = lrelu_f(v1.z * s1.z + h1.z, slope); v1.w = lrelu_f(v1.w * s1.w + h1.w, slope);
}

__device__ __forceinline__ void up_tap_dev(int par, int a, int& kidx, int& d) {
  if (par == 0) { kidx = a == 0 ? 1 : 3; d = a == 0 ? 0 : -1; }
  else          { kidx = a == 0 ? 0 : 2; d = a == 0 ? 1 : 0; }
}

// korder tap sequence of the stride-2 4x4 conv: index bits (py, px, jy, jx) -> tap (kh, kw) = (2 jy + py, 2 jx + px).  The four
// taps of one (py, px) class read the same input pixels shifted by whole output pixels, so they hit the lines the previous
// k-tile brought into L2.
__device__ __forceinline__ int rg_down_tap(int ti) {
  return ((((ti >> 1) & 1) * 2 + (ti >> 3)) << 2) | ((ti & 1) * 2 + ((ti >> 2) & 1));
}

struct G2Args {
  GArgs g;
  unsigned a_bytes, b_bytes;   // sizes for the buffer descriptors
  int nsplit;
  int xcd_swizzle;             // 1: remap blockIdx.x so each XCD (block b runs on XCD b % 8) owns a contiguous tile range
  int tiles_m;                 // number of row tiles (per parity class)
  int class_fast;              // MODE_UP: the 4 output-parity classes are the fastest-varying part of blockIdx.x (they
                               // read the same input rows: back to back on one XCD the rows are fetched from HBM once)
  float* slab;                 // [nsplit][rows_out][Ncols] fp32 when nsplit > 1 (bf16 with slab16, same element stride)
  long long slab_stride;       // elements per split
  int slab16;                  // conv8_kernel, EPI bf16: the split-K partial tiles are stored as bf16 (option slab16)
  int lgcpt, cmask;            // conv8_kernel: k-tile kt -> tap = kt >> lgcpt, channel block = kt & cmask
  int korder;                  // 1: channel-block-major k order (tap = kt & (taps-1), block = kt / taps; MODE_DOWN taps in
                               // parity-class order): the taps that revisit the same input lines are adjacent k-tiles
  int probe_iters;             // conv8_kernel<..., PROBE = 1> (rg_probe.hip): k-tiles the loop runs over the two resident stages
  unsigned a_plane, b_plane;   // conv8_kernel<..., NP > 0> (rg_conv8f.hip): bytes between consecutive bf16 planes of A / of B
};

typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// s_waitcnt immediate for vmcnt(n) only (expcnt/lgkmcnt left at their maxima); vmcnt is split over bits 3:0 and 15:14
static constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }


}  // namespace
