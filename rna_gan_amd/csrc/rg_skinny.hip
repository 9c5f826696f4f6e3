// rg_skinny.hip -- the two image-side layers (3 <-> 64 channels at 256x256).  K = 48 (or 3 output
// channels) is far from an MFMA-bound shape; these layers are HBM-bound (SURVEY 7 hard part 3).  The bf16
// path stages image rows through LDS and uses the matrix cores only to get the arithmetic out of the way
// (the VALU versions were compute-bound at 1/3 of the HBM rate); the fp32 parity path stays on the VALUs.
//   first_down : NCHW fp32 image  -> NHWC T, 4x4 s2 p1 conv (+bias, LeakyReLU)
//   last_up    : NHWC T           -> NCHW fp32 image, transposed conv (+bias, tanh)
//   wgrad      : dW[o][3][16] = sum_pix low[pix][o] * patch(pix)
#include "rg_internal.h"
#include <stdlib.h>

namespace {

constexpr int SK_I = 3;
constexpr int SK_K = SK_I * 16;   // 48

// ---------------------------------------------------------------------------------------------
// VALU kernels (the fp32 parity path, and bf16 shapes the matrix-core kernels below do not take): the fp32
// master weights are read with UNIFORM indices, so hipcc emits scalar loads (s_load_dwordx*) and the FMAs take
// the weight as an SGPR operand -- no LDS or vector-memory traffic for weights.  With T = bf16 the operands are
// rounded to bf16 first (Elem<T>::round) so that both bf16 paths compute the same thing: bf16 x bf16 products,
// fp32 accumulation.

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
        patch[ci * 16 + kh * 4 + kw] = v ? Elem<T>::round(x[(((long long)n * SK_I + ci) * H + hi) * W + wi]) : 0.f;
      }
  for (int oc = 0; oc < O; oc += OC) {
    float acc[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) {
      const float* wr = wq + (size_t)(oc + j) * SK_K;     // uniform -> scalar loads
      float a = bias ? bias[oc + j] : 0.f;
#pragma unroll
      for (int k = 0; k < SK_K; ++k) a = fmaf(patch[k], Elem<T>::round(wr[k]), a);
      acc[j] = lrelu_f(a, slope);
    }
    T* yo = y + p * O + oc;
#pragma unroll
    for (int q = 0; q < OC / 4; ++q) Vec<T, 4>::st(yo + q * 4, acc + q * 4);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 path of first_down / skinny_wgrad: row-staged patches on the matrix cores.
//
// Work unit = `chunk` (32..128) consecutive output pixels of one output row (n, ho).  Its 12 input rows
// (3 channels x 4 kh) are read with coalesced float4 loads and scattered ONCE into the transposed patch
// matrix  PT[k = ci*16 + kh*4 + kw][pixel]  (bf16) in LDS -- per-lane strided 4-byte gathers of the NCHW image
// were 60 % of the old kernels' time.  Column wi = 2*wo - 1 + kw, so an aligned float4 (4m .. 4m+3, relative
// to 2*wo0) lands as: kw=1 <- (x0, x2) at pixels 2m, 2m+1 (one packed 32-bit store), kw=2 <- (x1, x3), kw=3 <-
// x0 at 2m-1 and x2 at 2m, kw=0 <- x1 at 2m+1 and x3 at 2m+2; the two columns outside the aligned span
// (2*wo0 - 1 and 2*wo0 + 2*chunk) are loaded by 24 edge lanes.  The NEXT unit's loads are issued into registers
// before the current unit is computed (4 blocks/CU keep > 100 KB in flight per CU).
typedef rg_h16x8 sk_bf16x8;
typedef __attribute__((ext_vector_type(16))) float sk_f32x16;
typedef __attribute__((ext_vector_type(4))) short sk_s16x4;
typedef __attribute__((ext_vector_type(8))) short sk_s16x8;

__device__ __forceinline__ uint32_t sk_pack2(float a, float b) {
  return (uint32_t)f32_to_h16(a) | ((uint32_t)f32_to_h16(b) << 16);
}
__device__ __forceinline__ sk_s16x4 sk_tr_read(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sk_s16x4*)(p));
}

struct SkUnit { int n, ho, wo0; };
__device__ __forceinline__ SkUnit sk_unit(int u, int Ho, int cpr, int chunk) {
  SkUnit q;
  const int cx = u % cpr, row = u / cpr;
  q.wo0 = cx * chunk; q.ho = row % Ho; q.n = row / Ho;
  return q;
}

// registers of one unit's image rows: 3 float4 per thread (12 rows x chunk/2 float4) + one edge value
struct SkRows { float4 v0, v1, v2; float edge; };

__device__ __forceinline__ float4 sk_load4(const float* __restrict__ x, const SkUnit& q, int H, int W, int lgc, int f,
                                           int nf) {
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f >= nf) return z;
  const int row = f >> lgc, m = f & ((1 << lgc) - 1);        // row = ci*4 + kh
  const int hi = 2 * q.ho - 1 + (row & 3);
  if ((unsigned)hi >= (unsigned)H) return z;
  return *reinterpret_cast<const float4*>(x + (((size_t)q.n * SK_I + (row >> 2)) * H + hi) * W + 2 * q.wo0 + 4 * m);
}
__device__ __forceinline__ void sk_load_rows(SkRows& r, const float* __restrict__ x, const SkUnit& q, int H, int W,
                                             int chunk, int lgc, int t) {
  const int nf = 12 << lgc;
  r.v0 = sk_load4(x, q, H, W, lgc, t, nf);
  r.v1 = sk_load4(x, q, H, W, lgc, t + 256, nf);
  r.v2 = sk_load4(x, q, H, W, lgc, t + 512, nf);
  r.edge = 0.f;
  if (t < 24) {
    const int row = t >> 1, hi = 2 * q.ho - 1 + (row & 3);
    const int wi = (t & 1) ? 2 * q.wo0 + 2 * chunk : 2 * q.wo0 - 1;
    if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
      r.edge = x[(((size_t)q.n * SK_I + (row >> 2)) * H + hi) * W + wi];
  }
}
template <int PTS>
__device__ __forceinline__ void sk_scatter4(uint16_t* pt, float4 v, int lgc, int chunk, int f, int nf) {
  if (f >= nf) return;
  const int row = f >> lgc, m = f & ((1 << lgc) - 1);
  uint16_t* k0 = pt + (4 * row) * PTS + 2 * m;               // k = 4*row + kw
  *reinterpret_cast<uint32_t*>(k0 + 1 * PTS) = sk_pack2(v.x, v.z);
  *reinterpret_cast<uint32_t*>(k0 + 2 * PTS) = sk_pack2(v.y, v.w);
  k0[3 * PTS] = f32_to_h16(v.z);
  if (m > 0) k0[3 * PTS - 1] = f32_to_h16(v.x);
  k0[1] = f32_to_h16(v.y);
  if (2 * m + 2 < chunk) k0[2] = f32_to_h16(v.w);
}
__device__ __forceinline__ bool sk_pos16(uint32_t h) { return !(h & 0x8000u) && (h & 0x7fffu); }
template <int PTS>
__device__ __forceinline__ void sk_scatter_rows(uint16_t* pt, const SkRows& r, int chunk, int lgc, int t) {
  const int nf = 12 << lgc;
  sk_scatter4<PTS>(pt, r.v0, lgc, chunk, t, nf);
  sk_scatter4<PTS>(pt, r.v1, lgc, chunk, t + 256, nf);
  sk_scatter4<PTS>(pt, r.v2, lgc, chunk, t + 512, nf);
  if (t < 24) {
    const int row = t >> 1;
    if (t & 1) pt[(4 * row + 3) * PTS + chunk - 1] = f32_to_h16(r.edge);
    else       pt[(4 * row) * PTS] = f32_to_h16(r.edge);
  }
}

// ---------------------------------------------------------------------------------------------
// first_down (bf16):  D[ch][pix] = sum_k W[ch][k] * PT[k][pix].  A = weights (6 fragments in registers),
// B = transposed LDS reads of PT (rows 320 bytes apart: the 4 rows x 64 bytes one half-wave touches fall into
// disjoint bank quarters).  The 32x32 result (pixel on the lane, 16 channels in registers) gets bias +
// LeakyReLU and is transposed through a wave-private LDS tile so that every global store instruction writes
// 1 KB of consecutive NHWC bytes.
constexpr int FD_PTS = 160;
constexpr int FD_OTS = 72;      // out-tile row stride (bf16): 144 bytes

// bits (optional): packed sign bits of the activation, one 64-bit word per output pixel, bit c = (y[pixel][c] > 0) -- the
// LeakyReLU-backward mask of the data-gradient conv of layer 1 (rg_convp.hip reads 8 B per pixel instead of 128 B).
// mask_bits (optional): such bits of ANOTHER activation of the output's shape; then y *= (bit ? 1 : mslope) -- the tangent
// pass of the gradient penalty is lrelu'(a0) * conv(v), one kernel instead of the conv plus a pass over three 134 MB tensors.
template <bool BITS, bool MASK>      // compile-time presence of bits / mask_bits: the sign-bit packing is 40 % of the epilogue's VALU work
__global__ __launch_bounds__(256, 3) void first_down_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                                 unsigned long long* __restrict__ bits,
                                                                 const unsigned long long* __restrict__ mask_bits, float mslope,
                                                                 int N, int H, int W, float slope, int chunk, int nunits) {
  __shared__ __attribute__((aligned(16))) uint16_t pt[SK_K * FD_PTS];        // 15 KB
  __shared__ __attribute__((aligned(16))) uint16_t ot[4 * 32 * FD_OTS];      // 18 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Ho = H >> 1, Wo = W >> 1;
  const int cpr = Wo / chunk, lgc = 31 - __builtin_clz(chunk >> 1);
  const int nact = chunk >> 5;
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

  const int grp16 = lane >> 4, idx = lane & 15, q4 = idx >> 2, p4 = idx & 3, fh = grp16 >> 1, cb = grp16 & 1;
  uint16_t* otw = ot + wave * 32 * FD_OTS;
  const int per = (nunits + gridDim.x - 1) / gridDim.x;
  const int u0 = blockIdx.x * per, u1 = min(nunits, u0 + per);
  SkRows rows;
  if (u0 < u1) sk_load_rows(rows, x, sk_unit(u0, Ho, cpr, chunk), H, W, chunk, lgc, t);
  for (int u = u0; u < u1; ++u) {
    const SkUnit q = sk_unit(u, Ho, cpr, chunk);
    sk_scatter_rows<FD_PTS>(pt, rows, chunk, lgc, t);
    __syncthreads();
    if (u + 1 < u1) sk_load_rows(rows, x, sk_unit(u + 1, Ho, cpr, chunk), H, W, chunk, lgc, t);
    if (wave < nact) {
      sk_f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
      for (int ci = 0; ci < SK_I; ++ci) {
        const uint16_t* bp = pt + (16 * ci + 8 * fh + q4) * FD_PTS + wave * 32 + 16 * cb + 4 * p4;
        sk_s16x4 lo = sk_tr_read(bp), hi = sk_tr_read(bp + 4 * FD_PTS);
        sk_bf16x8 pb = __builtin_bit_cast(sk_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = rg_mfma_h16_32x32x16(wa[i][ci], pb, acc[i], 0, 0, 0);
      }
      // acc[i][4*g + e] = D[ch = 32*i + 8*g + 4*h + e][pixel r]
      unsigned nib = 0;                                    // nibble 4*i + g: sign bits of channels 32*i + 8*g + 4*h + 0..3
      unsigned long long mword = ~0ull;
      if (MASK) mword = mask_bits[((size_t)q.n * Ho + q.ho) * Wo + q.wo0 + wave * 32 + r];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float v0 = lrelu_f(acc[i][4 * g4 + 0] + bs[i][g4][0], slope), v1 = lrelu_f(acc[i][4 * g4 + 1] + bs[i][g4][1], slope);
          float v2 = lrelu_f(acc[i][4 * g4 + 2] + bs[i][g4][2], slope), v3 = lrelu_f(acc[i][4 * g4 + 3] + bs[i][g4][3], slope);
          if (MASK) {
            const unsigned mb = (unsigned)(mword >> (32 * i + 8 * g4 + 4 * h)) & 15u;
            v0 *= (mb & 1u) ? 1.f : mslope; v1 *= (mb & 2u) ? 1.f : mslope;
            v2 *= (mb & 4u) ? 1.f : mslope; v3 *= (mb & 8u) ? 1.f : mslope;
          }
          const uint32_t p01 = sk_pack2(v0, v1), p23 = sk_pack2(v2, v3);
          *reinterpret_cast<uint2*>(otw + r * FD_OTS + 32 * i + 8 * g4 + 4 * h) = make_uint2(p01, p23);
          // bit = the STORED bf16 value is > 0 (sign clear, magnitude non-zero): what rg_lmask tests on the bf16 activation
          if (BITS) {
            const unsigned b = (sk_pos16(p01) ? 1u : 0u) | (sk_pos16(p01 >> 16) ? 2u : 0u) | (sk_pos16(p23) ? 4u : 0u) |
                               (sk_pos16(p23 >> 16) ? 8u : 0u);
            nib |= b << (4 * (4 * i + g4));
          }
        }
      if (BITS) {
        // lanes r and r + 32 hold the low (h = 0) / high (h = 1) nibble of every byte of pixel r's word
        const unsigned other = (unsigned)__shfl_xor((int)nib, 32, 64);
        const unsigned lo = h ? other : nib, hi = h ? nib : other;
        unsigned long long word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          word |= ((unsigned long long)(((lo >> (4 * k)) & 15u) | (((hi >> (4 * k)) & 15u) << 4))) << (8 * k);
        if (h == 0) bits[((size_t)q.n * Ho + q.ho) * Wo + q.wo0 + wave * 32 + r] = word;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private tile: in-order DS pipe, no barrier needed
      uint16_t* yo = y + (((size_t)q.n * Ho + q.ho) * Wo + q.wo0 + wave * 32) * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sidx = j * 64 + lane, px = sidx >> 3, c = sidx & 7;
        *reinterpret_cast<uint4*>(yo + px * 64 + c * 8) = *reinterpret_cast<const uint4*>(otw + px * FD_OTS + c * 8);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// skinny weight gradient (bf16 `low`):  dW[o][k] = sum_pix low[pix][o] * P[pix][k]; the contraction index is
// the pixel.  Per unit the block stages PT (above; rows 272 bytes apart so the 16 rows of a ds_read_b128
// lane group hit disjoint banks) and the [pixel][64 ch] tile of `low` (coalesced 16-byte loads; 16-byte chunk
// c of pixel p sits at slot c ^ 4*((p>>1)&1), which makes the transposed reads conflict-free).  A = low^T via
// ds_read_b64_tr_b16, B = PT via ds_read_b128 (8 consecutive pixels of one k row); 2 k-steps x 4 MFMAs per
// wave and unit.  The 4 waves' accumulators are summed through LDS at the end; one fp32 slab per block,
// reduced deterministically afterwards.
constexpr int SW_PTS = 136;

__global__ __launch_bounds__(256, 3) void skinny_wgrad_rows_kernel(const uint16_t* __restrict__ low,
                                                                   const float* __restrict__ x, float* __restrict__ slab,
                                                                   int N, int H, int W, int chunk, int nunits) {
  __shared__ __attribute__((aligned(16))) uint16_t pt[SK_K * SW_PTS];        // 12.75 KB
  __shared__ __attribute__((aligned(16))) uint16_t lt[128 * 64];             // 16 KB (reused for the final reduction)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Ho = H >> 1, Wo = W >> 1;
  const int cpr = Wo / chunk, lgc = 31 - __builtin_clz(chunk >> 1);
  const int nact = chunk >> 5;

  sk_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int grp16 = lane >> 4, idx = lane & 15, q4 = idx >> 2, p4 = idx & 3, fh = grp16 >> 1, cb = grp16 & 1;
  const int per = (nunits + gridDim.x - 1) / gridDim.x;
  const int u0 = blockIdx.x * per, u1 = min(nunits, u0 + per);
  SkRows rows;
  uint4 lv0, lv1, lv2, lv3;
  // (pixel, 16-byte chunk) = (sidx >> 3, sidx & 7), sidx = j*256 + t
#define SK_LOAD_LOW(q)                                                                                       \
  do {                                                                                                       \
    const uint16_t* src_ = low + (((size_t)(q).n * Ho + (q).ho) * Wo + (q).wo0) * 64 + (size_t)t * 8;         \
    lv0 = lv1 = lv2 = lv3 = make_uint4(0, 0, 0, 0);    /* if-assign, not ?: (a ?: of lvalues selects POINTERS) */ \
    if ((t >> 3) < chunk) lv0 = *reinterpret_cast<const uint4*>(src_);                                       \
    if ((t >> 3) + 32 < chunk) lv1 = *reinterpret_cast<const uint4*>(src_ + 2048);                           \
    if ((t >> 3) + 64 < chunk) lv2 = *reinterpret_cast<const uint4*>(src_ + 4096);                           \
    if ((t >> 3) + 96 < chunk) lv3 = *reinterpret_cast<const uint4*>(src_ + 6144);                           \
  } while (0)
  if (u0 < u1) {
    const SkUnit q = sk_unit(u0, Ho, cpr, chunk);
    sk_load_rows(rows, x, q, H, W, chunk, lgc, t);
    SK_LOAD_LOW(q);
  }
  for (int u = u0; u < u1; ++u) {
    sk_scatter_rows<SW_PTS>(pt, rows, chunk, lgc, t);
    {
      // pixel px = (t >> 3) + 32 j: (px >> 1) & 1 does not depend on j
      const int px = t >> 3, c = t & 7;
      uint16_t* d = lt + px * 64 + ((c ^ (((px >> 1) & 1) << 2)) << 3);
      *reinterpret_cast<uint4*>(d) = lv0;
      *reinterpret_cast<uint4*>(d + 32 * 64) = lv1;
      *reinterpret_cast<uint4*>(d + 64 * 64) = lv2;
      *reinterpret_cast<uint4*>(d + 96 * 64) = lv3;
    }
    __syncthreads();
    if (u + 1 < u1) {
      const SkUnit q = sk_unit(u + 1, Ho, cpr, chunk);
      sk_load_rows(rows, x, q, H, W, chunk, lgc, t);
      SK_LOAD_LOW(q);
    }
    if (wave < nact) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int prow = wave * 32 + ks * 16 + 8 * fh + q4;
        const int sw = ((prow >> 1) & 1) << 2;                 // same for prow + 4
        sk_bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = 4 * i + 2 * cb + (p4 >> 1);
          const uint16_t* ap = lt + prow * 64 + ((c ^ sw) << 3) + (p4 & 1) * 4;
          sk_s16x4 lo = sk_tr_read(ap), hi = sk_tr_read(ap + 4 * 64);
          fa[i] = __builtin_bit_cast(sk_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          const int k = 32 * i + r;
          fb[i] = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(pt + (k < SK_K ? k : 0) * SW_PTS +
                                                                                wave * 32 + ks * 16 + 8 * h));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = rg_mfma_h16_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // block reduction of the 4 waves: acc[i][j][reg] = D[o = 32i + (reg&3)+8(reg>>2)+4h][k = 32j + r]
  float* red = reinterpret_cast<float*>(lt);
  for (int i = t; i < 64 * SK_K; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
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
  for (int i = t; i < 64 * SK_K; i += 256) sl[i] = red[i];
#undef SK_LOAD_LOW
}

// ---------------------------------------------------------------------------------------------
// last_up (bf16 input) on the matrix cores.  out[n][i][2hq+ph][2wq+pw] = sum over the 3x3 neighbourhood
// (dh, dw) of x[n][hq-1+dh][wq-1+dw][o] * w[o][i][kh][kw] with kh = ph + 3 - 2dh, kw = pw + 3 - 2dw (taps outside
// 0..3 do not exist).  As a GEMM: rows = 16 consecutive low-res pixels of one row, columns = (i, ph, pw) (12 of 16
// used), K = 9 shifts x 64 channels; v_mfma_f32_16x16x32_bf16, 18 per 16-pixel tile.  The B operand (weights,
// zero where the tap does not exist) is built once per block and lives in 72 VGPRs.  A block walks a strip of
// consecutive rows of one image keeping a 3-row ring of the NHWC input in LDS (pixels 144 bytes apart, so the
// 16-byte fragment reads of 16 neighbouring pixels spread over the banks): every input row is read from HBM once
// per strip, with the next row's loads in flight during the current row's MFMAs.  The 16x16 results get bias +
// tanh and go through an LDS tile so that the NCHW rows are written with 16-byte coalesced stores.
typedef __attribute__((ext_vector_type(4))) float sk_f32x4;
// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units (5 instructions instead of libm's ~40; 12.6 M evaluations per
// 64-image batch): absolute error <= 2e-7 over the whole range (saturates to +-1 through exp2 -> inf / 0), which is what an
// image in [-1, 1] needs; the relative error for |x| < 1e-3 is that of the cancellation, ~1e-4.
__device__ __forceinline__ float sk_fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);       // exp(2x)
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
constexpr int LU_PXS = 72;                 // bf16 elements per staged pixel (144 bytes)
constexpr int LU_MAXC = 128;               // pixels per unit row
constexpr int LU_OS = 2 * LU_MAXC + 4;     // floats per staged output row

// pre (optional): the input is the PRE-BatchNorm tensor z and the layer's train-mode BatchNorm + LeakyReLU is applied while
// a row is staged -- lrelu((z - mean) * (invstd * gamma) + beta), rounded to bf16 exactly as rg_bn_act would store it -- so a
// generator forward that keeps nothing for a backward pass (the fakes of the D-loss and penalty steps) skips the BatchNorm
// apply pass over its largest activation (134 MB read + 134 MB written).
struct LuPre { const float* mean; const float* invstd; const float* gamma; const float* beta; float slope; };
// post (optional), applied to the output rows on their way from the LDS tile to memory -- the first consumer's pass fused away:
//   tb_img: out *= 1 - tb_img^2, tb_img = the generator's image (tanh backward: the data gradient of D's layer 0 in the
//           generator-loss step IS the cotangent of G's tanh output; rg_tanh_bwd's arithmetic, 100 MB less traffic);
//   part:   float[gridDim.x][4] per-workgroup sums of the rows this workgroup wrote: channel 0, 1, 2 and the sum of squares
//           (the bias gradient of G's last layer = the channel sums; the penalty's ||d D / d xhat||^2 = the sum of squares).
struct LuPost { const float* tb_img; float* part; };
__global__ __launch_bounds__(256, 2) void last_up_rows_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int N,
                                                              int Ho, int Wo, int apply_tanh, int chunk, int strip,
                                                              int nstrips, LuPre pre, LuPost post) {
  __shared__ __attribute__((aligned(16))) uint16_t ring[3 * (LU_MAXC + 2) * LU_PXS];    // 54.8 KB
  __shared__ __attribute__((aligned(16))) float outt[6 * LU_OS];                          // 6.1 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lp = lane & 15, lq = lane >> 4;
  const int cpr = Wo / chunk, spi = Ho / strip;            // column chunks per row, strips per image
  const int H = 2 * Ho, W = 2 * Wo;
  // column n = i*4 + ph*2 + pw of the B operand
  const int ni = lp >> 2, nph = (lp >> 1) & 1, npw = lp & 1;
  sk_bf16x8 wf[3][3][2];
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int kh = nph + 3 - 2 * dh, kw = npw + 3 - 2 * dw;
      const bool ok = ni < SK_I && (unsigned)kh < 4u && (unsigned)kw < 4u;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = w[(size_t)(32 * c + 8 * lq + j) * SK_K + ni * 16 + kh * 4 + kw];
        }
        uint4 pk = make_uint4(sk_pack2(v[0], v[1]), sk_pack2(v[2], v[3]), sk_pack2(v[4], v[5]), sk_pack2(v[6], v[7]));
        wf[dh][dw][c] = __builtin_bit_cast(sk_bf16x8, pk);
      }
    }
  const float bv = (bias && ni < SK_I) ? bias[ni] : 0.f;
  // BatchNorm parameters of this thread's 8 input channels (t & 7) * 8 .. + 7
  float pm[8], pr[8], pb[8];
  if (pre.mean) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (t & 7) * 8 + j;
      pm[j] = pre.mean[c]; pr[j] = pre.invstd[c] * pre.gamma[c]; pb[j] = pre.beta[c];
    }
  }
  auto bn8 = [&](uint4& v) {
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = lrelu_f((h16_to_f32((uint16_t)d[j]) - pm[2 * j]) * pr[2 * j] + pb[2 * j], pre.slope);
      const float hi = lrelu_f((h16_to_f32((uint16_t)(d[j] >> 16)) - pm[2 * j + 1]) * pr[2 * j + 1] + pb[2 * j + 1], pre.slope);
      d[j] = sk_pack2(lo, hi);
    }
    v = make_uint4(d[0], d[1], d[2], d[3]);
  };
  bool row_ok = false, edge_ok = false;            // validity of the row / halo pixel held in lv*, le (padding stays zero)
  float pc0 = 0.f, pc1 = 0.f, pc2 = 0.f, pq = 0.f; // post.part: this thread's share

  for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
    const int cx = sidx % cpr, rest = sidx / cpr;
    const int hq0 = (rest % spi) * strip, n = rest / spi;
    const int wo0 = cx * chunk;
    const uint16_t* xn = x + (size_t)n * Ho * Wo * 64;
    uint4 lv0, lv1, lv2, lv3, le;
    // one input row -> registers: (pixel, 16-byte chunk) = (t >> 3 (+32 j), t & 7); 16 edge lanes take the two halo pixels
#define LU_LOAD_ROW(row)                                                                                      \
  do {                                                                                                        \
    lv0 = lv1 = lv2 = lv3 = le = make_uint4(0, 0, 0, 0);                                                      \
    row_ok = (unsigned)(row) < (unsigned)Ho;                                                                  \
    edge_ok = false;                                                                                          \
    if ((unsigned)(row) < (unsigned)Ho) {                                                                     \
      const uint16_t* src_ = xn + ((size_t)(row) * Wo + wo0) * 64 + (size_t)t * 8;                            \
      if ((t >> 3) < chunk) lv0 = *reinterpret_cast<const uint4*>(src_);                                      \
      if ((t >> 3) + 32 < chunk) lv1 = *reinterpret_cast<const uint4*>(src_ + 2048);                          \
      if ((t >> 3) + 64 < chunk) lv2 = *reinterpret_cast<const uint4*>(src_ + 4096);                          \
      if ((t >> 3) + 96 < chunk) lv3 = *reinterpret_cast<const uint4*>(src_ + 6144);                          \
      if (t < 16) {                                                                                           \
        const int col_ = (t >> 3) ? wo0 + chunk : wo0 - 1;                                                    \
        if ((unsigned)col_ < (unsigned)Wo) {                                                                  \
          le = *reinterpret_cast<const uint4*>(xn + ((size_t)(row) * Wo + col_) * 64 + (t & 7) * 8);          \
          edge_ok = true;                                                                                     \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)
    LU_LOAD_ROW(hq0 - 1);
    for (int k = 0; k < strip + 2; ++k) {
      uint16_t* slot = ring + (k % 3) * ((LU_MAXC + 2) * LU_PXS);
      {
        if (pre.mean && row_ok) {
          if ((t >> 3) < chunk) bn8(lv0);
          if ((t >> 3) + 32 < chunk) bn8(lv1);
          if ((t >> 3) + 64 < chunk) bn8(lv2);
          if ((t >> 3) + 96 < chunk) bn8(lv3);
          if (t < 16 && edge_ok) bn8(le);
        }
        uint16_t* d = slot + (1 + (t >> 3)) * LU_PXS + (t & 7) * 8;
        if ((t >> 3) < chunk) *reinterpret_cast<uint4*>(d) = lv0;
        if ((t >> 3) + 32 < chunk) *reinterpret_cast<uint4*>(d + 32 * LU_PXS) = lv1;
        if ((t >> 3) + 64 < chunk) *reinterpret_cast<uint4*>(d + 64 * LU_PXS) = lv2;
        if ((t >> 3) + 96 < chunk) *reinterpret_cast<uint4*>(d + 96 * LU_PXS) = lv3;
        if (t < 16) *reinterpret_cast<uint4*>(slot + ((t >> 3) ? chunk + 1 : 0) * LU_PXS + (t & 7) * 8) = le;
      }
      __syncthreads();
      if (k + 1 < strip + 2) LU_LOAD_ROW(hq0 + k);
      if (k >= 2) {
        const int hq = hq0 + k - 2;
        const uint16_t* s0 = ring + ((k - 2) % 3) * ((LU_MAXC + 2) * LU_PXS);
        const uint16_t* s1 = ring + ((k - 1) % 3) * ((LU_MAXC + 2) * LU_PXS);
        const uint16_t* s2 = slot;
        for (int st = wave; st < (chunk >> 4); st += 4) {
          sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const int aoff = (16 * st + lp) * LU_PXS + 8 * lq;
#pragma unroll
          for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int o = aoff + dw * LU_PXS + 32 * c;
              sk_bf16x8 a0 = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(s0 + o));
              sk_bf16x8 a1 = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(s1 + o));
              sk_bf16x8 a2 = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(s2 + o));
              acc = rg_mfma_h16_16x16x32(a0, wf[0][dw][c], acc, 0, 0, 0);
              acc = rg_mfma_h16_16x16x32(a1, wf[1][dw][c], acc, 0, 0, 0);
              acc = rg_mfma_h16_16x16x32(a2, wf[2][dw][c], acc, 0, 0, 0);
            }
          // acc[r] = D[pixel 16 st + 4 lq + r][column lp]
          if (ni < SK_I) {
            float* orow = outt + (ni * 2 + nph) * LU_OS + 2 * (16 * st + 4 * lq) + npw;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[r] + bv;
              if (apply_tanh) v = sk_fast_tanh(v);
              orow[2 * r] = v;
            }
          }
        }
        __syncthreads();
        const int lgh = 31 - __builtin_clz(chunk >> 1);          // float4 per output row = chunk / 2
        for (int f = t; f < (6 << lgh); f += 256) {
          const int rid = f >> lgh, m = f & ((1 << lgh) - 1);
          const int i = rid >> 1, ph = rid & 1;
          const size_t oi = (((size_t)n * SK_I + i) * H + 2 * hq + ph) * W + 2 * wo0 + 4 * m;
          float4 v = *reinterpret_cast<const float4*>(outt + rid * LU_OS + 4 * m);
          if (post.tb_img) {
            const float4 im = *reinterpret_cast<const float4*>(post.tb_img + oi);
            v = make_float4(v.x * (1.f - im.x * im.x), v.y * (1.f - im.y * im.y), v.z * (1.f - im.z * im.z),
                            v.w * (1.f - im.w * im.w));
          }
          *reinterpret_cast<float4*>(y + oi) = v;
          if (post.part) {
            const float sv = (v.x + v.y) + (v.z + v.w);
            pc0 += i == 0 ? sv : 0.f; pc1 += i == 1 ? sv : 0.f; pc2 += i == 2 ? sv : 0.f;
            pq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
        }
      }
      __syncthreads();
    }
#undef LU_LOAD_ROW
  }
  if (post.part) {          // fixed-order block sums (wave butterfly, then the four waves through LDS): deterministic
    float vals[4] = {pc0, pc1, pc2, pq};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) vals[k] += __shfl_xor(vals[k], o, 64);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) outt[wave * 4 + k] = vals[k];
    }
    __syncthreads();
    if (t < 4) post.part[(size_t)blockIdx.x * 4 + t] = (outt[t] + outt[4 + t]) + (outt[8 + t] + outt[12 + t]);
  }
}

// sums of the per-workgroup rows written by last_up_rows_kernel (post.part): out[c] (+)= sum_b part[b][c] for c < 3 (mode 0),
// or the penalty's coefficient from sq = sum_b part[b][3] (mode 1: gp_coef_kernel's arithmetic)
__global__ __launch_bounds__(256) void lu_part_final_kernel(const float* __restrict__ part, int nb, float* out, int accumulate,
                                                            int mode, float* loss, float* coef, float lambd, float in_inv,
                                                            float out_scale) {
  __shared__ float sm[4][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = t; b < nb; b += 256) {
    const float4 p = *reinterpret_cast<const float4*>(part + (size_t)b * 4);
    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[wave][k] = v[k];
  }
  __syncthreads();
  if (t < 4) {
    const float tot = (sm[0][t] + sm[1][t]) + (sm[2][t] + sm[3][t]);
    if (mode == 0) {
      if (t < 3) out[t] = accumulate ? out[t] + tot : tot;
    } else if (t == 3) {
      const float nrm = sqrtf(tot) * in_inv;                 // (in_inv, out_scale: gp_coef_kernel, rg_misc.hip)
      if (out) out[0] = tot * in_inv * in_inv;
      loss[0] = (nrm - 1.f) * (nrm - 1.f);
      coef[0] = lambd * 2.f * (nrm - 1.f) / nrm * (in_inv * out_scale);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// last_up: block = 8 x 32 low-res positions; the (8+2) x (32+2) halo tile of the NHWC input is staged
// once in LDS (16-byte coalesced loads, rows padded by 8 bytes -> conflict-free 8-byte reads with one
// lane per pixel); each thread produces its 2x2 output quad x 3 channels and stores float2 pairs that
// are contiguous across the wave (NCHW rows).
constexpr int LU_TW = 32;
// CCH > 0: the channels are staged CCH at a time (fp32 storage: the whole 64-channel halo tile is 36 KB per 64 threads, i.e.
// one wave per SIMD and a latency-bound kernel -- 541 us for 6.4 GFLOP; 16 channels at a time are 15 KB per 128 threads)
template <typename T, int LU_TH, int CCH = 0>
__global__ __launch_bounds__(LU_TH * 32) void last_up_kernel(const T* __restrict__ x, const float* __restrict__ wq,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N,
                                                      int Ho, int Wo, int O, int apply_tanh) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int CH = CCH ? CCH : O;                            // channels per staging pass
  const int rowb = CH * (int)sizeof(T) + 8;                // bytes per staged pixel (padded)
  const int t = threadIdx.x;
  const int tiles_w = (Wo + LU_TW - 1) / LU_TW, tiles_h = (Ho + LU_TH - 1) / LU_TH;
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w;
  const int th = b % tiles_h;
  const int n = b / tiles_h;
  const int h0 = th * LU_TH - 1, w0 = tw * LU_TW - 1;      // halo origin
  const T* xb = x + (long long)n * Ho * Wo * O;
  const int chunks = CH * (int)sizeof(T) / 16;
  const int npx = (LU_TH + 2) * (LU_TW + 2);
  const int lh = t / LU_TW, lw = t % LU_TW;
  const int hq = th * LU_TH + lh, wq_ = tw * LU_TW + lw;
  float acc[2][2][SK_I];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < SK_I; ++i) acc[a][c][i] = bias ? bias[i] : 0.f;
  for (int c0 = 0; c0 < O; c0 += CH) {
  if (c0) __syncthreads();                                 // the previous pass's readers are done
  // stage: (LU_TH+2)*(LU_TW+2) pixels x CH channels, 16 bytes per lane
  for (int i = t; i < npx * chunks; i += LU_TH * 32) {
    int px = i / chunks, ch = i - px * chunks;
    int hh = h0 + px / (LU_TW + 2), ww = w0 + px % (LU_TW + 2);
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((unsigned)hh < (unsigned)Ho && (unsigned)ww < (unsigned)Wo)
      v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(xb + ((long long)hh * Wo + ww) * O + c0) + ch * 16);
    // 8-byte stores keep the padded rows 8-byte aligned
    uint2* d = reinterpret_cast<uint2*>(smem + px * rowb + ch * 16);
    d[0] = make_uint2(v.x, v.y);
    d[1] = make_uint2(v.z, v.w);
  }
  __syncthreads();
  for (int o0 = 0; o0 < CH; o0 += 4) {
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
      const float* wr = wq + (size_t)(c0 + o0 + oo) * SK_K;      // uniform -> scalar loads
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
                acc[ph][pw][i] = fmaf(xval, Elem<T>::round(wr[i * 16 + kh * 4 + kw]), acc[ph][pw][i]);
            }
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

// ---------------------------------------------------------------------------------------------
// The three row-staged kernels again for the benchmark's geometry: 256 x 256 images, so an output / low-resolution row is
// Wo = 128 pixels = ONE unit, no column chunks, the halo columns are always padding.  Same arithmetic, same LDS images and the
// same summation order as the kernels above (bit-identical results; tests/test_fullsize_gpu.py compares the two forms), but a
// loop body WITHOUT control flow around vector-memory instructions.  In the general kernels every load sits in an exec-masked
// branch (chunk width, image border, edge lanes); hipcc's waitcnt insertion merges the counter state over those branches
// conservatively and the ISA shows `s_waitcnt vmcnt(0)` in front of each unit's loads and between them: every unit drained
// its own output STORES (and loaded its rows one after the other) before the next unit's loads were issued -- 3.3-3.5 TB/s.
// Here rows outside the image are read from a clamped address and zeroed by a select when they are staged, loads are issued
// unconditionally PF units ahead, and the waits in front of a unit's staging are counted (`vmcnt(N)` with the younger loads
// and stores left in flight).
constexpr int R128_H = 256, R128_W = 256, R128_HO = 128, R128_WO = 128;

struct Fd128Rows { float4 v[3]; };
// thread (wave kh, lane m): input row 2*ho - 1 + kh of the three channels, columns 4m .. 4m + 3
__device__ __forceinline__ void fd128_load(Fd128Rows& r, const float* __restrict__ x, int u, int wave, int lane) {
  const int n = u >> 7, ho = u & 127;
  int hi = 2 * ho - 1 + wave;
  hi = hi < 0 ? 0 : (hi > R128_H - 1 ? R128_H - 1 : hi);
  const float* p = x + ((size_t)n * SK_I * R128_H + hi) * R128_W + 4 * lane;
  r.v[0] = *reinterpret_cast<const float4*>(p);
  r.v[1] = *reinterpret_cast<const float4*>(p + R128_H * R128_W);
  r.v[2] = *reinterpret_cast<const float4*>(p + 2 * R128_H * R128_W);
}
__device__ __forceinline__ bool fd128_row_ok(int u, int wave) { return (unsigned)(2 * (u & 127) - 1 + wave) < (unsigned)R128_H; }
template <int PTS>
__device__ __forceinline__ void fd128_scatter(uint16_t* pt, const Fd128Rows& r, bool ok, int wave, int lane) {
#pragma unroll
  for (int ci = 0; ci < SK_I; ++ci) {
    float4 v = r.v[ci];
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    uint16_t* k0 = pt + (4 * (4 * ci + wave)) * PTS + 2 * lane;         // k = 4*row + kw, row = ci*4 + kh
    *reinterpret_cast<uint32_t*>(k0 + 1 * PTS) = sk_pack2(v.x, v.z);
    *reinterpret_cast<uint32_t*>(k0 + 2 * PTS) = sk_pack2(v.y, v.w);
    k0[3 * PTS] = f32_to_h16(v.z);
    if (lane > 0) k0[3 * PTS - 1] = f32_to_h16(v.x);
    k0[1] = f32_to_h16(v.y);
    if (lane < 63) k0[2] = f32_to_h16(v.w);
  }
}
// the two patch columns outside the image (kw = 0 at pixel 0, kw = 3 at pixel 127) are never written by fd128_scatter
template <int PTS>
__device__ __forceinline__ void fd128_zero_edges(uint16_t* pt, int t) {
  if (t < 24) {
    const int row = t >> 1;
    if (t & 1) pt[(4 * row + 3) * PTS + R128_WO - 1] = 0;
    else       pt[(4 * row) * PTS] = 0;
  }
}

template <bool BITS, bool MASK>
__global__ __launch_bounds__(256, 3) void first_down_rows128_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                                    unsigned long long* __restrict__ bits,
                                                                    const unsigned long long* __restrict__ mask_bits,
                                                                    float mslope, float slope, int nunits) {
  __shared__ __attribute__((aligned(16))) uint16_t pt[2][SK_K * FD_PTS];     // 2 x 15 KB: one barrier per unit
  __shared__ __attribute__((aligned(16))) uint16_t ot[4 * 32 * FD_OTS];      // 18 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
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
  float bs[2][4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int e = 0; e < 4; ++e) bs[i][g4][e] = bias ? bias[32 * i + 8 * g4 + 4 * h + e] : 0.f;
  const int grp16 = lane >> 4, idx = lane & 15, q4 = idx >> 2, p4 = idx & 3, fh = grp16 >> 1, cb = grp16 & 1;
  uint16_t* otw = ot + wave * 32 * FD_OTS;
  const int per = (nunits + gridDim.x - 1) / gridDim.x;
  const int u0 = blockIdx.x * per, u1 = min(nunits, u0 + per);
  if (u0 >= u1) return;
  fd128_zero_edges<FD_PTS>(pt[0], t);
  fd128_zero_edges<FD_PTS>(pt[1], t);
  Fd128Rows ra, rb;                       // units u and u + 1 in flight (no register copies: a unit's registers are re-loaded
                                          // for unit u + 2 right behind the barrier that follows their staging)
  fd128_load(ra, x, u0, wave, lane);
  fd128_load(rb, x, min(u0 + 1, u1 - 1), wave, lane);
  auto step = [&](int u, Fd128Rows& cur, uint16_t* ptu) __attribute__((always_inline)) {
    fd128_scatter<FD_PTS>(ptu, cur, fd128_row_ok(u, wave), wave, lane);
    __syncthreads();
    unsigned long long mword = ~0ull;
    if (MASK) mword = mask_bits[(size_t)u * R128_WO + wave * 32 + r];   // in front of the row loads: waited for with them in flight
    fd128_load(cur, x, min(u + 2, u1 - 1), wave, lane);
    sk_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int ci = 0; ci < SK_I; ++ci) {
      const uint16_t* bp = ptu + (16 * ci + 8 * fh + q4) * FD_PTS + wave * 32 + 16 * cb + 4 * p4;
      sk_s16x4 lo = sk_tr_read(bp), hi = sk_tr_read(bp + 4 * FD_PTS);
      sk_bf16x8 pb = __builtin_bit_cast(sk_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = rg_mfma_h16_32x32x16(wa[i][ci], pb, acc[i], 0, 0, 0);
    }
    unsigned nib = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float v0 = lrelu_f(acc[i][4 * g4 + 0] + bs[i][g4][0], slope), v1 = lrelu_f(acc[i][4 * g4 + 1] + bs[i][g4][1], slope);
        float v2 = lrelu_f(acc[i][4 * g4 + 2] + bs[i][g4][2], slope), v3 = lrelu_f(acc[i][4 * g4 + 3] + bs[i][g4][3], slope);
        if (MASK) {
          const unsigned mb = (unsigned)(mword >> (32 * i + 8 * g4 + 4 * h)) & 15u;
          v0 *= (mb & 1u) ? 1.f : mslope; v1 *= (mb & 2u) ? 1.f : mslope;
          v2 *= (mb & 4u) ? 1.f : mslope; v3 *= (mb & 8u) ? 1.f : mslope;
        }
        const uint32_t p01 = sk_pack2(v0, v1), p23 = sk_pack2(v2, v3);
        *reinterpret_cast<uint2*>(otw + r * FD_OTS + 32 * i + 8 * g4 + 4 * h) = make_uint2(p01, p23);
        if (BITS) {
          const unsigned b = (sk_pos16(p01) ? 1u : 0u) | (sk_pos16(p01 >> 16) ? 2u : 0u) | (sk_pos16(p23) ? 4u : 0u) |
                             (sk_pos16(p23 >> 16) ? 8u : 0u);
          nib |= b << (4 * (4 * i + g4));
        }
      }
    if (BITS) {
      const unsigned other = (unsigned)__shfl_xor((int)nib, 32, 64);
      const unsigned lo = h ? other : nib, hi = h ? nib : other;
      unsigned long long word = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        word |= ((unsigned long long)(((lo >> (4 * k)) & 15u) | (((hi >> (4 * k)) & 15u) << 4))) << (8 * k);
      // every lane stores (lanes r and r + 32 the same word to the same address): no exec-masked branch around the store
      bits[(size_t)u * R128_WO + wave * 32 + r] = word;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private tile: in-order DS pipe, no barrier needed
    uint16_t* yo = y + ((size_t)u * R128_WO + wave * 32) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sidx = j * 64 + lane, px = sidx >> 3, c = sidx & 7;
      *reinterpret_cast<uint4*>(yo + px * 64 + c * 8) = *reinterpret_cast<const uint4*>(otw + px * FD_OTS + c * 8);
    }
  };
  // The first pair is peeled so that the loop is ENTERED with the same vector-memory operations pending as at its back edge
  // (loads, stores, loads, stores): hipcc merges the two states, and with only the prologue's loads pending at the entry it
  // would wait for the younger unit's loads before staging the older one (vmcnt(5) instead of vmcnt(15) in the ISA).
  int u = u0;
  if (u + 1 < u1) {
    step(u, ra, pt[0]);
    step(u + 1, rb, pt[1]);
    u += 2;
#pragma unroll 1
    for (; u + 1 < u1; u += 2) {
      step(u, ra, pt[0]);
      step(u + 1, rb, pt[1]);
    }
  }
  if (u < u1) step(u, ra, pt[0]);          // odd tail outside the loop: the loop body stays free of control flow
}

// last_up at Ho = Wo = 128: a strip = 16 whole low-resolution rows (+ a halo row above and below); the halo PIXELS left and
// right of a row are padding (zeroed once per ring slot).  Two input rows in flight in registers behind the one being staged.
struct Lu128Row { uint4 v[4]; };
__device__ __forceinline__ void lu128_load(Lu128Row& r, const uint16_t* __restrict__ xn, int row, int t) {
  row = row < 0 ? 0 : (row > R128_HO - 1 ? R128_HO - 1 : row);
  const uint16_t* src = xn + (size_t)row * R128_WO * 64 + (size_t)t * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.v[j] = *reinterpret_cast<const uint4*>(src + 2048 * j);
}

template <bool PRE, bool TB, bool PART>
__global__ __launch_bounds__(256, 2) void last_up_rows128_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 int apply_tanh, int nstrips, LuPre pre, LuPost post) {
  constexpr int Ho = R128_HO, Wo = R128_WO, H = R128_H, W = R128_W, STRIP = 16, SLOT = (LU_MAXC + 2) * LU_PXS;
  __shared__ __attribute__((aligned(16))) uint16_t ring[3 * SLOT];                       // 54.8 KB
  __shared__ __attribute__((aligned(16))) float outt[6 * LU_OS];                          // 6.1 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lp = lane & 15, lq = lane >> 4;
  const int ni = lp >> 2, nph = (lp >> 1) & 1, npw = lp & 1;
  sk_bf16x8 wf[3][3][2];
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int kh = nph + 3 - 2 * dh, kw = npw + 3 - 2 * dw;
      const bool ok = ni < SK_I && (unsigned)kh < 4u && (unsigned)kw < 4u;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = w[(size_t)(32 * c + 8 * lq + j) * SK_K + ni * 16 + kh * 4 + kw];
        }
        uint4 pk = make_uint4(sk_pack2(v[0], v[1]), sk_pack2(v[2], v[3]), sk_pack2(v[4], v[5]), sk_pack2(v[6], v[7]));
        wf[dh][dw][c] = __builtin_bit_cast(sk_bf16x8, pk);
      }
    }
  const float bv = (bias && ni < SK_I) ? bias[ni] : 0.f;
  float pm[8], pr[8], pb[8];
  if (PRE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (t & 7) * 8 + j;
      pm[j] = pre.mean[c]; pr[j] = pre.invstd[c] * pre.gamma[c]; pb[j] = pre.beta[c];
    }
  }
  auto bn8 = [&](uint4& v) {
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = lrelu_f((h16_to_f32((uint16_t)d[j]) - pm[2 * j]) * pr[2 * j] + pb[2 * j], pre.slope);
      const float hi = lrelu_f((h16_to_f32((uint16_t)(d[j] >> 16)) - pm[2 * j + 1]) * pr[2 * j + 1] + pb[2 * j + 1], pre.slope);
      d[j] = sk_pack2(lo, hi);
    }
    v = make_uint4(d[0], d[1], d[2], d[3]);
  };
  float pc0 = 0.f, pc1 = 0.f, pc2 = 0.f, pq = 0.f;
  // the halo pixels (LDS pixel 0 and Wo + 1) of the three ring slots: padding, never written again
  if (t < 48) *reinterpret_cast<uint4*>(ring + (t >> 4) * SLOT + (((t >> 3) & 1) ? Wo + 1 : 0) * LU_PXS + (t & 7) * 8) =
      make_uint4(0, 0, 0, 0);
  // write-out map of one output row group (3 channels x 2 row parities x 256 floats = 384 float4): every thread one float4
  // of rows 0..3 and one float2 of rows 4, 5 -- no partial trip through a loop
  const int rid4 = t >> 6, m4 = t & 63;                  // float4 m4 of staged row rid4
  const int rid2 = 4 + (t >> 7), m2 = t & 127;           // float2 m2 of staged row rid2

  for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
    const int hq0 = (sidx & 7) * STRIP, n = sidx >> 3;
    const uint16_t* xn = x + (size_t)n * Ho * Wo * 64;
    Lu128Row ra, rb;
    lu128_load(ra, xn, hq0 - 1, t);
    lu128_load(rb, xn, hq0, t);
    // step k: stage row hq0 - 1 + k (held in `cur`) into ring slot k % 3, re-load `cur` with row hq0 + 1 + k (two steps ahead),
    // and for k >= 2 produce output row group hq0 + k - 2 from the three slots
    auto step = [&](int k, Lu128Row& cur, bool compute) __attribute__((always_inline)) {
      uint16_t* slot = ring + (k % 3) * SLOT;
      {
        const bool row_ok = (unsigned)(hq0 - 1 + k) < (unsigned)Ho;
        uint16_t* d = slot + (1 + (t >> 3)) * LU_PXS + (t & 7) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint4 v = cur.v[j];
          if (PRE) bn8(v);
          v.x = row_ok ? v.x : 0u; v.y = row_ok ? v.y : 0u; v.z = row_ok ? v.z : 0u; v.w = row_ok ? v.w : 0u;
          *reinterpret_cast<uint4*>(d + 32 * j * LU_PXS) = v;
        }
      }
      __syncthreads();
      lu128_load(cur, xn, hq0 + 1 + k, t);         // beyond the strip's halo: a harmless clamped re-read
      if (compute) {
        const int hq = hq0 + k - 2;
        // the tanh-backward factor's image values: issued before the MFMAs, used behind the barrier
        float4 im4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 im2 = make_float2(0.f, 0.f);
        const size_t oi4 = (((size_t)n * SK_I + (rid4 >> 1)) * H + 2 * hq + (rid4 & 1)) * W + 4 * m4;
        const size_t oi2 = (((size_t)n * SK_I + (rid2 >> 1)) * H + 2 * hq + (rid2 & 1)) * W + 2 * m2;
        if (TB) {
          im4 = *reinterpret_cast<const float4*>(post.tb_img + oi4);
          im2 = *reinterpret_cast<const float2*>(post.tb_img + oi2);
        }
        const uint16_t* s0 = ring + ((k - 2) % 3) * SLOT;
        const uint16_t* s1 = ring + ((k - 1) % 3) * SLOT;
        const uint16_t* s2 = slot;
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
          const int st = wave + 4 * sj;
          sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const int aoff = (16 * st + lp) * LU_PXS + 8 * lq;
#pragma unroll
          for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int o = aoff + dw * LU_PXS + 32 * c;
              sk_bf16x8 a0 = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(s0 + o));
              sk_bf16x8 a1 = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(s1 + o));
              sk_bf16x8 a2 = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(s2 + o));
              acc = rg_mfma_h16_16x16x32(a0, wf[0][dw][c], acc, 0, 0, 0);
              acc = rg_mfma_h16_16x16x32(a1, wf[1][dw][c], acc, 0, 0, 0);
              acc = rg_mfma_h16_16x16x32(a2, wf[2][dw][c], acc, 0, 0, 0);
            }
          if (ni < SK_I) {
            float* orow = outt + (ni * 2 + nph) * LU_OS + 2 * (16 * st + 4 * lq) + npw;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[r] + bv;
              if (apply_tanh) v = sk_fast_tanh(v);
              orow[2 * r] = v;
            }
          }
        }
        __syncthreads();
        float4 v4 = *reinterpret_cast<const float4*>(outt + rid4 * LU_OS + 4 * m4);
        float2 v2 = *reinterpret_cast<const float2*>(outt + rid2 * LU_OS + 2 * m2);
        if (TB) {
          v4 = make_float4(v4.x * (1.f - im4.x * im4.x), v4.y * (1.f - im4.y * im4.y), v4.z * (1.f - im4.z * im4.z),
                           v4.w * (1.f - im4.w * im4.w));
          v2 = make_float2(v2.x * (1.f - im2.x * im2.x), v2.y * (1.f - im2.y * im2.y));
        }
        *reinterpret_cast<float4*>(y + oi4) = v4;
        *reinterpret_cast<float2*>(y + oi2) = v2;
        if (PART) {
          const float s4 = (v4.x + v4.y) + (v4.z + v4.w), s2 = v2.x + v2.y;
          const int i4 = rid4 >> 1;                      // 0 or 1; the float2 rows are channel 2
          pc0 += i4 == 0 ? s4 : 0.f; pc1 += i4 == 1 ? s4 : 0.f; pc2 += s2;
          pq += ((v4.x * v4.x + v4.y * v4.y) + (v4.z * v4.z + v4.w * v4.w)) + (v2.x * v2.x + v2.y * v2.y);
        }
      }
    };
    step(0, ra, false);
    step(1, rb, false);
    step(2, ra, true);          // peeled: the loop is entered with the pending loads / stores of its back edge (see first_down)
    step(3, rb, true);
#pragma unroll 1
    for (int k = 4; k < STRIP + 2; k += 2) {
      step(k, ra, true);
      step(k + 1, rb, true);
    }
    // no barrier between strips: the next strip's first ring store (slot 0) comes behind the barrier that follows step 17's
    // MFMAs, the last readers of the ring
  }
  if (PART) {
    float vals[4] = {pc0, pc1, pc2, pq};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) vals[k] += __shfl_xor(vals[k], o, 64);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) outt[wave * 4 + k] = vals[k];
    }
    __syncthreads();
    if (t < 4) post.part[(size_t)blockIdx.x * 4 + t] = (outt[t] + outt[4 + t]) + (outt[8 + t] + outt[12 + t]);
  }
}

// skinny weight gradient at Wo = 128: unit = one whole row of `low` (128 pixels x 64 channels) and its 12 image rows
// (a native vector type: HIP's uint4 is a class, and a uint4 that is only copied -- global -> register -> LDS -- stays a
// memcpy through a private-memory object that hipcc then places in scratch / LDS)
typedef __attribute__((ext_vector_type(4))) unsigned sk_u32x4;
struct Sw128Low { sk_u32x4 v0, v1, v2, v3; };
__device__ __forceinline__ void sw128_load_low(Sw128Low& r, const uint16_t* __restrict__ low, int u, int t) {
  const uint16_t* src = low + (size_t)u * R128_WO * 64 + (size_t)t * 8;
  r.v0 = *reinterpret_cast<const sk_u32x4*>(src);
  r.v1 = *reinterpret_cast<const sk_u32x4*>(src + 2048);
  r.v2 = *reinterpret_cast<const sk_u32x4*>(src + 4096);
  r.v3 = *reinterpret_cast<const sk_u32x4*>(src + 6144);
}

__global__ __launch_bounds__(256, 2) void skinny_wgrad_rows128_kernel(const uint16_t* __restrict__ low,
                                                                      const float* __restrict__ x, float* __restrict__ slab,
                                                                      int nunits, float* __restrict__ bias_slab) {
  // both LDS images double-buffered (one barrier per unit), two units of loads in flight in registers (7 x 16 B per thread and
  // unit): 59 KB and ~200 registers, two workgroups per CU.
  // The MFMA tile has 64 columns for the 48 patch taps: column 48 is a row of ONES in the patch image, so that
  // D[o][48] = sum_pixels low[pixel][o] -- the layer's BIAS gradient partial (bias_slab[block][64], optional), a by-product of the
  // pass over `low` that rg_col_sum otherwise makes on its own (134-268 MB per call)
  __shared__ __attribute__((aligned(16))) uint16_t pt2[2][(SK_K + 1) * SW_PTS];    // 2 x 13 KB
  __shared__ __attribute__((aligned(16))) uint16_t lt2[2][128 * 64];         // 2 x 16 KB (lt2[0] reused for the final reduction)
  uint16_t* const lt = lt2[0];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  sk_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int grp16 = lane >> 4, idx = lane & 15, q4 = idx >> 2, p4 = idx & 3, fh = grp16 >> 1, cb = grp16 & 1;
  const int per = (nunits + gridDim.x - 1) / gridDim.x;
  const int u0 = blockIdx.x * per, u1 = min(nunits, u0 + per);
  if (u0 < u1) {
    fd128_zero_edges<SW_PTS>(pt2[0], t);
    fd128_zero_edges<SW_PTS>(pt2[1], t);
    if (t < SW_PTS) { pt2[0][SK_K * SW_PTS + t] = RG_H16_ONE; pt2[1][SK_K * SW_PTS + t] = RG_H16_ONE; }     // 1.0 in the 16-bit type
    Fd128Rows ra, rb;
    Sw128Low la, lb;
    fd128_load(ra, x, u0, wave, lane);
    sw128_load_low(la, low, u0, t);
    fd128_load(rb, x, min(u0 + 1, u1 - 1), wave, lane);
    sw128_load_low(lb, low, min(u0 + 1, u1 - 1), t);
    auto step = [&](int u, Fd128Rows& cur, Sw128Low& lcur, uint16_t* ptu, uint16_t* ltu) __attribute__((always_inline)) {
      fd128_scatter<SW_PTS>(ptu, cur, fd128_row_ok(u, wave), wave, lane);
      {
        const int px = t >> 3, c = t & 7;
        uint16_t* d = ltu + px * 64 + ((c ^ (((px >> 1) & 1) << 2)) << 3);
        *reinterpret_cast<sk_u32x4*>(d) = lcur.v0;
        *reinterpret_cast<sk_u32x4*>(d + 32 * 64) = lcur.v1;
        *reinterpret_cast<sk_u32x4*>(d + 64 * 64) = lcur.v2;
        *reinterpret_cast<sk_u32x4*>(d + 96 * 64) = lcur.v3;
      }
      __syncthreads();
      fd128_load(cur, x, min(u + 2, u1 - 1), wave, lane);
      sw128_load_low(lcur, low, min(u + 2, u1 - 1), t);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int prow = wave * 32 + ks * 16 + 8 * fh + q4;
        const int sw = ((prow >> 1) & 1) << 2;                 // same for prow + 4
        sk_bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = 4 * i + 2 * cb + (p4 >> 1);
          const uint16_t* ap = ltu + prow * 64 + ((c ^ sw) << 3) + (p4 & 1) * 4;
          sk_s16x4 lo = sk_tr_read(ap), hi = sk_tr_read(ap + 4 * 64);
          fa[i] = __builtin_bit_cast(sk_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          const int k = 32 * i + r;
          fb[i] = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4*>(ptu + (k <= SK_K ? k : 0) * SW_PTS +
                                                                                wave * 32 + ks * 16 + 8 * h));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = rg_mfma_h16_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    };
    int u = u0;
    if (u + 1 < u1) {
      step(u, ra, la, pt2[0], lt2[0]);
      step(u + 1, rb, lb, pt2[1], lt2[1]);
      u += 2;
#pragma unroll 1
      for (; u + 1 < u1; u += 2) {
        step(u, ra, la, pt2[0], lt2[0]);
        step(u + 1, rb, lb, pt2[1], lt2[1]);
      }
    }
    if (u < u1) step(u, ra, la, pt2[0], lt2[0]);
  }
  __syncthreads();
  // block reduction of the 4 waves: acc[i][j][reg] = D[o = 32i + (reg&3)+8(reg>>2)+4h][k = 32j + r]
  float* red = reinterpret_cast<float*>(lt);
  for (int i = t; i < 64 * SK_K + 64; i += 256) red[i] = 0.f;         // [64][48] + the bias column [64] behind it
  __syncthreads();
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = 32 * j + r;
          if (k < SK_K) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * SK_K + k] += acc[i][j][e];
          } else if (k == SK_K) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[64 * SK_K + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h] += acc[i][j][e];
          }
        }
    }
    __syncthreads();
  }
  float* sl = slab + (size_t)blockIdx.x * 64 * SK_K;
  for (int i = t; i < 64 * SK_K; i += 256) sl[i] = red[i];
  if (bias_slab && t < 64) bias_slab[(size_t)blockIdx.x * 64 + t] = red[64 * SK_K + t];
}

// row-staged bf16 kernels: unit width (output pixels) for an output row of Wo pixels; 3 resident blocks per CU (VGPR-limited)
constexpr int SK_ROWS_BLOCKS = 768;
// the benchmark's geometry takes the control-flow-free kernels (`skinny128` = 0 / RNAGAN_SKINNY128=0: the general ones)
bool sk_rows128(int H, int W) { return H == R128_H && W == R128_W && rg_option("skinny128", 1) != 0; }
bool sk_rows_chunk(int Wo, int* chunk) {
  if (Wo >= 128 && Wo % 128 == 0) { *chunk = 128; return true; }
  if (Wo == 32 || Wo == 64) { *chunk = Wo; return true; }
  return false;
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

// sign bits of a bf16 activation [npix][64] (fallback where the producer does not write them itself)
__global__ __launch_bounds__(256) void sign_pack64_kernel(const uint16_t* __restrict__ a, unsigned long long* __restrict__ bits,
                                                          size_t npix) {
  // 8 lanes per pixel, 16 B (8 channels) each; the 8 bytes of the word are gathered with 3 xor-shuffles
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t p = gid >> 3;
  const int c8 = (int)(gid & 7);
  unsigned long long part = 0;
  if (p < npix) {
    const uint4 v = *reinterpret_cast<const uint4*>(a + p * 64 + c8 * 8);
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) b |= ((sk_pos16(d[k]) ? 1u : 0u) | (sk_pos16(d[k] >> 16) ? 2u : 0u)) << (2 * k);
    part = (unsigned long long)b << (8 * c8);
  }
  part |= __shfl_xor(part, 1, 64);
  part |= __shfl_xor(part, 2, 64);
  part |= __shfl_xor(part, 4, 64);
  if (p < npix && c8 == 0) bits[p] = part;
}

int rg_skinny_sign_pack(const void* a, void* bits, long long npix, int C, int dtype, hipStream_t st) {
  RG_REQUIRE(C == 64 && dtype == RG_H16, RG_EUNSUPPORTED, "sign_pack: 64 bf16 channels only");
  const unsigned blocks = (unsigned)((npix * 8 + 255) / 256);
  hipLaunchKernelGGL(sign_pack64_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)a, (unsigned long long*)bits,
                     (size_t)npix);
  RG_LAUNCH_CHECK("sign_pack");
  return RG_OK;
}

// first_down with the result multiplied by the LeakyReLU derivative given as packed sign bits (see first_down_rows_kernel);
// false when this shape has no such kernel (the caller then runs first_down + lrelu_bwd)
bool rg_skinny_first_down_masked_supported(int H, int W, int I, int O, int dtype) {
  int chunk;
  return dtype == RG_H16 && I == SK_I && O == 64 && sk_rows_chunk(W / 2, &chunk);
}
int rg_skinny_first_down_masked(const float* x, const float* w, void* y, const void* mask_bits, float mslope, int N, int H,
                                int W, int I, int O, int dtype, hipStream_t st) {
  RG_REQUIRE(rg_skinny_first_down_masked_supported(H, W, I, O, dtype), RG_EUNSUPPORTED, "first_down_masked: shape");
  long long npix = (long long)N * (H / 2) * (W / 2);
  int chunk;
  sk_rows_chunk(W / 2, &chunk);
  RG_REQUIRE(npix / chunk < 0x7fffffff, RG_EUNSUPPORTED, "first_down_masked: too large");
  int nunits = (int)(npix / chunk);
  int blocks = nunits < SK_ROWS_BLOCKS ? nunits : SK_ROWS_BLOCKS;
  if (sk_rows128(H, W)) {
    hipLaunchKernelGGL((first_down_rows128_kernel<false, true>), dim3(blocks), dim3(256), 0, st, x, w, (const float*)nullptr,
                       (uint16_t*)y, (unsigned long long*)nullptr, (const unsigned long long*)mask_bits, mslope, 1.f, nunits);
    RG_LAUNCH_CHECK("first_down_masked(128)");
    return RG_OK;
  }
  hipLaunchKernelGGL((first_down_rows_kernel<false, true>), dim3(blocks), dim3(256), 0, st, x, w, (const float*)nullptr,
                     (uint16_t*)y, (unsigned long long*)nullptr, (const unsigned long long*)mask_bits, mslope, N, H, W, 1.f,
                     chunk, nunits);
  RG_LAUNCH_CHECK("first_down_masked");
  return RG_OK;
}

int rg_skinny_first_down(const float* x, const float* w, const float* bias, void* y, void* bits, int N, int H, int W, int I,
                         int O, float slope, int dtype, hipStream_t st) {
  (void)I;
  long long npix = (long long)N * (H / 2) * (W / 2);
  static int no_mfma = -1;
  if (no_mfma < 0) { const char* e = getenv("RNAGAN_SKINNY_VALU"); no_mfma = (e && e[0] == '1') ? 1 : 0; }
  int chunk;
  if (dtype == RG_H16 && O == 64 && sk_rows_chunk(W / 2, &chunk) && npix / chunk < 0x7fffffff && !no_mfma) {
    int nunits = (int)(npix / chunk);
    int blocks = nunits < SK_ROWS_BLOCKS ? nunits : SK_ROWS_BLOCKS;
    if (sk_rows128(H, W)) {
      if (bits)
        hipLaunchKernelGGL((first_down_rows128_kernel<true, false>), dim3(blocks), dim3(256), 0, st, x, w, bias, (uint16_t*)y,
                           (unsigned long long*)bits, (const unsigned long long*)nullptr, 1.f, slope, nunits);
      else
        hipLaunchKernelGGL((first_down_rows128_kernel<false, false>), dim3(blocks), dim3(256), 0, st, x, w, bias, (uint16_t*)y,
                           (unsigned long long*)nullptr, (const unsigned long long*)nullptr, 1.f, slope, nunits);
      RG_LAUNCH_CHECK("first_down(128)");
      return RG_OK;
    }
    if (bits)
      hipLaunchKernelGGL((first_down_rows_kernel<true, false>), dim3(blocks), dim3(256), 0, st, x, w, bias, (uint16_t*)y,
                         (unsigned long long*)bits, (const unsigned long long*)nullptr, 1.f, N, H, W, slope, chunk, nunits);
    else
      hipLaunchKernelGGL((first_down_rows_kernel<false, false>), dim3(blocks), dim3(256), 0, st, x, w, bias, (uint16_t*)y,
                         (unsigned long long*)nullptr, (const unsigned long long*)nullptr, 1.f, N, H, W, slope, chunk, nunits);
    RG_LAUNCH_CHECK("first_down(mfma)");
    return RG_OK;
  }
  if (bits) {
    int rc = rg_skinny_first_down(x, w, bias, y, nullptr, N, H, W, I, O, slope, dtype, st);
    return rc != RG_OK ? rc : rg_skinny_sign_pack(y, bits, npix, O, dtype, st);
  }
  unsigned blocks = (unsigned)((npix + 255) / 256);
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((first_down_kernel<T, 16>), dim3(blocks), dim3(256), 0, st, x, w, bias, (T*)y, N, H, W, O, slope);
    RG_LAUNCH_CHECK("first_down");
    return RG_OK;
  })
}

bool rg_skinny_last_up_pre_supported(int Wo, int O, int dtype) {
  int chunk;
  return dtype == RG_H16 && O == 64 && sk_rows_chunk(Wo, &chunk);
}
// number of per-workgroup partial rows rg_skinny_last_up writes with post_part (0: this shape has no rows kernel)
int rg_skinny_last_up_post_blocks(int N, int Ho, int Wo, int O, int dtype) {
  int chunk;
  if (!(dtype == RG_H16 && O == 64 && sk_rows_chunk(Wo, &chunk))) return 0;
  int strip = Ho < 16 ? Ho : 16;
  while (Ho % strip) --strip;
  long long nstrips = (long long)N * (Ho / strip) * (Wo / chunk);
  return nstrips < 512 ? (int)nstrips : 512;
}
int rg_skinny_lu_part_final(const float* part, int nb, float* out, int accumulate, int mode, float* loss, float* coef,
                            float lambd, hipStream_t st, float in_scale, float out_scale) {
  hipLaunchKernelGGL(lu_part_final_kernel, dim3(1), dim3(256), 0, st, part, nb, out, accumulate, mode, loss, coef, lambd,
                     1.f / in_scale, out_scale);
  RG_LAUNCH_CHECK("last_up_post_final");
  return RG_OK;
}
int rg_skinny_last_up(const void* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo, int O, int I,
                      int apply_tanh, int dtype, hipStream_t st, const float* pre_mean, const float* pre_invstd,
                      const float* pre_gamma, const float* pre_beta, float pre_slope, const float* post_tb_img,
                      float* post_part) {
  const LuPre pre{pre_mean, pre_invstd, pre_gamma, pre_beta, pre_slope};
  const LuPost post{post_tb_img, post_part};
  RG_REQUIRE(!(post_tb_img || post_part) || rg_skinny_last_up_post_blocks(N, Ho, Wo, O, dtype) > 0, RG_EUNSUPPORTED,
             "last_up: fused output epilogue");
  RG_REQUIRE(!pre_mean || rg_skinny_last_up_pre_supported(Wo, O, dtype), RG_EUNSUPPORTED, "last_up: fused BatchNorm input");
  (void)I;
  static int lu_valu = -1;
  if (lu_valu < 0) { const char* e = getenv("RNAGAN_SKINNY_VALU"); lu_valu = (e && e[0] == '1') ? 1 : 0; }
  int chunk;
  if (dtype == RG_H16 && O == 64 && sk_rows_chunk(Wo, &chunk) && !lu_valu) {
    int strip = Ho < 16 ? Ho : 16;
    while (Ho % strip) --strip;
    long long nstrips = (long long)N * (Ho / strip) * (Wo / chunk);
    int blocks = nstrips < 512 ? (int)nstrips : 512;
    if (sk_rows128(2 * Ho, 2 * Wo)) {
#define LU128(PRE_, TB_, PART_)                                                                                          \
  hipLaunchKernelGGL((last_up_rows128_kernel<PRE_, TB_, PART_>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)x, w, bias, \
                     y, apply_tanh, (int)nstrips, pre, post)
      if (pre_mean) {
        RG_REQUIRE(!post_tb_img && !post_part, RG_EUNSUPPORTED, "last_up: fused input and output passes together");
        LU128(true, false, false);
      } else if (post_tb_img) {
        if (post_part) LU128(false, true, true); else LU128(false, true, false);
      } else {
        if (post_part) LU128(false, false, true); else LU128(false, false, false);
      }
#undef LU128
      RG_LAUNCH_CHECK("last_up(128)");
      return RG_OK;
    }
    hipLaunchKernelGGL(last_up_rows_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)x, w, bias, y, N, Ho, Wo,
                       apply_tanh, chunk, strip, (int)nstrips, pre, post);
    RG_LAUNCH_CHECK("last_up(mfma)");
    return RG_OK;
  }
  if (dtype == RG_H16) {
    constexpr int TH = 8;
    int tiles = ((Wo + LU_TW - 1) / LU_TW) * ((Ho + TH - 1) / TH);
    size_t sh = (size_t)(TH + 2) * (LU_TW + 2) * (O * 2 + 8);
    hipLaunchKernelGGL((last_up_kernel<h16_t, TH>), dim3((unsigned)((long long)N * tiles)), dim3(TH * 32), sh, st,
                       (const h16_t*)x, w, bias, y, N, Ho, Wo, O, apply_tanh);
  } else if (dtype == RG_F32) {
    if (O % 16 == 0 && !getenv("RNAGAN_LASTUP_WHOLE")) {    // 16 channels per staging pass: 15 KB per 128 threads
      constexpr int TH = 4, CCH = 16;
      int tiles = ((Wo + LU_TW - 1) / LU_TW) * ((Ho + TH - 1) / TH);
      size_t sh = (size_t)(TH + 2) * (LU_TW + 2) * (CCH * 4 + 8);
      hipLaunchKernelGGL((last_up_kernel<float, TH, CCH>), dim3((unsigned)((long long)N * tiles)), dim3(TH * 32), sh, st,
                         (const float*)x, w, bias, y, N, Ho, Wo, O, apply_tanh);
    } else {
      constexpr int TH = 2;
      int tiles = ((Wo + LU_TW - 1) / LU_TW) * ((Ho + TH - 1) / TH);
      size_t sh = (size_t)(TH + 2) * (LU_TW + 2) * (O * 4 + 8);
      hipLaunchKernelGGL((last_up_kernel<float, TH>), dim3((unsigned)((long long)N * tiles)), dim3(TH * 32), sh, st,
                         (const float*)x, w, bias, y, N, Ho, Wo, O, apply_tanh);
    }
  } else {
    rg_set_error("bad dtype %d", dtype);
    return RG_EINVAL;
  }
  RG_LAUNCH_CHECK("last_up");
  return RG_OK;
}

constexpr int SKW_BLOCKS = SK_ROWS_BLOCKS;

size_t rg_skinny_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I) {
  (void)I;
  int ppb;
  int nb = skinny_wgrad_blocks((long long)N * Ho * Wo, &ppb);
  if (nb < SKW_BLOCKS) nb = SKW_BLOCKS;
  return (size_t)nb * (O * SK_K + 64) * sizeof(float);          // + the bias-gradient partials of the row kernel
}

// The row-kernel form with the per-workgroup partial gradients LEFT in `slab` ([*nslab_out][O * 48] fp32, the layout of dw): the
// caller's optimizer step sums them (rg_adam_step_slabs).  *nslab_out = 0: this shape / dtype has no such form, nothing was
// launched.
int rg_skinny_wgrad_slabs_impl(const void* low, const float* high_nchw, int N, int Ho, int Wo, int O, int I, int dtype,
                               void* slab, size_t slab_bytes, int* nslab_out, float* bias_slab, int* bias_done_out,
                               hipStream_t st) {
  (void)I;
  *nslab_out = 0;
  if (bias_done_out) *bias_done_out = 0;
  const size_t elems = (size_t)O * SK_K;
  const long long npix = (long long)N * Ho * Wo;
  int chunk;
  if (!(dtype == RG_H16 && O == 64 && sk_rows_chunk(Wo, &chunk) && npix / chunk < 0x7fffffff)) return RG_OK;
  const int nunits = (int)(npix / chunk);
  int nbm = nunits < SK_ROWS_BLOCKS ? nunits : SK_ROWS_BLOCKS;
  if (sk_rows128(2 * Ho, 2 * Wo)) {
    if (nbm > 512) nbm = 512;
    RG_REQUIRE(slab && slab_bytes >= (size_t)nbm * elems * sizeof(float), RG_EWORKSPACE, "skinny_wgrad_slabs: buffer too small");
    hipLaunchKernelGGL(skinny_wgrad_rows128_kernel, dim3(nbm), dim3(256), 0, st, (const uint16_t*)low, high_nchw, (float*)slab,
                       nunits, bias_slab);
    if (bias_done_out) *bias_done_out = bias_slab ? 1 : 0;
  } else {
    RG_REQUIRE(slab && slab_bytes >= (size_t)nbm * elems * sizeof(float), RG_EWORKSPACE, "skinny_wgrad_slabs: buffer too small");
    hipLaunchKernelGGL(skinny_wgrad_rows_kernel, dim3(nbm), dim3(256), 0, st, (const uint16_t*)low, high_nchw, (float*)slab, N,
                       2 * Ho, 2 * Wo, chunk, nunits);
  }
  RG_LAUNCH_CHECK("skinny_wgrad_slabs");
  *nslab_out = nbm;
  return RG_OK;
}

int rg_skinny_wgrad_impl(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                         int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, float* dbias, int bias_accumulate,
                         int* bias_done_out) {
  (void)I;
  int bias_done_local = 0;
  if (!bias_done_out) bias_done_out = &bias_done_local;
  *bias_done_out = 0;
  size_t elems = (size_t)O * SK_K;
  long long npix = (long long)N * Ho * Wo;
  static int no_mfma = -1;
  if (no_mfma < 0) { const char* e = getenv("RNAGAN_SKINNY_VALU"); no_mfma = (e && e[0] == '1') ? 1 : 0; }
  int chunk;
  if (dtype == RG_H16 && O == 64 && sk_rows_chunk(Wo, &chunk) && npix / chunk < 0x7fffffff && !no_mfma) {
    int nunits = (int)(npix / chunk);
    int nbm = nunits < SK_ROWS_BLOCKS ? nunits : SK_ROWS_BLOCKS;
    RG_REQUIRE(ws && ws_bytes >= (size_t)nbm * elems * sizeof(float), RG_EWORKSPACE, "skinny_wgrad: workspace too small");
    if (sk_rows128(2 * Ho, 2 * Wo)) {
      if (nbm > 512) nbm = 512;                     // two workgroups per CU (59 KB of LDS each)
      // dbias (optional): the kernel's bias column, partials behind the weight partials in the workspace
      float* bpart = (dbias && ws_bytes >= (size_t)nbm * (elems + 64) * sizeof(float)) ? (float*)ws + (size_t)nbm * elems : nullptr;
      hipLaunchKernelGGL(skinny_wgrad_rows128_kernel, dim3(nbm), dim3(256), 0, st, (const uint16_t*)low, high_nchw, (float*)ws,
                         nunits, bpart);
      RG_LAUNCH_CHECK("skinny_wgrad(128)");
      int rc = rg_reduce_slabs((const float*)ws, dw, elems, nbm, accumulate, 0, 0, st);
      if (rc || !bpart) return rc;
      *bias_done_out = 1;
      return rg_reduce_slabs(bpart, dbias, 64, nbm, bias_accumulate, 0, 0, st);
    }
    hipLaunchKernelGGL(skinny_wgrad_rows_kernel, dim3(nbm), dim3(256), 0, st, (const uint16_t*)low, high_nchw, (float*)ws,
                       N, 2 * Ho, 2 * Wo, chunk, nunits);
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
