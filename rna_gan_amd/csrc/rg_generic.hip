// rg_generic.hip -- generic (any shape, fp32 math on the vector ALUs) tiled kernels.
//
// One LDS-tiled GEMM skeleton, C[M][N] = sum_k A(m,k) * B(k,n), whose operands are fetched through
// functors (implicit im2col with zero padding, NHWC / NCHW, fp32 / bf16 storage).  It implements
// every conv / dense entry point of include/rnagan_hip.h for shapes the MFMA kernels do not take
// and IS the fp32 parity path (RG_F32 activations): same operation order on every launch,
// deterministic split-K (slabs summed in fixed order).
#include "rg_common.h"
#include "rg_internal.h"
#include <type_traits>
#include <algorithm>

namespace {

constexpr int GB_M = 64, GB_N = 64, GB_K = 16;

// A_KFAST: consecutive threads fetch consecutive k (else consecutive m); same idea for B.
template <bool A_KFAST, bool B_KFAST, class FA, class FB, class SC>
__global__ __launch_bounds__(256) void gemm_generic_kernel(FA fa, FB fb, SC sc, int M, int N, int K, int nsplit,
                                                           int klen) {
  __shared__ float As[GB_K][GB_M + 4];
  __shared__ float Bs[GB_K][GB_N + 4];
  const int bm = blockIdx.x * GB_M, bn = blockIdx.y * GB_N;
  const int zs = blockIdx.z % nsplit, zb = blockIdx.z / nsplit;
  const int k_begin = zs * klen;
  const int k_end = min(K, k_begin + klen);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = k_begin; k0 < k_end; k0 += GB_K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = threadIdx.x + i * 256;
      int kk, mm;
      if (A_KFAST) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      float v = 0.f;
      if (bm + mm < M && k0 + kk < k_end) v = fa(zb, bm + mm, k0 + kk);
      As[kk][mm] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = threadIdx.x + i * 256;
      int kk, nn;
      if (B_KFAST) { kk = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kk = idx >> 6; }
      float v = 0.f;
      if (bn + nn < N && k0 + kk < k_end) v = fb(zb, k0 + kk, bn + nn);
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GB_K; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int m = bm + ty * 4 + i, n = bn + tx * 4 + j;
      if (m < M && n < N) sc(zb, zs, m, n, acc[i][j]);
    }
}

// ----------------------------------------------------------------------------------------------
// The same GEMM on the f32 MATRIX cores: v_mfma_f32_32x32x2_f32 (f32 operands, f32 accumulate -- exact f32 arithmetic, a
// fused-multiply-add chain per accumulator like the loop above; 64 FLOP/clk/SIMD = the vector FMA rate, but the operand
// reads and the issue slots of the vector pipe are left to the functors' address arithmetic).  This is the fp32 mode's
// conv / dense kernel wherever the output tile is worth it (the reference's own arithmetic is fp32: src/betaVAE.py:184,
// 223, 230-236; SURVEY 8d "~157 TFLOP/s fp32 matrix").
//   block = 256 threads = 2 x 2 waves, tile 128 x 128, each wave 64 x 64 = 2 x 2 accumulator tiles of 32 x 32 (or 64 x 64 with
//   one accumulator tile per wave where the output has too few large tiles to fill the chip);
//   k-tile 16: operands fetched through the SAME functors into registers one k-tile ahead (the global loads fly under
//   the MFMAs of the current tile), staged in LDS k-major ([k][m]: a fragment read is 32 consecutive floats per k row,
//   conflict-free ds_read_b32); MFMA step s consumes k = 2s (lanes 0-31) and 2s + 1 (lanes 32-63).
// Same split-K contract and the same store functors as gemm_generic_kernel.
// ----------------------------------------------------------------------------------------------
#ifndef RG_MBK
#define RG_MBK 16
#endif
constexpr int MB_M = 128, MB_N = 128, MB_K = RG_MBK;      // k-tile depth (build-time knob: 16 or 32)
typedef float mb_f32x16 __attribute__((ext_vector_type(16)));

// Two-phase operand access.  A thread of gemm_mfma32_kernel fetches the same 1 or 8 tile rows (m for A, n for B) in every
// k-tile and 8 or 1 k positions per k-tile, so whatever an operand functor derives from the row index alone (pixel
// coordinates, base offsets) is computed once per thread, and what it derives from k alone once per k position -- not once per
// fetched element as operator() does (~25 integer operations per element for the gathered operands, which made the vector
// pipe, not the matrix pipe, the bound of the first version of this kernel: 55 of 157 TFLOP/s).
//   Op<F, A_SIDE>::row(f, zb, r) / ::col(f, zb, k) / ::at(f, zb, row, col, ok)
// The primary template wraps operator() (no saving); the gathered operands of the conv layers specialise it below.
// P2 (compile-time): every divisor the functor uses is a power of two (pow2(f), decided once per launch: the kernel branches
// once per k-tile between the two instantiations of its fetch code instead of once per index computation).
// (Launches whose both operands are structured -- power-of-two conv geometry, dense layers -- take gemm_mfma32s_kernel below.)
template <class F, bool A_SIDE> struct Op {
  struct R { int r; };
  struct C { int k; };
  static __device__ __forceinline__ bool pow2(const F&) { return true; }
  static bool fits32(const F&) { return true; }          // host: element offsets of the two-phase form fit an int
  template <bool P2> static __device__ __forceinline__ R row(const F&, int, int r) { return {r}; }
  template <bool P2> static __device__ __forceinline__ C col(const F&, int, int k) { return {k}; }
  // ok: false where the element is implicit padding (the caller zeroes the value WHEN IT STORES IT: the load's first use then
  // sits behind the MFMAs of the current k-tile)
  static __device__ __forceinline__ float at(const F& f, int zb, const R& r, const C& c, bool& ok) {
    ok = true;
    return A_SIDE ? f(zb, r.r, c.k) : f(zb, c.k, r.r);
  }
};

// TM x TN: the wave tile (64 x 64 = 2 x 2 accumulator tiles, or 32 x 32 = one); block tile = 2 TM x 2 TN.  The small form
// has 4 x the workgroups: for the deep layers whose output is only 1-2 hundred 128 x 128 tiles (three workgroups fit a CU).
template <int TM, int TN, bool A_KFAST, bool B_KFAST, class FA, class FB, class SC>
__global__ __launch_bounds__(256, 3) void gemm_mfma32_kernel(FA fa, FB fb, SC sc, int M, int N, int K, int nsplit, int klen) {
  constexpr int BM = 2 * TM, BN = 2 * TN, IA = TM / 32, IB = TN / 32;
  constexpr int NSA = BM * MB_K / 256, NSB = BN * MB_K / 256;          // fetch slots per thread and k-tile (8 or 4)
  constexpr int KSA = 256 / BM, KSB = 256 / BN;                        // !KFAST: k rows covered by one pass of the block (2 or 4)
  __shared__ float As[2][MB_K][BM + 4];
  __shared__ float Bs[2][MB_K][BN + 4];
  using OA = Op<FA, true>;
  using OB = Op<FB, false>;
  const int bm = blockIdx.x * BM, bn = blockIdx.y * BN;
  const int zs = blockIdx.z % nsplit, zb = blockIdx.z / nsplit;
  const int k_begin = zs * klen;
  const int k_end = min(K, k_begin + klen);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  mb_f32x16 acc[IA][IB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fetch slots of this thread: slot i = element (tid + 256 i) of the BM x MB_K tile.  KFAST: k = tid % MB_K (one k, rows
  // tid / MB_K + RP i, RP = 256 / MB_K); otherwise row = tid % BM (one row, k = tid / BM + KSA i).
  constexpr int RP = 256 / MB_K;
  constexpr int NRA = A_KFAST ? NSA : 1, NRB = B_KFAST ? NSB : 1;
  typename OA::R rowa[NRA];
  typename OB::R rowb[NRB];
  bool oka[NRA], okb[NRB];
  const bool p2 = (int)OA::pow2(fa) & (int)OB::pow2(fb);           // uniform: one branch per k-tile selects the shift / division code
  auto init_rows = [&](auto P2) __attribute__((always_inline)) {
    constexpr bool p = decltype(P2)::value;
#pragma unroll
    for (int i = 0; i < NRA; ++i) {
      const int mm = A_KFAST ? tid / MB_K + RP * i : (tid & (BM - 1));
      oka[i] = bm + mm < M;
      rowa[i] = OA::template row<p>(fa, zb, oka[i] ? bm + mm : 0);
    }
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
      const int nn = B_KFAST ? tid / MB_K + RP * i : (tid & (BN - 1));
      okb[i] = bn + nn < N;
      rowb[i] = OB::template row<p>(fb, zb, okb[i] ? bn + nn : 0);
    }
  };
  if (p2) init_rows(std::true_type{}); else init_rows(std::false_type{});
  float ra[NSA], rb[NSB];
  unsigned pa = 0, pb = 0;            // validity bits of the fetched slots: applied when the values are stashed, i.e. AFTER the
                                      // MFMAs of the current k-tile, so that nothing waits for the loads before them
  // (always_inline: an out-of-line lambda body would reach everything it captures -- the functors, the register arrays --
  // through memory, i.e. scratch loads in the k-loop)
  auto fetch_t = [&](int k0, auto P2) __attribute__((always_inline)) {
    constexpr bool p = decltype(P2)::value;
    pa = 0; pb = 0;
    if (A_KFAST) {
      const int k = k0 + (tid & (MB_K - 1));
      const bool kv = k < k_end;
      const typename OA::C c = OA::template col<p>(fa, zb, kv ? k : k_begin);
#pragma unroll
      for (int i = 0; i < NSA; ++i) {               // (row and k are clamped: the access itself is always in bounds)
        bool ok;
        ra[i] = OA::at(fa, zb, rowa[i], c, ok);
        pa |= ((unsigned)kv & (unsigned)oka[i] & (unsigned)ok) << i;    // (bitwise &: a short-circuit && is control flow, and the load sinks into it)
      }
    } else {
#pragma unroll
      for (int i = 0; i < NSA; ++i) {
        const int k = k0 + tid / BM + KSA * i;
        const bool kv = k < k_end;
        bool ok;
        ra[i] = OA::at(fa, zb, rowa[0], OA::template col<p>(fa, zb, kv ? k : k_begin), ok);
        pa |= ((unsigned)kv & (unsigned)oka[0] & (unsigned)ok) << i;
      }
    }
    if (B_KFAST) {
      const int k = k0 + (tid & (MB_K - 1));
      const bool kv = k < k_end;
      const typename OB::C c = OB::template col<p>(fb, zb, kv ? k : k_begin);
#pragma unroll
      for (int i = 0; i < NSB; ++i) {
        bool ok;
        rb[i] = OB::at(fb, zb, rowb[i], c, ok);
        pb |= ((unsigned)kv & (unsigned)okb[i] & (unsigned)ok) << i;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NSB; ++i) {
        const int k = k0 + tid / BN + KSB * i;
        const bool kv = k < k_end;
        bool ok;
        rb[i] = OB::at(fb, zb, rowb[0], OB::template col<p>(fb, zb, kv ? k : k_begin), ok);
        pb |= ((unsigned)kv & (unsigned)okb[0] & (unsigned)ok) << i;
      }
    }
  };
  auto fetch = [&](int k0) __attribute__((always_inline)) { if (p2) fetch_t(k0, std::true_type{}); else fetch_t(k0, std::false_type{}); };
  auto stash = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NSA; ++i) {
      const float v = (pa >> i) & 1u ? ra[i] : 0.f;
      if (A_KFAST) As[st][tid & (MB_K - 1)][tid / MB_K + RP * i] = v; else As[st][tid / BM + KSA * i][tid & (BM - 1)] = v;
    }
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
      const float v = (pb >> i) & 1u ? rb[i] : 0.f;
      if (B_KFAST) Bs[st][tid & (MB_K - 1)][tid / MB_K + RP * i] = v; else Bs[st][tid / BN + KSB * i][tid & (BN - 1)] = v;
    }
  };
  if (k_begin < k_end) { fetch(k_begin); stash(0); }
  __syncthreads();
  int cur = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += MB_K) {
    const bool more = k0 + MB_K < k_end;
    if (more) fetch(k0 + MB_K);                    // global loads of the next k-tile fly under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    // the fragments of step st + 1 are read while the MFMAs of step st run (two register sets)
    float a[2][IA], b[2][IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) a[0][i] = As[cur][lh][wm * TM + 32 * i + lr];
#pragma unroll
    for (int j = 0; j < IB; ++j) b[0][j] = Bs[cur][lh][wn * TN + 32 * j + lr];
#pragma unroll
    for (int st = 0; st < MB_K / 2; ++st) {
      if (st + 1 < MB_K / 2) {
#pragma unroll
        for (int i = 0; i < IA; ++i) a[(st + 1) & 1][i] = As[cur][2 * st + 2 + lh][wm * TM + 32 * i + lr];
#pragma unroll
        for (int j = 0; j < IB; ++j) b[(st + 1) & 1][j] = Bs[cur][2 * st + 2 + lh][wn * TN + 32 * j + lr];
      }
#pragma unroll
      for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < IB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][i], b[st & 1][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);             // the loads stay above the MFMAs, their first use (the stash) below
    if (more) stash(cur ^ 1);                      // the other stage: its last readers passed the barrier of the previous tile
    __syncthreads();
    cur ^= 1;
  }
  // accumulator register r of a 32 x 32 tile: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = bn + wn * TN + j * 32 + lr;
        if (m < M && n < N) sc(zb, zs, m, n, acc[i][j][r]);
      }
}

// f32mma (option; 0 off, 1 the f32 instruction everywhere, 2 = default: see gemm_bf16x3s_kernel): fp32-storage launches with at least
// 64 rows and more than 32 columns run on the matrix cores;
// everything else (the 3-column image layers) -- and every bf16-storage launch -- keeps gemm_generic_kernel.  (Its bf16 results
// are deterministic but NOT pinned bit for bit across rounds: round 4 made the transposed conv's K order tap-major, permuted the
// G.0 columns and re-associated the split-K slab sums; the bf16 tests compare with tolerances.)
static bool use_mfma32(bool f32, int M, int N) { return f32 && M >= 64 && N >= 33 && rg_option("f32mma", RG_F32MMA_DEFAULT) != 0; }

// (defined with gemm_mfma32s_kernel below: launches it and returns true when both operands have a structured form)
template <bool AK, bool BK, class FA, class FB, class SC>
static bool launch_structured(const char* name, FA fa, FB fb, SC sc, int M, int N, int K, int nbatch, int klen, dim3 grid,
                              bool small, hipStream_t st);

template <bool AK, bool BK, class FA, class FB, class SC>
int launch_generic(const char* name, FA fa, FB fb, SC sc, int M, int N, int K, int nbatch, int nsplit,
                   hipStream_t st, bool f32 = false) {
  if (M <= 0 || N <= 0 || K <= 0) return RG_OK;
  int klen = (K + nsplit - 1) / nsplit;
  if (use_mfma32(f32, M, N) && Op<FA, true>::fits32(fa) && Op<FB, false>::fits32(fb)) {
    klen = (klen + MB_K - 1) / MB_K * MB_K;
    dim3 grid((M + MB_M - 1) / MB_M, (N + MB_N - 1) / MB_N, nbatch * nsplit);
    RG_REQUIRE(grid.y <= 65535 && grid.z <= 65535, RG_EINVAL, "%s: grid too large", name);
    if ((long long)grid.x * grid.y * grid.z < 512 || N <= 64 || M <= 64) {   // fewer than two 128 x 128 tiles per CU, or an
      // output no wider / taller than 64 (the 64-channel layers, batch-64 dense layers): 64 x 64 tiles, 4 x the blocks
      grid = dim3((M + 63) / 64, (N + 63) / 64, nbatch * nsplit);
      RG_REQUIRE(grid.y <= 65535, RG_EINVAL, "%s: grid too large", name);
      if (!launch_structured<AK, BK>(name, fa, fb, sc, M, N, K, nbatch, klen, grid, true, st))
        hipLaunchKernelGGL((gemm_mfma32_kernel<32, 32, AK, BK, FA, FB, SC>), grid, dim3(256), 0, st, fa, fb, sc, M, N, K, nsplit, klen);
    } else {
      // (measured and rejected: a 128 x 256 block tile -- wave tile 64 x 128, 8 accumulator tiles, one or two workgroups per CU by
      // registers -- 97.9 against 95.6 ms per iteration, two interleaved rounds)
      if (!launch_structured<AK, BK>(name, fa, fb, sc, M, N, K, nbatch, klen, grid, false, st))
        hipLaunchKernelGGL((gemm_mfma32_kernel<64, 64, AK, BK, FA, FB, SC>), grid, dim3(256), 0, st, fa, fb, sc, M, N, K, nsplit, klen);
    }
    RG_LAUNCH_CHECK(name);
    return RG_OK;
  }
  klen = (klen + GB_K - 1) / GB_K * GB_K;
  dim3 grid((M + GB_M - 1) / GB_M, (N + GB_N - 1) / GB_N, nbatch * nsplit);
  RG_REQUIRE(grid.y <= 65535 && grid.z <= 65535, RG_EINVAL, "%s: grid too large", name);
  hipLaunchKernelGGL((gemm_generic_kernel<AK, BK, FA, FB, SC>), grid, dim3(256), 0, st, fa, fb, sc, M, N, K, nsplit,
                     klen);
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}

// ----------------------------------------------------------------------------------------------
// split-K slab reduction: dst[idx] = (accumulate ? dst[idx] : 0) + sum_s slab[s][idx], fixed summation order.
// (Slabs have the destination's layout: the conv weight gradients are tap-major like their masters.)
// ----------------------------------------------------------------------------------------------
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dst, size_t n, int nsplit,
                                    int accumulate) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float s = accumulate ? dst[idx] : 0.f;
  for (int z = 0; z < nsplit; ++z) s += slab[(size_t)z * n + idx];
  dst[idx] = s;
}

// identity layout, 16 bytes per thread (the tap-major conv weight gradients: slab layout == dW layout)
__global__ void reduce_slabs_vec4_kernel(const float* __restrict__ slab, float* __restrict__ dst, size_t n4, size_t n,
                                         int nsplit, int accumulate) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n4) return;
  float4 s = accumulate ? reinterpret_cast<const float4*>(dst)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < nsplit; ++z) {
    const float4 v = reinterpret_cast<const float4*>(slab + (size_t)z * n)[idx];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(dst)[idx] = s;
}

// identity layout, MANY slabs of a small tensor (image-side weight gradients: 1024 x 12 KB): 16 elements x
// 16 slab lanes per block, fixed summation order (lane-strided partials, then lane 0..15).
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                                size_t n, int nsplit, int accumulate) {
  __shared__ float sm[16][17];
  const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
  const size_t idx = (size_t)blockIdx.x * 16 + e;
  float s = 0.f;
  if (idx < n)
    for (int z = l; z < nsplit; z += 16) s += slab[(size_t)z * n + idx];
  sm[l][e] = s;
  __syncthreads();
  if (l == 0 && idx < n) {
    float t = accumulate ? dst[idx] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][e];
    dst[idx] = t;
  }
}

// identity layout, many slabs (>= 16) of a mid-sized tensor (the weight gradients of the shallow conv layers: 32-128 slabs
// of 0.5-2 MB): 16 bytes per thread, 4 slab lanes per column group -- 64 threads read 1 KB contiguous per slab, 8 loads in
// flight per thread; fixed summation order (lane-strided partials, then lanes 0..3).  The 16-element form above moved 64-byte
// pieces (3.8 TB/s on 67 MB of slabs); this one streams at the rate of the vec4 kernel.
__global__ __launch_bounds__(256) void reduce_slabs_mid_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                               size_t n4, size_t n, int nsplit, int accumulate) {
  __shared__ float4 sm[4][64];
  const int c = threadIdx.x & 63, l = threadIdx.x >> 6;
  const size_t idx = (size_t)blockIdx.x * 64 + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < n4) {
    int z = l;
    for (; z + 28 < nsplit; z += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(slab + (size_t)(z + 4 * u) * n)[idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < nsplit; z += 4) {
      const float4 v = reinterpret_cast<const float4*>(slab + (size_t)z * n)[idx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sm[l][c] = s;
  __syncthreads();
  if (l == 0 && idx < n4) {
    float4 t = accumulate ? reinterpret_cast<const float4*>(dst)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) { t.x += sm[k][c].x; t.y += sm[k][c].y; t.z += sm[k][c].z; t.w += sm[k][c].w; }
    reinterpret_cast<float4*>(dst)[idx] = t;
  }
}

}  // namespace

int rg_reduce_slabs(const float* slab, float* dst, size_t n, int nsplit, int accumulate, int perm_mode, int Q,
                    hipStream_t st) {
  (void)Q;
  if (n == 0) return RG_OK;
  RG_REQUIRE(perm_mode == 0, RG_EINVAL, "reduce_slabs: slabs must have the destination's layout");
  if (nsplit >= 16 && n % 4 == 0 && n >= 65536 && n <= (1u << 22) && (((uintptr_t)slab | (uintptr_t)dst) & 15) == 0) {
    hipLaunchKernelGGL(reduce_slabs_mid_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, st, slab, dst, n / 4, n,
                       nsplit, accumulate);
  } else if (nsplit >= 64 && n <= (1u << 20)) {
    hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, slab, dst, n, nsplit,
                       accumulate);
  } else if (n % 4 == 0 && n >= 4096) {
    hipLaunchKernelGGL(reduce_slabs_vec4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, slab, dst, n / 4,
                       n, nsplit, accumulate);
  } else {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, dst, n, nsplit,
                       accumulate);
  }
  RG_LAUNCH_CHECK("reduce_slabs");
  return RG_OK;
}

namespace {
// ----------------------------------------------------------------------------------------------
// functors
// ----------------------------------------------------------------------------------------------
struct Geo {
  int N, Hl, Wl, Hh, Wh, O, I;  // low-res dims (Hl,Wl), high-res dims (Hh=2Hl, Wh=2Wl)
  int sWl, sHl, sI, sO;         // log2 + 1 of Wl / Hl / I / O when they are powers of two (0: general division), see geo_pow2
};
static int lg1(int v) { return v > 0 && (v & (v - 1)) == 0 ? rg_ilog2(v) + 1 : 0; }
static Geo geo_pow2(Geo g) { g.sWl = lg1(g.Wl); g.sHl = lg1(g.Hl); g.sI = lg1(g.I); g.sO = lg1(g.O); return g; }
// q = x / d, r = x % d for x >= 0; s = log2(d) + 1 when d is a power of two (shift / mask instead of a ~40-instruction
// integer division: the functors below run once per fetched operand element)
template <bool P2> __device__ __forceinline__ void rg_divmod_t(int x, int d, int s, int& q, int& r) {
  if (P2) { q = x >> (s - 1); r = x & (d - 1); } else { q = x / d; r = x - q * d; }
}
__device__ __forceinline__ void rg_divmod(int x, int d, int s, int& q, int& r) {
  if (s) { q = x >> (s - 1); r = x & (d - 1); } else { q = x / d; r = x - q * d; }
}

// ---- conv_down: M = N*Hl*Wl, K = 16*I (k = tap*I + ci: tap-major masters w[O][16][I]), Ncols = O
template <typename T> struct DownA {
  const T* x; Geo g;
  __device__ float operator()(int, int m, int k) const {
    int wo, t, ho, n, tap, ci;
    rg_divmod(m, g.Wl, g.sWl, t, wo); rg_divmod(t, g.Hl, g.sHl, n, ho); rg_divmod(k, g.I, g.sI, tap, ci);
    int hi = 2 * ho - 1 + (tap >> 2), wi = 2 * wo - 1 + (tap & 3);
    if (hi < 0 || hi >= g.Hh || wi < 0 || wi >= g.Wh) return 0.f;
    return Elem<T>::ld(x + (((size_t)n * g.Hh + hi) * g.Wh + wi) * g.I + ci);
  }
};
template <typename T> struct DownB {
  const float* w; Geo g;
  __device__ float operator()(int, int k, int o) const { return Elem<T>::round(w[(size_t)o * g.I * 16 + k]); }
};
template <typename T> struct RowMajorC {
  T* y; int ld;
  __device__ void operator()(int, int, int m, int n, float v) const { Elem<T>::st(y + (size_t)m * ld + n, v); }
};

// ---- conv_up: batch z = parity class (ph,pw); M = N*Hl*Wl, K = O*4 (k = o*4 + t4), Ncols = I
__device__ __forceinline__ void up_tap(int par, int a, int q, int& kidx, int& src) {
  // output index 2q+par along one axis, tap a in {0,1}: kernel index and low-res source index
  if (par == 0) { kidx = a == 0 ? 1 : 3; src = a == 0 ? q : q - 1; }
  else          { kidx = a == 0 ? 0 : 2; src = a == 0 ? q + 1 : q; }
}
// k index of the transposed conv's GEMM (K = 4 taps x O channels): tap-major, k = t4 * O + o, so that consecutive k are
// consecutive channels of one source pixel (64 contiguous bytes per 16 k in fp32; the channel-major order k = o * 4 + t4 of
// the first version made every group of 16 k touch four pixels).  RG_UP_KORDER_OLD restores the old order (A/B builds).
__device__ __forceinline__ void up_k(int k, const Geo& g, int& o, int& t4) {
#ifdef RG_UP_KORDER_OLD
  o = k >> 2; t4 = k & 3;
#else
  rg_divmod(k, g.O, g.sO, t4, o);
#endif
}
template <typename T> struct UpA {
  const T* x; Geo g;
  __device__ float operator()(int zb, int m, int k) const {
    int wq, t, hq, n;
    rg_divmod(m, g.Wl, g.sWl, t, wq); rg_divmod(t, g.Hl, g.sHl, n, hq);
    int o, t4, kh, kw, ho, wo;
    up_k(k, g, o, t4);
    up_tap(zb >> 1, t4 >> 1, hq, kh, ho);
    up_tap(zb & 1, t4 & 1, wq, kw, wo);
    if (ho < 0 || ho >= g.Hl || wo < 0 || wo >= g.Wl) return 0.f;
    return Elem<T>::ld(x + (((size_t)n * g.Hl + ho) * g.Wl + wo) * g.O + o);
  }
};
template <typename T> struct UpB {
  const float* w; Geo g;
  __device__ float operator()(int zb, int k, int i) const {
    int o, t4, kh, kw, d;
    up_k(k, g, o, t4);
    up_tap(zb >> 1, t4 >> 1, 0, kh, d);
    up_tap(zb & 1, t4 & 1, 0, kw, d);
    return Elem<T>::round(w[((size_t)o * 16 + kh * 4 + kw) * g.I + i]);      // tap-major master
  }
};
// image-side last_up keeps the PyTorch layout w[O][I][4][4]
struct UpBOihw {
  const float* w; Geo g;
  __device__ float operator()(int zb, int k, int i) const {
    int o, t4, kh, kw, d;
    up_k(k, g, o, t4);
    up_tap(zb >> 1, t4 >> 1, 0, kh, d);
    up_tap(zb & 1, t4 & 1, 0, kw, d);
    return w[((size_t)o * g.I + i) * 16 + kh * 4 + kw];
  }
};
template <typename T> struct UpC {
  T* y; Geo g; const T* mask; float mslope;       // optional fused LeakyReLU backward: v *= lrelu'(mask)
  __device__ void operator()(int zb, int, int m, int i, float v) const {
    int wq, t, hq, n;
    rg_divmod(m, g.Wl, g.sWl, t, wq); rg_divmod(t, g.Hl, g.sHl, n, hq);
    int hi = 2 * hq + (zb >> 1), wi = 2 * wq + (zb & 1);
    size_t idx = (((size_t)n * g.Hh + hi) * g.Wh + wi) * g.I + i;
    if (mask) v *= lrelu_mask(Elem<T>::ld(mask + idx), mslope);
    Elem<T>::st(y + idx, v);
  }
};
// image-side variant: NCHW fp32 output with bias + optional tanh
struct UpCNchw {
  float* y; const float* bias; int tanh_; Geo g;
  __device__ void operator()(int zb, int, int m, int i, float v) const {
    int wq, t, hq, n;
    rg_divmod(m, g.Wl, g.sWl, t, wq); rg_divmod(t, g.Hl, g.sHl, n, hq);
    int hi = 2 * hq + (zb >> 1), wi = 2 * wq + (zb & 1);
    if (bias) v += bias[i];
    if (tanh_) v = tanhf(v);
    y[(((size_t)n * g.I + i) * g.Hh + hi) * g.Wh + wi] = v;
  }
};

// ---- image-side first_down: A from NCHW fp32; epilogue bias + lrelu
struct FirstDownA {
  const float* x; Geo g;
  __device__ float operator()(int, int m, int k) const {
    int wo, t, ho, n;
    rg_divmod(m, g.Wl, g.sWl, t, wo); rg_divmod(t, g.Hl, g.sHl, n, ho);
    int ci = k >> 4, tap = k & 15;
    int hi = 2 * ho - 1 + (tap >> 2), wi = 2 * wo - 1 + (tap & 3);
    if (hi < 0 || hi >= g.Hh || wi < 0 || wi >= g.Wh) return 0.f;
    return x[(((size_t)n * g.I + ci) * g.Hh + hi) * g.Wh + wi];
  }
};
template <typename T> struct BiasActC {
  T* y; const float* bias; float slope; int ld;
  __device__ void operator()(int, int, int m, int n, float v) const {
    if (bias) v += bias[n];
    Elem<T>::st(y + (size_t)m * ld + n, lrelu_f(v, slope));
  }
};

// ---- wgrad: M = O, Ncols = 16*I, K = N*Hl*Wl pixels.  col = tap*I + i (tap-major dW[O][16][I]) for the 4x4 conv
// layers, col = i*16 + tap (PyTorch layout) for the image-side layer
template <typename T> struct WgradA {
  const T* low; Geo g;
  __device__ float operator()(int, int o, int pix) const { return Elem<T>::ld(low + (size_t)pix * g.O + o); }
};
template <typename T> struct WgradB {
  const T* high; Geo g;
  __device__ float operator()(int, int pix, int col) const {
    int wo, t, ho, n, tap, i;
    rg_divmod(pix, g.Wl, g.sWl, t, wo); rg_divmod(t, g.Hl, g.sHl, n, ho); rg_divmod(col, g.I, g.sI, tap, i);
    int hi = 2 * ho - 1 + (tap >> 2), wi = 2 * wo - 1 + (tap & 3);
    if (hi < 0 || hi >= g.Hh || wi < 0 || wi >= g.Wh) return 0.f;
    return Elem<T>::ld(high + (((size_t)n * g.Hh + hi) * g.Wh + wi) * g.I + i);
  }
};
struct WgradBNchw {
  const float* high; Geo g;
  __device__ float operator()(int, int pix, int col) const {
    int wo, t, ho, n;
    rg_divmod(pix, g.Wl, g.sWl, t, wo); rg_divmod(t, g.Hl, g.sHl, n, ho);
    int i = col >> 4, tap = col & 15;
    int hi = 2 * ho - 1 + (tap >> 2), wi = 2 * wo - 1 + (tap & 3);
    if (hi < 0 || hi >= g.Hh || wi < 0 || wi >= g.Wh) return 0.f;
    return high[(((size_t)n * g.I + i) * g.Hh + hi) * g.Wh + wi];
  }
};
struct SlabC {  // slab[zs][m*ld + n]
  float* slab; size_t slab_elems; int ld;
  __device__ void operator()(int, int zs, int m, int n, float v) const {
    slab[(size_t)zs * slab_elems + (size_t)m * ld + n] = v;
  }
};

// ---- g0: fwd  A(n,e) = z, B(e, col=tap*C+c) = w[e][c][tap];  wgrad A(e,n)=z, B(n, col'=c*16+tap)=gy[n][tap*C+c]
template <typename T> struct G0A {
  const float* z; int E;
  __device__ float operator()(int, int n, int e) const { return Elem<T>::round(z[(size_t)n * E + e]); }
};
template <typename T> struct G0B {
  const float* w; int C; int sC;
  __device__ float operator()(int, int e, int col) const {
    int tap, c;
    rg_divmod(col, C, sC, tap, c);
    return Elem<T>::round(w[((size_t)e * C + c) * 16 + tap]);
  }
};
// The same GEMM with its columns in MEMORY order of w[E][C][16] (col' = c * 16 + tap: consecutive threads read consecutive
// floats; with col = tap * C + c a 64-lane load touched 64 different 64-byte segments, 16 x the weight bytes through L2:
// 3.25 GB fetched for a 134 MB tensor, 500 us) and the permutation in the store (the output is 16 x smaller than w).
template <typename T> struct G0Bm {
  const float* w; int C;
  __device__ float operator()(int, int e, int colm) const { return Elem<T>::round(w[(size_t)e * C * 16 + colm]); }
};
template <typename T> struct G0Cm {
  T* y; int C;
  __device__ void operator()(int, int, int m, int colm, float v) const {
    Elem<T>::st(y + (size_t)m * 16 * C + (size_t)(colm & 15) * C + (colm >> 4), v);
  }
};
template <typename T> struct G0WA {
  const float* z; int E;
  __device__ float operator()(int, int e, int n) const { return Elem<T>::round(z[(size_t)n * E + e]); }
};
template <typename T> struct G0WB {
  const T* gy; int C;
  __device__ float operator()(int, int n, int col) const {
    int c = col >> 4, tap = col & 15;
    return Elem<T>::ld(gy + (size_t)n * 16 * C + (size_t)tap * C + c);
  }
};
struct AccumC {
  float* dst; int ld; int accumulate;
  __device__ void operator()(int, int, int m, int n, float v) const {
    size_t idx = (size_t)m * ld + n;
    dst[idx] = accumulate ? dst[idx] + v : v;
  }
};

// ---- linear
struct LinA {
  const float* x; int ldx;
  __device__ float operator()(int, int m, int k) const { return x[(size_t)m * ldx + k]; }
};
struct LinB {
  const float* w; int K;
  __device__ float operator()(int, int k, int j) const { return w[(size_t)j * K + k]; }
};
struct LinC {
  float* y; int ldy; const float* scale; const float* shift; float slope;
  __device__ void operator()(int, int, int m, int j, float v) const {
    if (scale) v *= scale[j];
    if (shift) v += shift[j];
    y[(size_t)m * ldy + j] = lrelu_f(v, slope);
  }
};

// ---- two-phase forms (Op<>, see gemm_mfma32_kernel) of the gathered conv operands.  Element offsets are 32-bit (fits32: the
// launcher keeps the vector kernel for a tensor of 2^31 elements or more) and, in the P2 instantiation, built with shifts:
// Hh = 2 Hl and Wh = 2 Wl are powers of two with log2 = sHl, sWl (the "+ 1" of the encoding), log2(I) = sI - 1.
template <typename T> struct Op<DownA<T>, true> {
  struct R { int base, h0, w0; };
  struct C { int off, dh, dw; };
  static __device__ __forceinline__ bool pow2(const DownA<T>& f) { return f.g.sWl && f.g.sHl && f.g.sI && f.g.I >= MB_K; }
  static bool fits32(const DownA<T>& f) { return (long long)f.g.N * f.g.Hh * f.g.Wh * f.g.I < (1ll << 29); }
  template <bool P2> static __device__ __forceinline__ R row(const DownA<T>& f, int, int m) {
    int wo, t, ho, n;
    rg_divmod_t<P2>(m, f.g.Wl, f.g.sWl, t, wo); rg_divmod_t<P2>(t, f.g.Hl, f.g.sHl, n, ho);
    const int h0 = 2 * ho - 1, w0 = 2 * wo - 1;
    // (multiplications by powers of two, not shifts: h0 / w0 are -1 on the border, and a left shift of a negative int is undefined)
    const int base = P2 ? (((n << f.g.sHl) + h0) * (1 << f.g.sWl) + w0) * (1 << (f.g.sI - 1)) : ((n * f.g.Hh + h0) * f.g.Wh + w0) * f.g.I;
    return {base, h0, w0};
  }
  template <bool P2> static __device__ __forceinline__ C col(const DownA<T>& f, int, int k) {
    int tap, ci;
    rg_divmod_t<P2>(k, f.g.I, f.g.sI, tap, ci);
    const int dh = tap >> 2, dw = tap & 3;
    const int off = P2 ? ((((dh << f.g.sWl) + dw) << (f.g.sI - 1)) + ci) : (dh * f.g.Wh + dw) * f.g.I + ci;
    return {off, dh, dw};
  }
  static __device__ __forceinline__ float at(const DownA<T>& f, int, const R& r, const C& c, bool& ok) {
    const int hi = r.h0 + c.dh, wi = r.w0 + c.dw;
    ok = (int)((unsigned)hi < (unsigned)f.g.Hh) & (int)((unsigned)wi < (unsigned)f.g.Wh);
    return Elem<T>::ld(f.x + (ok ? r.base + c.off : 0));        // branch-free: a padding tap reads element 0
  }
};
template <typename T> struct Op<UpA<T>, true> {
  struct R { int base, hq, wq; };
  struct C { int off, dh, dw; };
  static __device__ __forceinline__ bool pow2(const UpA<T>& f) { return f.g.sWl && f.g.sHl && f.g.sO && f.g.O >= MB_K; }
  static bool fits32(const UpA<T>& f) { return (long long)f.g.N * f.g.Hl * f.g.Wl * f.g.O < (1ll << 29); }
  template <bool P2> static __device__ __forceinline__ R row(const UpA<T>& f, int, int m) {
    int wq, t, hq, n;
    rg_divmod_t<P2>(m, f.g.Wl, f.g.sWl, t, wq); rg_divmod_t<P2>(t, f.g.Hl, f.g.sHl, n, hq);
    const int base = P2 ? ((((n << (f.g.sHl - 1)) + hq) << (f.g.sWl - 1)) + wq) << (f.g.sO - 1) : ((n * f.g.Hl + hq) * f.g.Wl + wq) * f.g.O;
    return {base, hq, wq};
  }
  template <bool P2> static __device__ __forceinline__ C col(const UpA<T>& f, int zb, int k) {
    int o, t4;
#ifdef RG_UP_KORDER_OLD
    o = k >> 2; t4 = k & 3;
#else
    rg_divmod_t<P2>(k, f.g.O, f.g.sO, t4, o);
#endif
    int kh, kw, sh, sw;
    up_tap(zb >> 1, t4 >> 1, 0, kh, sh);          // source offset of the tap along each axis (q = 0: src = the offset)
    up_tap(zb & 1, t4 & 1, 0, kw, sw);
    const int off = P2 ? (((sh << (f.g.sWl - 1)) + sw) << (f.g.sO - 1)) + o : (sh * f.g.Wl + sw) * f.g.O + o;
    return {off, sh, sw};
  }
  static __device__ __forceinline__ float at(const UpA<T>& f, int, const R& r, const C& c, bool& ok) {
    const int ho = r.hq + c.dh, wo = r.wq + c.dw;
    ok = (int)((unsigned)ho < (unsigned)f.g.Hl) & (int)((unsigned)wo < (unsigned)f.g.Wl);
    return Elem<T>::ld(f.x + (ok ? r.base + c.off : 0));
  }
};
template <typename T> struct Op<WgradB<T>, false> {       // B side: operator()(zb, pix = k, col = n)
  struct R { int off, kh, kw; };                          // of the output column (tap, i)
  struct C { int base, h0, w0; };                         // of the pixel
  static __device__ __forceinline__ bool pow2(const WgradB<T>& f) { return f.g.sWl && f.g.sHl && f.g.sI; }
  static bool fits32(const WgradB<T>& f) { return (long long)f.g.N * f.g.Hh * f.g.Wh * f.g.I < (1ll << 31); }
  template <bool P2> static __device__ __forceinline__ R row(const WgradB<T>& f, int, int col) {
    int tap, i;
    rg_divmod_t<P2>(col, f.g.I, f.g.sI, tap, i);
    const int kh = tap >> 2, kw = tap & 3;
    const int off = P2 ? ((((kh << f.g.sWl) + kw) << (f.g.sI - 1)) + i) : (kh * f.g.Wh + kw) * f.g.I + i;
    return {off, kh, kw};
  }
  template <bool P2> static __device__ __forceinline__ C col(const WgradB<T>& f, int, int pix) {
    int wo, t, ho, n;
    rg_divmod_t<P2>(pix, f.g.Wl, f.g.sWl, t, wo); rg_divmod_t<P2>(t, f.g.Hl, f.g.sHl, n, ho);
    const int h0 = 2 * ho - 1, w0 = 2 * wo - 1;
    // (multiplications by powers of two, not shifts: h0 / w0 are -1 on the border, and a left shift of a negative int is undefined)
    const int base = P2 ? (((n << f.g.sHl) + h0) * (1 << f.g.sWl) + w0) * (1 << (f.g.sI - 1)) : ((n * f.g.Hh + h0) * f.g.Wh + w0) * f.g.I;
    return {base, h0, w0};
  }
  static __device__ __forceinline__ float at(const WgradB<T>& f, int, const R& r, const C& c, bool& ok) {
    const int hi = c.h0 + r.kh, wi = c.w0 + r.kw;
    ok = (int)((unsigned)hi < (unsigned)f.g.Hh) & (int)((unsigned)wi < (unsigned)f.g.Wh);
    return Elem<T>::ld(f.high + (ok ? c.base + r.off : 0));
  }
};
// the weight operands: a row base per output column, a k offset per k position
template <typename T> struct Op<DownB<T>, false> {        // operator()(zb, k, o) = w[o * 16 I + k]
  struct R { int base; };
  struct C { int k; };
  static __device__ __forceinline__ bool pow2(const DownB<T>&) { return true; }
  static bool fits32(const DownB<T>& f) { return (long long)f.g.O * f.g.I * 16 < (1ll << 29); }
  template <bool P2> static __device__ __forceinline__ R row(const DownB<T>& f, int, int o) { return {o * f.g.I * 16}; }
  template <bool P2> static __device__ __forceinline__ C col(const DownB<T>&, int, int k) { return {k}; }
  static __device__ __forceinline__ float at(const DownB<T>& f, int, const R& r, const C& c, bool& ok) {
    ok = true;
    return Elem<T>::round(f.w[r.base + c.k]);
  }
};
template <typename T> struct Op<UpB<T>, false> {          // operator()(zb, k, i) = w[(o * 16 + kh * 4 + kw) * I + i]
  struct R { int i; };
  struct C { int off; };
  static __device__ __forceinline__ bool pow2(const UpB<T>& f) { return f.g.sO != 0 && f.g.O >= MB_K; }
  static bool fits32(const UpB<T>& f) { return (long long)f.g.O * f.g.I * 16 < (1ll << 29); }
  template <bool P2> static __device__ __forceinline__ R row(const UpB<T>&, int, int i) { return {i}; }
  template <bool P2> static __device__ __forceinline__ C col(const UpB<T>& f, int zb, int k) {
    int o, t4, kh, kw, d;
#ifdef RG_UP_KORDER_OLD
    o = k >> 2; t4 = k & 3;
#else
    rg_divmod_t<P2>(k, f.g.O, f.g.sO, t4, o);
#endif
    up_tap(zb >> 1, t4 >> 1, 0, kh, d);
    up_tap(zb & 1, t4 & 1, 0, kw, d);
    return {(o * 16 + kh * 4 + kw) * f.g.I};
  }
  static __device__ __forceinline__ float at(const UpB<T>& f, int, const R& r, const C& c, bool& ok) {
    ok = true;
    return Elem<T>::round(f.w[c.off + r.i]);
  }
};
// (low[pix][o]: one multiply per element; in the P2 form a shift)
template <typename T> struct Op<WgradA<T>, true> {
  struct R { int o; };
  struct C { int poff; };
  static __device__ __forceinline__ bool pow2(const WgradA<T>& f) { return f.g.sO != 0; }
  static bool fits32(const WgradA<T>& f) { return (long long)f.g.N * f.g.Hl * f.g.Wl * f.g.O < (1ll << 29); }
  template <bool P2> static __device__ __forceinline__ R row(const WgradA<T>&, int, int o) { return {o}; }
  template <bool P2> static __device__ __forceinline__ C col(const WgradA<T>& f, int, int pix) {
    return {P2 ? pix << (f.g.sO - 1) : pix * f.g.O};
  }
  static __device__ __forceinline__ float at(const WgradA<T>& f, int, const R& r, const C& c, bool& ok) {
    ok = true;
    return Elem<T>::ld(f.low + (c.poff + r.o));
  }
};

// ----------------------------------------------------------------------------------------------
// gemm_mfma32s_kernel: the matrix-core GEMM above for launches whose BOTH operands are STRUCTURED (power-of-two geometry):
// the operand's K axis is a sequence of segments (a tap's channels, a weight row, a row of output pixels) inside which the
// 16 k of a k-tile are consecutive elements at a uniform distance.  A thread keeps, per fetch slot,
//     base (bytes, everything that depends on the slot's row and k position) and m (its "invalid" bits),
// the k-loop derives per k-tile -- on the SCALAR unit --
//     id (segment), so (byte offset inside the segment: the load's scalar offset), add (bytes), f (which invalid bits count),
// and a slot's load offset is  (base + add) | (min(m & f, 1) << 31):  bit 31 = beyond the buffer's 2 GB range, the load
// returns 0 -- that is the zero padding of the convolution, the rows beyond M / N and the k tail, with no predicate, no
// select and no branch around any load; the offsets are rebuilt only when id changes (4 vector instructions per slot).
// The two-phase kernel above spent ~570 non-matrix instructions per k-tile of 32 MFMAs on the same launches (per-slot
// exec-mask branches the compiler built from its selects); this one ~130.  It remains the kernel for every other launch.
// ----------------------------------------------------------------------------------------------
struct SSlot { int base; unsigned m; };
struct SSeg { int id; unsigned so; int add; unsigned f; };
template <class F> struct SOp { static constexpr bool OK = false; };

template <> struct SOp<DownA<float>> {                    // A(m = (n, ho, wo), k = tap * I + c) = x[n][2ho-1+kh][2wo-1+kw][c]
  using F = DownA<float>;
  static constexpr bool OK = true;
  static bool ok(const F& f, int, int) { return f.g.sWl && f.g.sHl && f.g.sI && f.g.I >= MB_K && (long long)f.g.N * f.g.Hh * f.g.Wh * f.g.I < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.x; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int m, bool ok, int ks) {
    const int wo = m & (f.g.Wl - 1), t = m >> (f.g.sWl - 1), ho = t & (f.g.Hl - 1), n = t >> (f.g.sHl - 1);
    const int base = (((n * f.g.Hh + 2 * ho - 1) * f.g.Wh + 2 * wo - 1) * f.g.I + ks) * 4;
    // invalid bits: kh in bits 0-3 (row 2ho-1+kh outside the image), kw in bits 4-7
    const unsigned im = (ho == 0 ? 1u : 0u) | (ho == f.g.Hl - 1 ? 8u : 0u) | (wo == 0 ? 0x10u : 0u) | (wo == f.g.Wl - 1 ? 0x80u : 0u);
    return {base, ok ? im : 0xffu};
  }
  static __device__ __forceinline__ void seg(const F& f, int, int k0, SSeg& u) {
    const int tap = k0 >> (f.g.sI - 1), dh = tap >> 2, dw = tap & 3;
    u.id = tap;
    u.so = (unsigned)(k0 & (f.g.I - 1)) * 4u;
    u.add = ((dh * f.g.Wh + dw) * f.g.I) * 4;
    u.f = (1u << dh) | (0x10u << dw);
  }
};
template <> struct SOp<DownB<float>> {                    // B(k, o) = w[o * 16 I + k]
  using F = DownB<float>;
  static constexpr bool OK = true;
  static bool ok(const F& f, int, int) { return (long long)f.g.O * f.g.I * 16 < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.w; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int o, bool ok, int ks) { return {(o * f.g.I * 16 + ks) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F&, int, int k0, SSeg& u) { u.id = 0; u.so = (unsigned)k0 * 4u; u.add = 0; u.f = 1u; }
};
template <> struct SOp<UpA<float>> {                      // A(m = (n, hq, wq), k = t4 * O + o) = x[n][hq + sh][wq + sw][o], class zb
  using F = UpA<float>;
#ifndef RG_UP_KORDER_OLD
  static constexpr bool OK = true;
#else
  static constexpr bool OK = false;
#endif
  static bool ok(const F& f, int, int) { return f.g.sWl && f.g.sHl && f.g.sO && f.g.O >= MB_K && (long long)f.g.N * f.g.Hl * f.g.Wl * f.g.O < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.x; }
  static __device__ __forceinline__ SSlot slot(const F& f, int zb, int m, bool ok, int ks) {
    const int wq = m & (f.g.Wl - 1), t = m >> (f.g.sWl - 1), hq = t & (f.g.Hl - 1), n = t >> (f.g.sHl - 1);
    const int base = (((n * f.g.Hl + hq) * f.g.Wl + wq) * f.g.O + ks) * 4;
    // parity 0: tap a = 1 reads q - 1; parity 1: tap a = 0 reads q + 1 (up_tap).  Invalid bits: a in bits 0-1, c in bits 4-5
    const int ph = zb >> 1, pw = zb & 1;
    const unsigned im = (ph == 0 ? (hq == 0 ? 2u : 0u) : (hq == f.g.Hl - 1 ? 1u : 0u)) |
                        (pw == 0 ? (wq == 0 ? 0x20u : 0u) : (wq == f.g.Wl - 1 ? 0x10u : 0u));
    return {base, ok ? im : 0xffu};
  }
  static __device__ __forceinline__ void seg(const F& f, int zb, int k0, SSeg& u) {
    const int t4 = k0 >> (f.g.sO - 1);
    int kh, kw, sh, sw;
    up_tap(zb >> 1, t4 >> 1, 0, kh, sh);
    up_tap(zb & 1, t4 & 1, 0, kw, sw);
    u.id = t4;
    u.so = (unsigned)(k0 & (f.g.O - 1)) * 4u;
    u.add = ((sh * f.g.Wl + sw) * f.g.O) * 4;
    u.f = (1u << (t4 >> 1)) | (0x10u << (t4 & 1));
  }
};
template <> struct SOp<UpB<float>> {                      // B(k = t4 * O + o, i) = w[(o * 16 + kh * 4 + kw) * I + i]
  using F = UpB<float>;
#ifndef RG_UP_KORDER_OLD
  static constexpr bool OK = true;
#else
  static constexpr bool OK = false;
#endif
  static bool ok(const F& f, int, int) { return f.g.sO != 0 && f.g.O >= MB_K && (long long)f.g.O * f.g.I * 16 < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.w; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int i, bool ok, int ks) { return {(ks * 16 * f.g.I + i) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F& f, int zb, int k0, SSeg& u) {
    const int t4 = k0 >> (f.g.sO - 1);
    int kh, kw, d;
    up_tap(zb >> 1, t4 >> 1, 0, kh, d);
    up_tap(zb & 1, t4 & 1, 0, kw, d);
    u.id = 0;
    u.so = (unsigned)(((k0 & (f.g.O - 1)) * 16 + kh * 4 + kw) * f.g.I) * 4u;
    u.add = 0;
    u.f = 1u;
  }
};
template <> struct SOp<WgradA<float>> {                   // A(o, k = pixel) = low[pixel * O + o]
  using F = WgradA<float>;
  static constexpr bool OK = true;
  static bool ok(const F& f, int, int) { return (long long)f.g.N * f.g.Hl * f.g.Wl * f.g.O < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.low; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int o, bool ok, int ks) { return {(ks * f.g.O + o) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F& f, int, int k0, SSeg& u) { u.id = 0; u.so = (unsigned)(k0 * f.g.O) * 4u; u.add = 0; u.f = 1u; }
};
template <> struct SOp<WgradB<float>> {                   // B(k = pixel (n, ho, wo), col = tap * I + i) = high[n][2ho-1+kh][2wo-1+kw][i]
  using F = WgradB<float>;
  // A k-tile = 16 consecutive output pixels: a piece of one output row (Wl >= 16; the piece's first wo goes into the scalar
  // offset) or 16 / Wl whole rows of one image (Wl < 16, Hl * Wl >= 16).  The segment is the k-tile itself.
  static constexpr bool OK = true;
  static bool ok(const F& f, int, int) {
#ifdef RG_WGRADB_NARROW_OFF
    if (f.g.Wl < MB_K) return false;
#endif
    return f.g.sWl && f.g.sHl && f.g.sI && f.g.Hl * f.g.Wl >= MB_K && (long long)f.g.N * f.g.Hh * f.g.Wh * f.g.I < (1ll << 29);
  }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.high; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int col, bool ok, int ks) {
    const int tap = col >> (f.g.sI - 1), i = col & (f.g.I - 1), kh = tap >> 2, kw = tap & 3;
    if (f.g.Wl >= MB_K) {
      const int base = ((kh * f.g.Wh + kw - 1 + 2 * ks) * f.g.I + i) * 4;
      // invalid bits: 0 = needs ho > 0, 1 = needs ho < Hl - 1, 2 = needs wo0 > 0 (first pixel of the tile), 3 = needs wo0 < Wl - 16 (last)
      const unsigned im = (kh == 0 ? 1u : 0u) | (kh == 3 ? 2u : 0u) | ((int)(kw == 0) & (int)(ks == 0) ? 4u : 0u) |
                          ((int)(kw == 3) & (int)(ks == MB_K - 1) ? 8u : 0u);
      return {base, ok ? im : 0x10u};
    }
    // rows r = ks / Wl of the tile, pixel wo = ks % Wl: the column padding is known per slot (bit 4 = always), the row padding
    // needs the tile's first row ho0: bit 0 = needs ho0 > 0 (first row of the tile), bit 1 = needs ho0 < Hl - rows (last)
    const int wo = ks & (f.g.Wl - 1), r = ks >> (f.g.sWl - 1), rows = MB_K >> (f.g.sWl - 1);
    const int base = (((2 * r + kh) * f.g.Wh + 2 * wo - 1 + kw) * f.g.I + i) * 4;
    const bool wbad = ((int)(kw == 0) & (int)(wo == 0)) | ((int)(kw == 3) & (int)(wo == f.g.Wl - 1));
    const unsigned im = ((int)(kh == 0) & (int)(r == 0) ? 1u : 0u) | ((int)(kh == 3) & (int)(r == rows - 1) ? 2u : 0u) | (wbad ? 0x10u : 0u);
    return {base, ok ? im : 0x10u};
  }
  static __device__ __forceinline__ void seg(const F& f, int, int k0, SSeg& u) {
    u.id = k0 >> 4;
    if (f.g.Wl >= MB_K) {
      const int wo0 = k0 & (f.g.Wl - 1), t = k0 >> (f.g.sWl - 1), ho = t & (f.g.Hl - 1), n = t >> (f.g.sHl - 1);
      // (the buffer's range check sees base + add only, not so: a valid slot's base + add must not be negative.  The one that
      // would be -- image 0, input row 0, kw = 0, the tile's first pixel, valid when wo0 > 0 -- is kept at >= 0 by moving one
      // pixel of the scalar offset into add.)
      const int mv = wo0 ? f.g.I : 0;
      u.so = (unsigned)(2 * wo0 * f.g.I - mv) * 4u;
      u.add = ((n * f.g.Hh + 2 * ho - 1) * f.g.Wh * f.g.I + mv) * 4;
      u.f = (ho == 0 ? 1u : 0u) | (ho == f.g.Hl - 1 ? 2u : 0u) | (wo0 == 0 ? 4u : 0u) | (wo0 == f.g.Wl - MB_K ? 8u : 0u) | 0x10u;
    } else {
      const int t = k0 >> (f.g.sWl - 1), ho0 = t & (f.g.Hl - 1), n = t >> (f.g.sHl - 1), rows = MB_K >> (f.g.sWl - 1);
      u.so = 0u;
      u.add = ((n * f.g.Hh + 2 * ho0 - 1) * f.g.Wh * f.g.I) * 4;
      u.f = (ho0 == 0 ? 1u : 0u) | (ho0 == f.g.Hl - rows ? 2u : 0u) | 0x10u;
    }
  }
};
template <> struct SOp<FirstDownA> {                      // A(m = (n, ho, wo), k = ci * 16 + tap) = x[n][ci][2ho-1+kh][2wo-1+kw] (NCHW fp32 image)
  using F = FirstDownA;                                   // a k-tile = the 16 taps of one input channel: a slot's tap is fixed,
  static constexpr bool OK = true;                        // its padding a constant of the slot; the channel is the scalar offset
  static bool ok(const F& f, int, int) { return f.g.sWl && f.g.sHl && (long long)f.g.N * f.g.I * f.g.Hh * f.g.Wh < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.x; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int m, bool ok, int ks) {
    const int wo = m & (f.g.Wl - 1), t = m >> (f.g.sWl - 1), ho = t & (f.g.Hl - 1), n = t >> (f.g.sHl - 1);
    const int hi = 2 * ho - 1 + (ks >> 2), wi = 2 * wo - 1 + (ks & 3);
    const bool v = (int)ok & (int)((unsigned)hi < (unsigned)f.g.Hh) & (int)((unsigned)wi < (unsigned)f.g.Wh);
    return {((n * f.g.I * f.g.Hh + hi) * f.g.Wh + wi) * 4, v ? 0u : 1u};
  }
  static __device__ __forceinline__ void seg(const F& f, int, int k0, SSeg& u) {
    u.id = 0; u.so = (unsigned)((k0 >> 4) * f.g.Hh * f.g.Wh) * 4u; u.add = 0; u.f = 1u;
  }
};
template <> struct SOp<WgradBNchw> {                      // B(k = pixel (n, ho, wo), col = i * 16 + tap) = x[n][i][2ho-1+kh][2wo-1+kw]
  using F = WgradBNchw;                                   // (the image-side layer's weight gradient; output rows of >= 16 pixels)
  static constexpr bool OK = true;
  static bool ok(const F& f, int, int) { return f.g.sWl && f.g.sHl && f.g.Wl >= MB_K && (long long)f.g.N * f.g.I * f.g.Hh * f.g.Wh < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.high; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int col, bool ok, int ks) {
    const int i = col >> 4, kh = (col >> 2) & 3, kw = col & 3;
    const int base = ((i * f.g.Hh + kh) * f.g.Wh + kw - 1 + 2 * ks) * 4;
    const unsigned im = (kh == 0 ? 1u : 0u) | (kh == 3 ? 2u : 0u) | ((int)(kw == 0) & (int)(ks == 0) ? 4u : 0u) |
                        ((int)(kw == 3) & (int)(ks == MB_K - 1) ? 8u : 0u);
    return {base, ok ? im : 0x10u};
  }
  static __device__ __forceinline__ void seg(const F& f, int, int k0, SSeg& u) {
    const int wo0 = k0 & (f.g.Wl - 1), t = k0 >> (f.g.sWl - 1), ho = t & (f.g.Hl - 1), n = t >> (f.g.sHl - 1);
    const int mv = wo0 ? 1 : 0;                           // (as SOp<WgradB>: keeps a valid slot's base + add >= 0)
    u.id = k0 >> 4;
    u.so = (unsigned)(2 * wo0 - mv) * 4u;
    u.add = ((n * f.g.I * f.g.Hh + 2 * ho - 1) * f.g.Wh + mv) * 4;
    u.f = (ho == 0 ? 1u : 0u) | (ho == f.g.Hl - 1 ? 2u : 0u) | (wo0 == 0 ? 4u : 0u) | (wo0 == f.g.Wl - MB_K ? 8u : 0u) | 0x10u;
  }
};
template <> struct SOp<G0A<float>> {                      // A(n, e) = z[n * E + e]
  using F = G0A<float>;
  static constexpr bool OK = true;
  static bool ok(const F& f, int rows, int) { return (long long)rows * f.E < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.z; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int m, bool ok, int ks) { return {(m * f.E + ks) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F&, int, int k0, SSeg& u) { u.id = 0; u.so = (unsigned)k0 * 4u; u.add = 0; u.f = 1u; }
};
template <> struct SOp<G0Bm<float>> {                     // B(e, col') = w[e * 16 C + col']
  using F = G0Bm<float>;
  static constexpr bool OK = true;
  static bool ok(const F&, int cols, int K) { return (long long)cols * K < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.w; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int colm, bool ok, int ks) { return {(ks * f.C * 16 + colm) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F& f, int, int k0, SSeg& u) { u.id = 0; u.so = (unsigned)(k0 * f.C * 16) * 4u; u.add = 0; u.f = 1u; }
};
template <> struct SOp<LinA> {                            // A(m, k) = x[m * ldx + k]
  using F = LinA;
  static constexpr bool OK = true;
  static bool ok(const F& f, int rows, int) { return (long long)rows * f.ldx < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.x; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int m, bool ok, int ks) { return {(m * f.ldx + ks) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F&, int, int k0, SSeg& u) { u.id = 0; u.so = (unsigned)k0 * 4u; u.add = 0; u.f = 1u; }
};
template <> struct SOp<LinB> {                            // B(k, j) = w[j * K + k]
  using F = LinB;
  static constexpr bool OK = true;
  static bool ok(const F& f, int rows, int) { return (long long)rows * f.K < (1ll << 29); }
  static __device__ __forceinline__ const void* ptr(const F& f) { return f.w; }
  static __device__ __forceinline__ SSlot slot(const F& f, int, int j, bool ok, int ks) { return {(j * f.K + ks) * 4, ok ? 0u : 1u}; }
  static __device__ __forceinline__ void seg(const F&, int, int k0, SSeg& u) { u.id = 0; u.so = (unsigned)k0 * 4u; u.add = 0; u.f = 1u; }
};

template <int TM, int TN, bool A_KFAST, bool B_KFAST, class FA, class FB, class SC>
__global__ __launch_bounds__(256, 3) void gemm_mfma32s_kernel(FA fa, FB fb, SC sc, int M, int N, int K, int lgb, int klen) {
  constexpr int BM = 2 * TM, BN = 2 * TN, IA = TM / 32, IB = TN / 32;
  constexpr int NSA = BM * MB_K / 256, NSB = BN * MB_K / 256;
  constexpr int KSA = 256 / BM, KSB = 256 / BN;
  constexpr int RP = 256 / MB_K;
  constexpr unsigned TAIL = 0x100u;                       // invalid bit of the slots beyond K in the last k-tile
  static_assert(MB_K == 16, "the structured operands assume 16-deep k-tiles");
  __shared__ float As[2][MB_K][BM + 4];
  __shared__ float Bs[2][MB_K][BN + 4];
  using SA = SOp<FA>;
  using SB = SOp<FB>;
  const int bm = blockIdx.x * BM, bn = blockIdx.y * BN;
  // blockIdx.z = zs << lgb | zb (batch count = 2^lgb: 1, or the 4 output classes of the transposed conv).  Shift / mask, not
  // the / nsplit of the kernels above: an integer division is a vector-unit sequence, and in the transposed-conv and weight-
  // gradient instantiations the compiler then kept everything derived from it (k0, every load's scalar offset) in vector
  // registers, where instruction legalisation wraps each buffer load in a waterfall loop (readfirstlane / compare / masked
  // load / branch per load: the transposed convs ran at 97 TFLOP/s against the forward convs' 120).
  const int zb = blockIdx.z & ((1 << lgb) - 1), zs = blockIdx.z >> lgb;
  const int k_begin = zs * klen;
  const int k_end = min(K, k_begin + klen);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  mb_f32x16 acc[IA][IB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ktail = K & (MB_K - 1);                       // 0: no partial k-tile
  SSlot sla[NSA], slb[NSB];
#pragma unroll
  for (int i = 0; i < NSA; ++i) {
    const int mm = A_KFAST ? tid / MB_K + RP * i : (tid & (BM - 1));
    const int ks = A_KFAST ? (tid & (MB_K - 1)) : tid / BM + KSA * i;
    const bool ok = bm + mm < M;
    sla[i] = SA::slot(fa, zb, ok ? bm + mm : 0, ok, ks);
    sla[i].m |= ((int)(ktail != 0) & (int)(ks >= ktail)) ? TAIL : 0u;
  }
#pragma unroll
  for (int i = 0; i < NSB; ++i) {
    const int nn = B_KFAST ? tid / MB_K + RP * i : (tid & (BN - 1));
    const int ks = B_KFAST ? (tid & (MB_K - 1)) : tid / BN + KSB * i;
    const bool ok = bn + nn < N;
    slb[i] = SB::slot(fb, zb, ok ? bn + nn : 0, ok, ks);
    slb[i].m |= ((int)(ktail != 0) & (int)(ks >= ktail)) ? TAIL : 0u;
  }
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)SA::ptr(fa), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)SB::ptr(fb), 0, 0x7ffffff0, 0x00020000);
  unsigned voa[NSA], vob[NSB];
  float ra[NSA], rb[NSB];
  int ida = -1, idb = -1;
  auto issue = [&](int k0) __attribute__((always_inline)) {
    const bool tail = k0 + MB_K > K;                      // uniform
    SSeg ua, ub;
    SA::seg(fa, zb, k0, ua);
    SB::seg(fb, zb, k0, ub);
    if (tail) { ua.id |= 0x40000000; ua.f |= TAIL; ub.id |= 0x40000000; ub.f |= TAIL; }
    if (ua.id != ida) {
      ida = ua.id;
#pragma unroll
      for (int i = 0; i < NSA; ++i) voa[i] = (unsigned)(sla[i].base + ua.add) | (min(sla[i].m & ua.f, 1u) << 31);
    }
    if (ub.id != idb) {
      idb = ub.id;
#pragma unroll
      for (int i = 0; i < NSB; ++i) vob[i] = (unsigned)(slb[i].base + ub.add) | (min(slb[i].m & ub.f, 1u) << 31);
    }
#pragma unroll
    for (int i = 0; i < NSA; ++i) ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, (int)voa[i], (int)ua.so, 0));
#pragma unroll
    for (int i = 0; i < NSB; ++i) rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsb, (int)vob[i], (int)ub.so, 0));
  };
  auto stash = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NSA; ++i) {
      if (A_KFAST) As[st][tid & (MB_K - 1)][tid / MB_K + RP * i] = ra[i]; else As[st][tid / BM + KSA * i][tid & (BM - 1)] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
      if (B_KFAST) Bs[st][tid & (MB_K - 1)][tid / MB_K + RP * i] = rb[i]; else Bs[st][tid / BN + KSB * i][tid & (BN - 1)] = rb[i];
    }
  };
  if (k_begin < k_end) { issue(k_begin); stash(0); }
  __syncthreads();
  int cur = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += MB_K) {
    const bool more = k0 + MB_K < k_end;
    if (more) issue(k0 + MB_K);                    // global loads of the next k-tile fly under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    float a[2][IA], b[2][IB];                      // the fragments of step st + 1 are read while the MFMAs of step st run
#pragma unroll
    for (int i = 0; i < IA; ++i) a[0][i] = As[cur][lh][wm * TM + 32 * i + lr];
#pragma unroll
    for (int j = 0; j < IB; ++j) b[0][j] = Bs[cur][lh][wn * TN + 32 * j + lr];
#pragma unroll
    for (int st = 0; st < MB_K / 2; ++st) {
      if (st + 1 < MB_K / 2) {
#pragma unroll
        for (int i = 0; i < IA; ++i) a[(st + 1) & 1][i] = As[cur][2 * st + 2 + lh][wm * TM + 32 * i + lr];
#pragma unroll
        for (int j = 0; j < IB; ++j) b[(st + 1) & 1][j] = Bs[cur][2 * st + 2 + lh][wn * TN + 32 * j + lr];
      }
#pragma unroll
      for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < IB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][i], b[st & 1][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);             // the loads stay above the MFMAs, their first use (the stash) below
    if (more) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = bn + wn * TN + j * 32 + lr;
        if (m < M && n < N) sc(zb, zs, m, n, acc[i][j][r]);
      }
}

// ----------------------------------------------------------------------------------------------
// gemm_bf16x3s_kernel: the structured fp32 GEMM above on the BF16 matrix cores (option f32mma = 2, the default).  An fp32 value is the exact
// sum of three bf16 numbers,  v = h + m + l  (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m): 8 + 8 + 8 significant bits,
// round-to-nearest each, the residuals are exact in fp32), a product of two bf16 numbers is exact in fp32, and
//     a b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + [am bl + al bm + al bl: < 2^-23 |a b|, dropped]
// so SIX v_mfma_f32_32x32x16_bf16 (fp32 accumulate) per 16-deep k-tile and accumulator tile give the fp32 product to fp32
// rounding accuracy -- 6 x 32 = 192 matrix-pipe cycles where the 8 v_mfma_f32_32x32x2_f32 of the kernel above take 512.
// Same loads, same slot / segment machinery and store functors; the split happens once per element when a thread stashes
// its slots (v_cvt_pk_bf16_f32), into three bf16 planes per operand laid out [row][16 k] (32-byte rows: a fragment is one
// ds_read_b128 per plane, contiguous over the wave).  An m-fast operand's thread holds NS CONSECUTIVE k of one row (one 8- or
// 16-byte LDS store per plane), a k-fast operand's thread one k of NS rows (2-byte stores, contiguous over the wave).
// Not bit-identical to the f32 MFMA chain (neither is that to the vector-ALU kernel: summation order); tested to the same
// tolerances.  MEASURED: the wave's issue port, not the matrix pipe, then paces the loop -- per k-tile and wave ~150
// vector instructions (the split is 7 of them per element) + 60 LDS instructions + 24 MFMAs x 8 issue cycles, summed over the two
// waves of a SIMD -- so the 128 x 128 tile gains 10-25 % (586 -> 442 us on the 64 -> 128 channel layer) instead of the 2.6 x of
// the matrix pipe, and the 64 x 64 tile (6 MFMAs per k-tile) LOSES 30-40 %: the option routes only the large tile here.  Variants
// measured and dropped: a truncating split with all nine plane products (exact product; 504 us: more MFMA issue slots), the same
// with eight products, two register sets and the split interleaved between the MFMAs by sched_group_barrier (475 us).
// ----------------------------------------------------------------------------------------------
typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
struct BxSplit { unsigned short h, m, l; };
__device__ __forceinline__ BxSplit bx_split(float v) {
  const __bf16 h = (__bf16)v;                   // v_cvt_pk_bf16_f32: round to nearest even
  const float hf = (float)h;
  // a non-finite h (v = +-inf / NaN, or |v| above the largest bf16) keeps its class: residuals 0, not inf - inf = NaN
  const float r1 = __builtin_fabsf(hf) <= 3.38953139e38f ? v - hf : 0.f;
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  const __bf16 l = (__bf16)r2;
  return {__builtin_bit_cast(unsigned short, h), __builtin_bit_cast(unsigned short, m), __builtin_bit_cast(unsigned short, l)};
}

template <int TM, int TN, bool A_KFAST, bool B_KFAST, class FA, class FB, class SC>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3s_kernel(FA fa, FB fb, SC sc, int M, int N, int K, int lgb, int klen) {
  constexpr int BM = 2 * TM, BN = 2 * TN, IA = TM / 32, IB = TN / 32;
  constexpr int NSA = BM * MB_K / 256, NSB = BN * MB_K / 256;
  constexpr int RP = 256 / MB_K;
  constexpr unsigned TAIL = 0x100u;
  static_assert(MB_K == 16, "one v_mfma_f32_32x32x16_bf16 per k-tile and plane pair");
  __shared__ __attribute__((aligned(16))) unsigned short Ap[2][3][BM][MB_K];
  __shared__ __attribute__((aligned(16))) unsigned short Bp[2][3][BN][MB_K];
  using SA = SOp<FA>;
  using SB = SOp<FB>;
  const int bm = blockIdx.x * BM, bn = blockIdx.y * BN;
  const int zb = blockIdx.z & ((1 << lgb) - 1), zs = blockIdx.z >> lgb;
  const int k_begin = zs * klen;
  const int k_end = min(K, k_begin + klen);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  mb_f32x16 acc[IA][IB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ktail = K & (MB_K - 1);
  SSlot sla[NSA], slb[NSB];
#pragma unroll
  for (int i = 0; i < NSA; ++i) {
    const int mm = A_KFAST ? tid / MB_K + RP * i : (tid & (BM - 1));
    const int ks = A_KFAST ? (tid & (MB_K - 1)) : (tid / BM) * NSA + i;       // m-fast: NSA consecutive k of one row
    const bool ok = bm + mm < M;
    sla[i] = SA::slot(fa, zb, ok ? bm + mm : 0, ok, ks);
    sla[i].m |= ((int)(ktail != 0) & (int)(ks >= ktail)) ? TAIL : 0u;
  }
#pragma unroll
  for (int i = 0; i < NSB; ++i) {
    const int nn = B_KFAST ? tid / MB_K + RP * i : (tid & (BN - 1));
    const int ks = B_KFAST ? (tid & (MB_K - 1)) : (tid / BN) * NSB + i;
    const bool ok = bn + nn < N;
    slb[i] = SB::slot(fb, zb, ok ? bn + nn : 0, ok, ks);
    slb[i].m |= ((int)(ktail != 0) & (int)(ks >= ktail)) ? TAIL : 0u;
  }
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)SA::ptr(fa), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)SB::ptr(fb), 0, 0x7ffffff0, 0x00020000);
  unsigned voa[NSA], vob[NSB];
  float ra[NSA], rb[NSB];
  int ida = -1, idb = -1;
  auto issue = [&](int k0) __attribute__((always_inline)) {
    const bool tail = k0 + MB_K > K;                      // uniform
    SSeg ua, ub;
    SA::seg(fa, zb, k0, ua);
    SB::seg(fb, zb, k0, ub);
    if (tail) { ua.id |= 0x40000000; ua.f |= TAIL; ub.id |= 0x40000000; ub.f |= TAIL; }
    if (ua.id != ida) {
      ida = ua.id;
#pragma unroll
      for (int i = 0; i < NSA; ++i) voa[i] = (unsigned)(sla[i].base + ua.add) | (min(sla[i].m & ua.f, 1u) << 31);
    }
    if (ub.id != idb) {
      idb = ub.id;
#pragma unroll
      for (int i = 0; i < NSB; ++i) vob[i] = (unsigned)(slb[i].base + ub.add) | (min(slb[i].m & ub.f, 1u) << 31);
    }
#pragma unroll
    for (int i = 0; i < NSA; ++i) ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, (int)voa[i], (int)ua.so, 0));
#pragma unroll
    for (int i = 0; i < NSB; ++i) rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsb, (int)vob[i], (int)ub.so, 0));
  };
  auto stash = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NSA; ++i) {
      const BxSplit q = bx_split(ra[i]);
      const int row = A_KFAST ? tid / MB_K + RP * i : (tid & (BM - 1)), k = A_KFAST ? (tid & (MB_K - 1)) : (tid / BM) * NSA + i;
      Ap[st][0][row][k] = q.h; Ap[st][1][row][k] = q.m; Ap[st][2][row][k] = q.l;
    }
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
      const BxSplit q = bx_split(rb[i]);
      const int row = B_KFAST ? tid / MB_K + RP * i : (tid & (BN - 1)), k = B_KFAST ? (tid & (MB_K - 1)) : (tid / BN) * NSB + i;
      Bp[st][0][row][k] = q.h; Bp[st][1][row][k] = q.m; Bp[st][2][row][k] = q.l;
    }
  };
  if (k_begin < k_end) { issue(k_begin); stash(0); }
  __syncthreads();
  int cur = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += MB_K) {
    const bool more = k0 + MB_K < k_end;
    if (more) issue(k0 + MB_K);                    // global loads of the next k-tile fly under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    bx_bf16x8 a[3][IA], b[3][IB];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < IA; ++i) a[p][i] = *reinterpret_cast<const bx_bf16x8*>(&Ap[cur][p][wm * TM + 32 * i + lr][lh * 8]);
#pragma unroll
      for (int j = 0; j < IB; ++j) b[p][j] = *reinterpret_cast<const bx_bf16x8*>(&Bp[cur][p][wn * TN + 32 * j + lr][lh * 8]);
    }
    // smallest terms first; the IA x IB accumulators between two uses of the same one keep the matrix pipe fed
#define BX_PAIR(PA, PB)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < IA; ++i) _Pragma("unroll") for (int j = 0; j < IB; ++j)                          \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][i], b[PB][j], acc[i][j], 0, 0, 0)
    BX_PAIR(1, 1); BX_PAIR(2, 0); BX_PAIR(0, 2); BX_PAIR(1, 0); BX_PAIR(0, 1); BX_PAIR(0, 0);
#undef BX_PAIR
    __builtin_amdgcn_sched_barrier(0);             // the loads stay above the MFMAs, their first use (the stash) below
    if (more) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = bn + wn * TN + j * 32 + lr;
        if (m < M && n < N) sc(zb, zs, m, n, acc[i][j][r]);
      }
}

template <bool AK, bool BK, class FA, class FB, class SC>
static bool launch_structured(const char* name, FA fa, FB fb, SC sc, int M, int N, int K, int nbatch, int klen, dim3 grid,
                              bool small, hipStream_t st) {
  if constexpr (SOp<FA>::OK && SOp<FB>::OK) {
    if (SOp<FA>::ok(fa, M, K) && SOp<FB>::ok(fb, N, K) && lg1(nbatch)) {
      const int lgb = lg1(nbatch) - 1;
      if (!small && rg_option("f32mma", RG_F32MMA_DEFAULT) == 2) {      // each product as six bf16 matrix-core products
        hipLaunchKernelGGL((gemm_bf16x3s_kernel<64, 64, AK, BK, FA, FB, SC>), grid, dim3(256), 0, st, fa, fb, sc, M, N, K, lgb, klen);
        return true;
      }
      if (small) hipLaunchKernelGGL((gemm_mfma32s_kernel<32, 32, AK, BK, FA, FB, SC>), grid, dim3(256), 0, st, fa, fb, sc, M, N, K, lgb, klen);
      else hipLaunchKernelGGL((gemm_mfma32s_kernel<64, 64, AK, BK, FA, FB, SC>), grid, dim3(256), 0, st, fa, fb, sc, M, N, K, lgb, klen);
      return true;
    }
  }
  return false;
}

int pick_split(int tiles, int K) {
  // enough blocks to fill 256 CUs a few times, at least 64 k-steps per split
  int want = (1024 + tiles - 1) / tiles;
  int maxs = K / (GB_K * 16);
  if (maxs < 1) maxs = 1;
  int s = want < maxs ? want : maxs;
  if (s > 256) s = 256;
  return s < 1 ? 1 : s;
}

}  // namespace

// ==============================================================================================
// host entry points for the generic path (called by the dispatchers in rg_api.hip)
// ==============================================================================================
int rg_generic_conv_down(const void* x, const float* w, void* y, int N, int Hi, int Wi, int I, int O, int dtype,
                         hipStream_t st) {
  Geo g = geo_pow2(Geo{N, Hi / 2, Wi / 2, Hi, Wi, O, I});
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<true, true>("conv_down(generic)", DownA<T>{(const T*)x, g}, DownB<T>{w, g},
                                      RowMajorC<T>{(T*)y, O}, N * g.Hl * g.Wl, O, I * 16, 1, 1, st, std::is_same<T, float>::value);
  })
}

int rg_generic_conv_up(const void* x, const float* w, void* y, int N, int Ho, int Wo, int O, int I, const void* mask,
                       float mslope, int dtype, hipStream_t st) {
  Geo g = geo_pow2(Geo{N, Ho, Wo, 2 * Ho, 2 * Wo, O, I});
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<true, false>("conv_up(generic)", UpA<T>{(const T*)x, g}, UpB<T>{w, g},
                                       UpC<T>{(T*)y, g, (const T*)mask, mslope},
                                       N * Ho * Wo, I, O * 4, 4, 1, st, std::is_same<T, float>::value);
  })
}

// fp32 image-side layers (3 input channels): true when the matrix-core kernel with structured operands takes the launch
// (f32mma on, power-of-two image, output rows of >= 16 pixels) -- rg_first_down / rg_skinny_wgrad then come here instead of the
// vector-ALU kernels of rg_skinny.hip (first_down 285 -> see DESIGN 13.2; the 3-column transposed conv stays there: an MFMA
// tile would be 29 / 32 padding).  H, W: the layer's INPUT (high-resolution) size.
bool rg_generic_f32_image_side(int N, int H, int W, int I, int O) {
#ifdef RG_F32_IMAGE_VALU       // (A/B builds)
  return false;
#endif
  return rg_option("f32mma", RG_F32MMA_DEFAULT) != 0 && lg1(H / 2) && lg1(W / 2) && W / 2 >= MB_K && O >= 33 && (long long)N * (H / 2) * (W / 2) >= 64 &&
         (long long)N * I * H * W < (1ll << 29);
}

int rg_generic_first_down(const float* x, const float* w, const float* bias, void* y, int N, int H, int W, int I,
                          int O, float slope, int dtype, hipStream_t st) {
  Geo g = geo_pow2(Geo{N, H / 2, W / 2, H, W, O, I});
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<true, true>("first_down(generic)", FirstDownA{x, g}, DownB<float>{w, g},
                                      BiasActC<T>{(T*)y, bias, slope, O}, N * g.Hl * g.Wl, O, I * 16, 1, 1, st,
                                      std::is_same<T, float>::value);
  })
}

int rg_generic_last_up(const void* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo, int O,
                       int I, int apply_tanh, int dtype, hipStream_t st) {
  Geo g = geo_pow2(Geo{N, Ho, Wo, 2 * Ho, 2 * Wo, O, I});
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<true, false>("last_up(generic)", UpA<T>{(const T*)x, g}, UpBOihw{w, g},
                                       UpCNchw{y, bias, apply_tanh, g}, N * Ho * Wo, I, O * 4, 4, 1, st);
  })
}

// f32 = the launch will take the matrix-core kernel (fp32 storage, f32mma on, >= 64 x 33 outputs): enough splits for three
// 128 x 128 tiles per CU (three workgroups fit a CU) with at least 32 k-tiles each -- the 64 x 64 tile that a short grid would
// fall back to has half the FLOP per LDS byte and per barrier (PMC: MFMA-busy 0.41 against 0.52-0.57 on the weight gradients).
static int generic_wgrad_split(int N, int Ho, int Wo, int O, int I, bool f32) {
  const int K = N * Ho * Wo;
  if (f32 && use_mfma32(true, O, I * 16)) {
    const int tiles = ((O + MB_M - 1) / MB_M) * ((I * 16 + MB_N - 1) / MB_N);
    int want = (768 + tiles - 1) / tiles, maxs = K / (MB_K * 32);
    if (maxs < 1) maxs = 1;
    int s = want < maxs ? want : maxs;
    // (one or two output tiles -- the image-side layer's 64 x 48 gradient over a million pixels: the slabs are tiny, and 256
    // workgroups would be one wave per SIMD)
    const int cap = tiles <= 2 ? 1024 : 256;
    if (s > cap) s = cap;
    return s < 1 ? 1 : s;
  }
  int tiles = ((O + GB_M - 1) / GB_M) * ((I * 16 + GB_N - 1) / GB_N);
  return pick_split(tiles, K);
}

size_t rg_generic_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I) {
  // (the dtype is not known here: room for the larger of the two split plans)
  int s = std::max(generic_wgrad_split(N, Ho, Wo, O, I, false), generic_wgrad_split(N, Ho, Wo, O, I, true));
  return s > 1 ? (size_t)s * O * I * 16 * sizeof(float) : 0;
}

template <class FB>
static int generic_wgrad_impl(const char* name, const void* low, FB fb, float* dw, Geo g, int dtype, int accumulate,
                              void* ws, size_t ws_bytes, hipStream_t st) {
  int K = g.N * g.Hl * g.Wl;
  int s = generic_wgrad_split(g.N, g.Hl, g.Wl, g.O, g.I, dtype == RG_F32);
  size_t elems = (size_t)g.O * g.I * 16;
  if (s > 1) {
    RG_REQUIRE(ws && ws_bytes >= (size_t)s * elems * sizeof(float), RG_EWORKSPACE, "%s: workspace too small", name);
    RG_DISPATCH_DTYPE(dtype, T, {
      int rc = launch_generic<false, false>(name, WgradA<T>{(const T*)low, g}, fb, SlabC{(float*)ws, elems, g.I * 16},
                                            g.O, g.I * 16, K, 1, s, st, std::is_same<T, float>::value);
      if (rc) return rc;
    })
    return rg_reduce_slabs((const float*)ws, dw, elems, s, accumulate, 0, 0, st);
  }
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<false, false>(name, WgradA<T>{(const T*)low, g}, fb, AccumC{dw, g.I * 16, accumulate}, g.O,
                                        g.I * 16, K, 1, 1, st, std::is_same<T, float>::value);
  })
}

int rg_generic_conv_wgrad(const void* low, const void* high, float* dw, int N, int Ho, int Wo, int O, int I,
                          int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  Geo g = geo_pow2(Geo{N, Ho, Wo, 2 * Ho, 2 * Wo, O, I});
  if (dtype == RG_F32)
    return generic_wgrad_impl("conv_wgrad(generic)", low, WgradB<float>{(const float*)high, g}, dw, g, dtype,
                              accumulate, ws, ws_bytes, st);
  return generic_wgrad_impl("conv_wgrad(generic)", low, WgradB<h16_t>{(const h16_t*)high, g}, dw, g, dtype,
                            accumulate, ws, ws_bytes, st);
}

int rg_generic_skinny_wgrad(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                            int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  Geo g = geo_pow2(Geo{N, Ho, Wo, 2 * Ho, 2 * Wo, O, I});
  return generic_wgrad_impl("skinny_wgrad(generic)", low, WgradBNchw{high_nchw, g}, dw, g, dtype, accumulate, ws,
                            ws_bytes, st);
}

int rg_generic_g0_fwd(const float* z, const float* w, void* y, int N, int E, int C, int dtype, hipStream_t st) {
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<true, false>("g0_fwd(generic)", G0A<T>{z, E}, G0Bm<T>{w, C}, G0Cm<T>{(T*)y, C}, N,
                                       16 * C, E, 1, 1, st, std::is_same<T, float>::value);
  })
}

int rg_generic_g0_wgrad(const float* z, const void* gy, float* dw, int N, int E, int C, int dtype, int accumulate,
                        hipStream_t st) {
  RG_DISPATCH_DTYPE(dtype, T, {
    return launch_generic<false, false>("g0_wgrad(generic)", G0WA<T>{z, E}, G0WB<T>{(const T*)gy, C},
                                        AccumC{dw, 16 * C, accumulate}, E, 16 * C, N, 1, 1, st, std::is_same<T, float>::value);
  })
}

// split-K plan of the fp32 linear layers: a batch-64 layer is ONE row tile (the frozen betaVAE encoder, 19198 -> 6000 -> 4000 ->
// 2048 -> 2048: 94 / 63 / 32 / 32 workgroups of 64 x 64 on 256 CUs, 350 us per layer in round 4) -- K is cut so that about two
// workgroups per CU run, every split at least 1024 deep; the partial tiles go to fp32 slabs, one pass sums them in slab order
// and applies the affine + activation epilogue.
int rg_generic_linear_nsplit(int M, int K, int Nout) {
  if (M > 128 || K < 4096) return 1;
  const long long tiles = (long long)((M + 63) / 64) * ((Nout + 63) / 64);
  long long ns = (512 + tiles - 1) / tiles;
  if (ns > K / 1024) ns = K / 1024;
  if (ns > 16) ns = 16;
  return ns < 2 ? 1 : (int)ns;
}
size_t rg_generic_linear_ws_bytes(int M, int K, int Nout) {
  const int ns = rg_generic_linear_nsplit(M, K, Nout);
  return ns > 1 ? (size_t)ns * M * Nout * sizeof(float) : 0;
}
namespace {
__global__ __launch_bounds__(256) void reduce_linear_f32_kernel(const float* __restrict__ slab, float* __restrict__ y, int M,
                                                                int Nout, int ldy, int nsplit, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)M * Nout;
  if (i >= n) return;
  const int m = (int)(i / Nout), j = (int)(i - (size_t)m * Nout);
  float v = 0.f;
  for (int z = 0; z < nsplit; ++z) v += slab[(size_t)z * n + i];
  if (scale) v *= scale[j];
  if (shift) v += shift[j];
  y[(size_t)m * ldy + j] = lrelu_f(v, slope);
}
}  // namespace
int rg_generic_linear(const float* x, int ldx, const float* w, const float* scale, const float* shift, float* y,
                      int ldy, int M, int K, int Nout, float slope, hipStream_t st, void* ws, size_t ws_bytes) {
  const int ns = rg_generic_linear_nsplit(M, K, Nout);
  if (ns > 1 && ws && ws_bytes >= (size_t)ns * M * Nout * sizeof(float)) {
    int rc = launch_generic<true, true>("linear(generic, split-K)", LinA{x, ldx}, LinB{w, K},
                                        SlabC{(float*)ws, (size_t)M * Nout, Nout}, M, Nout, K, 1, ns, st, true);
    if (rc) return rc;
    const size_t n = (size_t)M * Nout;
    hipLaunchKernelGGL(reduce_linear_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, y, M,
                       Nout, ldy, ns, scale, shift, slope);
    RG_LAUNCH_CHECK("linear(generic, split-K reduce)");
    return RG_OK;
  }
  return launch_generic<true, true>("linear(generic)", LinA{x, ldx}, LinB{w, K}, LinC{y, ldy, scale, shift, slope}, M,
                                    Nout, K, 1, 1, st, true);
}

// ================================================================================================
// Resize-convolution block of DCGANUpGenerator (src/dcgan.py:45-56, 76-84):
//   y = Conv2d(Cin, Cout, 3, 1, 0)(ReflectionPad2d(1)(Upsample(x2, bilinear, align_corners=False)(x))) + bias
// The upsampled + padded image is never materialised in the forward and weight-gradient passes: the GEMM's
// operand functor interpolates it on the fly (and rounds it to the activation dtype, the GEMM operand precision).
// ================================================================================================
namespace {
struct UGeo { int N, H, W, Cin, Cout; };     // input H x W, output 2H x 2W

template <typename T>
__device__ __forceinline__ float uppad_at(const T* x, const UGeo& g, int n, int i, int j, int c) {
  int h0, h1, w0, w1;
  float lh, lw;
  up_taps(up_reflect(i, 2 * g.H), g.H, h0, h1, lh);
  up_taps(up_reflect(j, 2 * g.W), g.W, w0, w1, lw);
  const T* xn = x + (size_t)n * g.H * g.W * g.Cin + c;
  float v00 = Elem<T>::ld(xn + ((size_t)h0 * g.W + w0) * g.Cin), v01 = Elem<T>::ld(xn + ((size_t)h0 * g.W + w1) * g.Cin);
  float v10 = Elem<T>::ld(xn + ((size_t)h1 * g.W + w0) * g.Cin), v11 = Elem<T>::ld(xn + ((size_t)h1 * g.W + w1) * g.Cin);
  return (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
}

// forward: M = N*2H*2W, K = Cin*9 (k = ci*9 + kh*3 + kw), Ncols = Cout
template <typename T> struct UpFwdA {
  const T* x; UGeo g;
  __device__ float operator()(int, int m, int k) const {
    int W2 = 2 * g.W, H2 = 2 * g.H;
    int wo = m % W2, t = m / W2, ho = t % H2, n = t / H2;
    int ci = k / 9, r = k - ci * 9, kh = r / 3, kw = r - kh * 3;
    return Elem<T>::round(uppad_at(x, g, n, ho + kh, wo + kw, ci));
  }
};
template <typename T> struct UpFwdB {
  const float* w; UGeo g;
  __device__ float operator()(int, int k, int o) const { return Elem<T>::round(w[(size_t)o * g.Cin * 9 + k]); }
};
struct UpNchwC {   // image-side output: NCHW fp32 + bias
  float* y; const float* bias; UGeo g;
  __device__ void operator()(int, int, int m, int o, float v) const {
    int W2 = 2 * g.W, H2 = 2 * g.H;
    int wo = m % W2, t = m / W2, ho = t % H2, n = t / H2;
    y[(((size_t)n * g.Cout + o) * H2 + ho) * W2 + wo] = v + (bias ? bias[o] : 0.f);
  }
};

// gradient wrt the padded upsampled image: M = N*(2H+2)*(2W+2), K = Cout*9 (k = o*9 + kh*3 + kw), Ncols = Cin
template <typename T, bool NCHW> struct UpBwdA {
  const void* gy; UGeo g;
  __device__ float operator()(int, int m, int k) const {
    int Wp = 2 * g.W + 2, Hp = 2 * g.H + 2;
    int j = m % Wp, t = m / Wp, i = t % Hp, n = t / Hp;
    int o = k / 9, r = k - o * 9, kh = r / 3, kw = r - kh * 3;
    int ho = i - kh, wo = j - kw;
    if (ho < 0 || ho >= 2 * g.H || wo < 0 || wo >= 2 * g.W) return 0.f;
    if (NCHW) return ((const float*)gy)[(((size_t)n * g.Cout + o) * (2 * g.H) + ho) * (2 * g.W) + wo];
    return Elem<T>::ld((const T*)gy + (((size_t)n * (2 * g.H) + ho) * (2 * g.W) + wo) * g.Cout + o);
  }
};
template <typename T> struct UpBwdB {
  const float* w; UGeo g;
  __device__ float operator()(int, int k, int c) const {
    int o = k / 9, r = k - o * 9;
    return Elem<T>::round(w[((size_t)o * g.Cin + c) * 9 + r]);
  }
};
// adjoint of (reflection pad o bilinear x2): gx[n][h][w][c] from gpad[n][2H+2][2W+2][c] (fp32), V channels per thread
template <typename T, int V>
__global__ void uppad_adjoint_kernel(const float* __restrict__ gpad, T* __restrict__ gx, UGeo g) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int CV = g.Cin / V;
  size_t tot = (size_t)g.N * g.H * g.W * CV;
  if (idx >= tot) return;
  int c = (int)(idx % CV) * V;
  size_t t = idx / CV;
  int w = (int)(t % g.W); t /= g.W;
  int h = (int)(t % g.H);
  int n = (int)(t / g.H);
  const int H2 = 2 * g.H, W2 = 2 * g.W, Wp = W2 + 2;
  const float* gp = gpad + (size_t)n * (H2 + 2) * Wp * g.Cin + c;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  for (int du = -1; du <= 2; ++du) {
    int u = 2 * h + du;
    if (u < 0 || u >= H2) continue;
    int a0, a1; float la;
    up_taps(u, g.H, a0, a1, la);
    float ch = (a0 == h ? 1.f - la : 0.f) + (a1 == h ? la : 0.f);
    if (ch == 0.f) continue;
    for (int dv = -1; dv <= 2; ++dv) {
      int v = 2 * w + dv;
      if (v < 0 || v >= W2) continue;
      int b0, b1; float lb;
      up_taps(v, g.W, b0, b1, lb);
      float cw = (b0 == w ? 1.f - lb : 0.f) + (b1 == w ? lb : 0.f);
      if (cw == 0.f) continue;
      // padded rows / columns that read upsampled row u / column v
      float s[V];
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] = 0.f;
      for (int ri = 0; ri < 3; ++ri) {
        int i = ri == 0 ? u + 1 : (ri == 1 ? (u == 1 ? 0 : -1) : (u == H2 - 2 ? H2 + 1 : -1));
        if (i < 0) continue;
        for (int rj = 0; rj < 3; ++rj) {
          int j = rj == 0 ? v + 1 : (rj == 1 ? (v == 1 ? 0 : -1) : (v == W2 - 2 ? W2 + 1 : -1));
          if (j < 0) continue;
          float gv[V];
          Vec<float, V>::ld(gp + ((size_t)i * Wp + j) * g.Cin, gv);
#pragma unroll
          for (int k = 0; k < V; ++k) s[k] += gv[k];
        }
      }
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += ch * cw * s[k];
    }
  }
  Vec<T, V>::st(gx + (size_t)idx * V, acc);
}

// weight gradient: M = Cout, Ncols = Cin*9 (col = c*9 + kh*3 + kw), K = N*2H*2W pixels
template <typename T, bool NCHW> struct UpWgA {
  const void* gy; UGeo g;
  __device__ float operator()(int, int o, int pix) const {
    if (NCHW) {
      int HW = 4 * g.H * g.W, n = pix / HW, r = pix - n * HW;
      return ((const float*)gy)[((size_t)n * g.Cout + o) * HW + r];
    }
    return Elem<T>::ld((const T*)gy + (size_t)pix * g.Cout + o);
  }
};
template <typename T> struct UpWgB {
  const T* x; UGeo g;
  __device__ float operator()(int, int pix, int col) const {
    int W2 = 2 * g.W, H2 = 2 * g.H;
    int wo = pix % W2, t = pix / W2, ho = t % H2, n = t / H2;
    int c = col / 9, r = col - c * 9, kh = r / 3, kw = r - kh * 3;
    return Elem<T>::round(uppad_at(x, g, n, ho + kh, wo + kw, c));
  }
};
int upconv3_split(const UGeo& g) {
  long long K = (long long)g.N * 4 * g.H * g.W;
  long long tiles = (long long)((g.Cout + GB_M - 1) / GB_M) * ((g.Cin * 9 + GB_N - 1) / GB_N);
  long long want = (1024 + tiles - 1) / tiles, maxs = K / 256;
  long long s = want < maxs ? want : maxs;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}
}  // namespace

size_t rg_generic_upconv3_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  UGeo g{N, H, W, Cin, Cout};
  size_t a = (size_t)N * (2 * H + 2) * (2 * W + 2) * Cin * sizeof(float);              // gpad
  size_t b = (size_t)upconv3_split(g) * Cout * Cin * 9 * sizeof(float);                 // wgrad slabs
  return a > b ? a : b;
}

int rg_generic_upconv3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                           int Cout, int out_nchw, int dtype, hipStream_t st) {
  UGeo g{N, H, W, Cin, Cout};
  const int M = N * 4 * H * W;
  RG_DISPATCH_DTYPE(dtype, T, {
    if (out_nchw)
      return launch_generic<true, true>("upconv3_fwd", UpFwdA<T>{(const T*)x, g}, UpFwdB<T>{w, g},
                                        UpNchwC{(float*)y, bias, g}, M, Cout, Cin * 9, 1, 1, st);
    return launch_generic<true, true>("upconv3_fwd", UpFwdA<T>{(const T*)x, g}, UpFwdB<T>{w, g},
                                      BiasActC<T>{(T*)y, bias, 1.0f, Cout}, M, Cout, Cin * 9, 1, 1, st);
  })
}

int rg_generic_upconv3_bwd_data(const void* gy, int gy_nchw, const float* w, void* gx, int N, int H, int W, int Cin,
                                int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t st) {
  UGeo g{N, H, W, Cin, Cout};
  const int Mp = N * (2 * H + 2) * (2 * W + 2);
  RG_REQUIRE(ws && ws_bytes >= (size_t)Mp * Cin * sizeof(float), RG_EWORKSPACE, "upconv3_bwd_data: workspace too small");
  float* gpad = (float*)ws;
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc;
    if (gy_nchw)
      rc = launch_generic<true, false>("upconv3_bwd_data", UpBwdA<T, true>{gy, g}, UpBwdB<T>{w, g},
                                       RowMajorC<float>{gpad, Cin}, Mp, Cin, Cout * 9, 1, 1, st);
    else
      rc = launch_generic<true, false>("upconv3_bwd_data", UpBwdA<T, false>{gy, g}, UpBwdB<T>{w, g},
                                       RowMajorC<float>{gpad, Cin}, Mp, Cin, Cout * 9, 1, 1, st);
    if (rc) return rc;
    if (Cin % 4 == 0) {
      size_t tot = (size_t)N * H * W * (Cin / 4);
      hipLaunchKernelGGL((uppad_adjoint_kernel<T, 4>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, gpad, (T*)gx, g);
    } else {
      size_t tot = (size_t)N * H * W * Cin;
      hipLaunchKernelGGL((uppad_adjoint_kernel<T, 1>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, gpad, (T*)gx, g);
    }
    RG_LAUNCH_CHECK("upconv3_bwd_data");
    return RG_OK;
  })
}

int rg_generic_upconv3_wgrad(const void* gy, int gy_nchw, const void* x, float* dw, int N, int H, int W, int Cin,
                             int Cout, int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  UGeo g{N, H, W, Cin, Cout};
  const int K = N * 4 * H * W, s = upconv3_split(g);
  const size_t elems = (size_t)Cout * Cin * 9;
  RG_REQUIRE(ws && ws_bytes >= (size_t)s * elems * sizeof(float), RG_EWORKSPACE, "upconv3_wgrad: workspace too small");
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc;
    if (gy_nchw)
      rc = launch_generic<false, false>("upconv3_wgrad", UpWgA<T, true>{gy, g}, UpWgB<T>{(const T*)x, g},
                                        SlabC{(float*)ws, elems, Cin * 9}, Cout, Cin * 9, K, 1, s, st);
    else
      rc = launch_generic<false, false>("upconv3_wgrad", UpWgA<T, false>{gy, g}, UpWgB<T>{(const T*)x, g},
                                        SlabC{(float*)ws, elems, Cin * 9}, Cout, Cin * 9, K, 1, s, st);
    if (rc) return rc;
  })
  return rg_reduce_slabs((const float*)ws, dw, elems, s, accumulate, 0, 0, st);
}
