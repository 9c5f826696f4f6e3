// rg_probe.hip -- on-box ceilings for bench.py's roofline object ("re-measure both on the box (MFMA-loop and
// stream-copy microbenchmarks) and report against both nominal and measured peak").  Three measurement kernels, none of them
// on a product path:
//   (a) probe_mfma_bare_kernel : bf16 MFMAs back to back on random operands held in registers (no LDS, no memory): the
//       matrix pipe at the clock the chip holds under that load; one or two waves per SIMD, either MFMA shape, with the
//       accumulator footprint of conv8_kernel's wave tile (128 x 64).
//   (b) conv8_kernel<MODE_PLAIN, 2, 4, MF, 2, 0, PROBE = 1> : the product's 8-wave ping-pong k-loop (rg_conv8.hip, the SAME
//       source, included below) over two LDS-resident stages of a 256 x 256 x 64 tile -- fragment ds_read_b128s, counted waits,
//       barriers, MFMAs -- with the LDS-DMA issue compiled out: what the schedule yields when no operand has to arrive.
//   (c) probe_copy_kernel : float4 stream copy (the guide's 6.29 TB/s figure is this kernel's shape).
// The caller times back-to-back launches with events on its stream (>= 2 s of them first: MI355X_MICROARCH "DVFS give-back").
#define RG_CONV8_KERNEL_ONLY 1
#include "rg_conv8.hip"

namespace {

// (bf16 library only: build.py leaves this file out of librnagan_hip_f16.so)
typedef __attribute__((ext_vector_type(4))) float pf32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }

__device__ __forceinline__ unsigned probe_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// 8 bf16 values uniform in [-1, 1) from a seed (top 7 mantissa bits random, exponent / sign random within the range)
__device__ __forceinline__ bf16x8_t probe_frag(unsigned seed) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned h = probe_hash(seed * 4u + i);
    const float lo = (float)(h & 0xffffu) * (2.f / 65536.f) - 1.f, hi = (float)(h >> 16) * (2.f / 65536.f) - 1.f;
    w[i] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
  }
  u32x4_t v = {w[0], w[1], w[2], w[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// One "k-tile" = the MFMAs conv8_kernel issues per wave and 64-deep k-tile: 128 x 64 x 64 -> 32 of 32x32x16 / 64 of 16x16x32.
// (the launcher asks for 96 KB of dynamic LDS: one workgroup per CU also when the block has 4 waves -- one wave per SIMD)
template <int MF, int NT>
__global__ __launch_bounds__(NT, NT == 512 ? 2 : 1) void probe_mfma_bare_kernel(float* out, int iters) {
  const int t = threadIdx.x;
  const unsigned sd = (blockIdx.x * NT + t) * 64u;
  using acc_t = std::conditional_t<MF == 32, f32x16_t, pf32x4_t>;
  constexpr int NA_T = 128 / MF, NB_T = 64 / MF, NKK = MF == 32 ? 4 : 2, ACC_R = MF == 32 ? 16 : 4;
  acc_t acc[NA_T * NB_T];
#pragma unroll
  for (int i = 0; i < NA_T * NB_T; ++i)
#pragma unroll
    for (int r = 0; r < ACC_R; ++r) acc[i][r] = 0.f;
  // operand fragments of ONE quadrant row / column (what conv8 holds at a time): A 64 rows, B 32 columns, all k-steps
  constexpr int NAQ = 64 / MF, NBQ = 32 / MF > 0 ? 32 / MF : 1;
  bf16x8_t a[2][NAQ * NKK], b[2][NBQ * NKK];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < NAQ * NKK; ++i) a[h][i] = probe_frag(sd + h * 16 + i);
#pragma unroll
    for (int i = 0; i < NBQ * NKK; ++i) b[h][i] = probe_frag(sd + 32 + h * 16 + i);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int qj = 0; qj < 2; ++qj)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
          for (int ta = 0; ta < NAQ; ++ta)
#pragma unroll
            for (int tb = 0; tb < NBQ; ++tb) {
              const int ai = (qi * NAQ + ta) * (NB_T) + qj * NBQ + tb;
              if constexpr (MF == 32)
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[qi][ta * NKK + kk], b[qj][tb * NKK + kk], acc[ai], 0, 0, 0);
              else
                acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qi][ta * NKK + kk], b[qj][tb * NKK + kk], acc[ai], 0, 0, 0);
            }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA_T * NB_T; ++i)
#pragma unroll
    for (int r = 0; r < ACC_R; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;                      // keeps the chain alive, writes (practically) never
}

// a workgroup copies contiguous 32 KB pieces (8 x 16-byte loads in flight per thread, each instruction 4 KB contiguous over the
// workgroup), grid-striding over the pieces; NT: non-temporal loads and stores (the data is touched once)
template <bool NT>
__global__ __launch_bounds__(256) void probe_copy_kernel(const pf32x4_t* __restrict__ src, pf32x4_t* __restrict__ dst, size_t n4) {
  constexpr int U = 8;
  const size_t pieces = n4 / (256 * U);
  for (size_t pc = blockIdx.x; pc < pieces; pc += gridDim.x) {
    const size_t base = pc * (256 * U) + threadIdx.x;
    pf32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + base + u * 256) : src[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + base + u * 256);
      else dst[base + u * 256] = v[u];
    }
  }
  for (size_t i = pieces * (256 * U) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ void probe_fill_bf16_kernel(uint16_t* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const unsigned h = probe_hash((unsigned)i * 2654435761u + seed);
    p[i] = f32_to_bf16((float)(h & 0xffffu) * (2.f / 65536.f) - 1.f);
  }
}

}  // namespace

// (a) one launch = `blocks` workgroups x (waves_per_simd * 4) waves x `iters` k-tiles of 128 x 64 x 64 (2^20 FLOP each)
extern "C" int rg_probe_mfma_bare(int mfma_shape, int waves_per_simd, int blocks, int iters, float* scratch, double* flops_out,
                                  void* stream) {
  RG_REQUIRE((mfma_shape == 16 || mfma_shape == 32) && (waves_per_simd == 1 || waves_per_simd == 2) && blocks > 0 && iters > 0 &&
                 scratch, RG_EINVAL, "probe_mfma_bare: bad args");
  hipStream_t st = rg_stream(stream);
  const int nt = waves_per_simd * 256;
  constexpr unsigned PAD = 96 * 1024;        // dynamic LDS nobody touches: a second workgroup does not fit beside the first
  if (mfma_shape == 32 && nt == 256) hipLaunchKernelGGL((probe_mfma_bare_kernel<32, 256>), dim3(blocks), dim3(256), PAD, st, scratch, iters);
  else if (mfma_shape == 32) hipLaunchKernelGGL((probe_mfma_bare_kernel<32, 512>), dim3(blocks), dim3(512), PAD, st, scratch, iters);
  else if (nt == 256) hipLaunchKernelGGL((probe_mfma_bare_kernel<16, 256>), dim3(blocks), dim3(256), PAD, st, scratch, iters);
  else hipLaunchKernelGGL((probe_mfma_bare_kernel<16, 512>), dim3(blocks), dim3(512), PAD, st, scratch, iters);
  RG_LAUNCH_CHECK("probe_mfma_bare");
  if (flops_out) *flops_out = (double)blocks * (nt / 64) * (double)iters * 2.0 * 128 * 64 * 64;
  return RG_OK;
}

// (b) a: [blocks * 256][128] bf16, b: [256][128] bf16 (random, rg_probe_fill_bf16), c: [blocks * 256][256] bf16.
extern "C" int rg_probe_lds_mfma(int mfma_shape, int blocks, int iters, const void* a, const void* b, void* c, double* flops_out,
                                 void* stream) {
  RG_REQUIRE((mfma_shape == 16 || mfma_shape == 32) && blocks > 0 && iters > 0 && iters % 2 == 0 && a && b && c, RG_EINVAL,
             "probe_lds_mfma: bad args");
  G2Args a2{};
  GArgs& g = a2.g;
  g.A = (const uint16_t*)a; g.B = (const uint16_t*)b; g.C = c;
  g.M = blocks * 256; g.Ncols = 256; g.Cin = 128; g.taps = 1;
  g.lgW = 0; g.lgH = 0; g.Hs = 1; g.Ws = 1; g.ldc = 256; g.b_col = 128; g.b_tap = 0; g.tiles_n = 1;
  a2.a_bytes = (unsigned)((size_t)g.M * 128 * 2); a2.b_bytes = 256 * 128 * 2;
  a2.nsplit = 1; a2.tiles_m = blocks; a2.lgcpt = 30; a2.cmask = 0x3fffffff; a2.probe_iters = iters;
  hipStream_t st = rg_stream(stream);
  if (mfma_shape == 16) hipLaunchKernelGGL((conv8_kernel<MODE_PLAIN, 2, 4, 16, 2, 0, 1>), dim3(blocks), dim3(512), 0, st, a2);
  else hipLaunchKernelGGL((conv8_kernel<MODE_PLAIN, 2, 4, 32, 2, 0, 1>), dim3(blocks), dim3(512), 0, st, a2);
  RG_LAUNCH_CHECK("probe_lds_mfma");
  if (flops_out) *flops_out = (double)blocks * (double)iters * 2.0 * 256 * 256 * 64;
  return RG_OK;
}

// (c) n bytes (multiple of 16) src -> dst; variant: 0 plain / 1 non-temporal loads and stores; blocks: grid size (0 = 2048)
extern "C" int rg_probe_copy(const void* src, void* dst, size_t nbytes, int variant, int blocks, void* stream) {
  RG_REQUIRE(src && dst && nbytes % 16 == 0 && nbytes > 0 && blocks >= 0, RG_EINVAL, "probe_copy: bad args");
  const dim3 grid(blocks ? blocks : 2048);
  if (variant == 1)
    hipLaunchKernelGGL(probe_copy_kernel<true>, grid, dim3(256), 0, rg_stream(stream), (const pf32x4_t*)src, (pf32x4_t*)dst, nbytes / 16);
  else
    hipLaunchKernelGGL(probe_copy_kernel<false>, grid, dim3(256), 0, rg_stream(stream), (const pf32x4_t*)src, (pf32x4_t*)dst, nbytes / 16);
  RG_LAUNCH_CHECK("probe_copy");
  return RG_OK;
}

extern "C" int rg_probe_fill_bf16(void* p, size_t n, unsigned seed, void* stream) {
  RG_REQUIRE(p && n > 0, RG_EINVAL, "probe_fill_bf16: bad args");
  hipLaunchKernelGGL(probe_fill_bf16_kernel, dim3(1024), dim3(256), 0, rg_stream(stream), (uint16_t*)p, n, seed);
  RG_LAUNCH_CHECK("probe_fill_bf16");
  return RG_OK;
}
