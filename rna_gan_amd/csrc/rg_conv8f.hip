// rg_conv8f.hip -- the fp32 mode's convolutions on the bf16 matrix cores from operands split ONCE PER TENSOR.
//
// An fp32 number is the exact sum of three bf16 numbers v = h + m + l.  rg_split_planes writes the three planes of a tensor
// (plane-major [3][n] bf16: 6 bytes per element instead of 4) in one bandwidth-bound pass; the convolutions then run the
// product's 8-wave bf16 kernel (rg_conv8.hip, the SAME source, template parameter NP) over K-CONCATENATED planes -- NP plane
// pairs per k-tile, fp32 accumulation in the MFMA chain, fp32 result:
//   products = 6: hh hm mh hl lh mm   (every term down to 2^-24 |a b|: the accuracy class of the f32 instruction)
//   products = 3: hh hm mh            (2^-16 |a b| per product)
// Before this, the fp32 mode split every operand element per TILE inside its GEMM kernel (gemm_bf16x3s_kernel, rg_generic.hip:
// 7 vector instructions per element, issue-port bound at 0.33 of what six bf16 products allow).  (bf16 library only.)
#define RG_CONV8_KERNEL_ONLY 1
#include "rg_conv8.hip"

namespace {

// v -> (h, m, l) with exact residuals.  A non-finite v keeps its class: h carries it (inf / NaN), the residual planes are
// zero -- v - h would be NaN for an infinity; magnitudes that round to an infinite h (> 0x7f7f bf16) likewise.
struct Planes3 { uint16_t h, m, l; };
__device__ __forceinline__ Planes3 split3(float v) {
  const __bf16 h = (__bf16)v;
  const float hf = (float)h;
  const bool fin = __builtin_fabsf(hf) <= 3.38953139e38f;          // largest finite bf16
  const float r1 = fin ? v - hf : 0.f;
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  const __bf16 l = (__bf16)r2;
  return {__builtin_bit_cast(uint16_t, h), __builtin_bit_cast(uint16_t, m), __builtin_bit_cast(uint16_t, l)};
}

// 8 elements per thread: two 16-byte loads, three 16-byte stores (one per plane)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
  const size_t n8 = n >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t oh[4], om[4], ol[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Planes3 p0 = split3(v[2 * k]), p1 = split3(v[2 * k + 1]);
      oh[k] = (uint32_t)p0.h | ((uint32_t)p1.h << 16);
      om[k] = (uint32_t)p0.m | ((uint32_t)p1.m << 16);
      ol[k] = (uint32_t)p0.l | ((uint32_t)p1.l << 16);
    }
    reinterpret_cast<uint4*>(dst)[i] = make_uint4(oh[0], oh[1], oh[2], oh[3]);
    reinterpret_cast<uint4*>(dst + n)[i] = make_uint4(om[0], om[1], om[2], om[3]);
    reinterpret_cast<uint4*>(dst + 2 * n)[i] = make_uint4(ol[0], ol[1], ol[2], ol[3]);
  }
  // tail (n % 8 elements)
  const size_t t = (n8 << 3) + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    const Planes3 p = split3(src[t]);
    dst[t] = p.h; dst[n + t] = p.m; dst[2 * n + t] = p.l;
  }
}

struct FPlan { int bm, bn, nsplit, nkt; };

// tile and split-K plan of a planes launch (the bf16 plan's rules, rg_mfma.hip gather_plan, on NP times as many k-tiles)
bool fplan(int up, int N, int Hl, int Wl, int O, int I, int products, FPlan* pl) {
  const int M = N * Hl * Wl, Ncols = up ? I : O, Cin = up ? O : I, taps = up ? 4 : 16, nclass = up ? 4 : 1;
  if (!(products == 3 || products == 6) || N <= 0 || !rg_is_pow2(Hl) || !rg_is_pow2(Wl) || Cin % 64 || !rg_is_pow2(Cin >> 6)) return false;
  int bm = 0, bn = 0;
  if (Ncols % 256 == 0 && M >= 256) { bm = 256; bn = 256; }
  else if (Ncols % 128 == 0 && M >= 512) { bm = 512; bn = 128; }
  if (!bm) return false;
  const size_t a_plane = (size_t)N * Hl * Wl * (up ? O : 4 * I) * 2, b_plane = (size_t)O * 16 * I * 2;
  if (3 * a_plane >= 0x7fffff00ull || 3 * b_plane >= 0x7fffff00ull) return false;
  const int nkt = taps * (Cin >> 6) * products;
  if (nkt < 4 || (nkt & 1)) return false;
  const long long tiles = (long long)((M + bm - 1) / bm) * (Ncols / bn) * nclass;
  int ns = 1;
  while (tiles * ns < 256 && ns < 8 && nkt % (ns * 4) == 0 && nkt / (ns * 2) >= 8) ns *= 2;
  pl->bm = bm; pl->bn = bn; pl->nsplit = ns; pl->nkt = nkt;
  return true;
}

template <int MODE, int NP>
void launch_planes(const G2Args& a2, int bm, dim3 grid, hipStream_t st) {
  if (bm == 256) hipLaunchKernelGGL((conv8_kernel<MODE, 2, 4, 16, 2, 0, 0, NP>), grid, dim3(512), 0, st, a2);
  else hipLaunchKernelGGL((conv8_kernel<MODE, 4, 2, 16, 2, 0, 0, NP>), grid, dim3(512), 0, st, a2);
}

}  // namespace

extern "C" int rg_split_planes(const float* src, void* planes_bf16, size_t n, void* stream) {
  RG_REQUIRE(src && planes_bf16 && n > 0 && n % 8 == 0 && (((uintptr_t)src | (uintptr_t)planes_bf16) & 15) == 0, RG_EINVAL,
             "split_planes: n must be a multiple of 8 and the buffers 16-byte aligned");
  size_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, rg_stream(stream), src, (uint16_t*)planes_bf16, n);
  RG_LAUNCH_CHECK("split_planes");
  return RG_OK;
}

extern "C" int rg_f32p_conv_supported(int up, int N, int Hlow, int Wlow, int O, int I, int products) {
  FPlan pl;
  if (fplan(up, N, Hlow, Wlow, O, I, products, &pl)) return 1;
  return up && rg_mfma_conv_up_planes64_supported(N, Hlow, Wlow, O, I, products) ? 1 : 0;     // 64 output channels: rg_mfma.hip
}
extern "C" size_t rg_f32p_conv_workspace_bytes(int up, int N, int Hlow, int Wlow, int O, int I, int products) {
  FPlan pl;
  if (!fplan(up, N, Hlow, Wlow, O, I, products, &pl) || pl.nsplit == 1) return 0;
  const size_t rows_out = (size_t)N * Hlow * Wlow * (up ? 4 : 1);
  return (size_t)pl.nsplit * rows_out * (up ? I : O) * sizeof(float);
}
// partial rows of BatchNorm column sums the launch writes (0: a split launch -- the BatchNorm op reduces its input itself)
extern "C" int rg_f32p_conv_stats_rows(int up, int N, int Hlow, int Wlow, int O, int I, int products) {
  FPlan pl;
  if (!fplan(up, N, Hlow, Wlow, O, I, products, &pl) || pl.nsplit > 1) return 0;
  const int M = N * Hlow * Wlow;
  return (up ? 4 : 1) * ((M + pl.bm - 1) / pl.bm) * (pl.bm / 128);
}

// 1 when rg_f32p_conv applies mask_f32 / mask_slope itself for this shape (the 64-column transposed conv); 0: the caller runs
// rg_lrelu_bwd behind the conv
extern "C" int rg_f32p_conv_mask_supported(int up, int N, int Hlow, int Wlow, int O, int I, int products) {
  FPlan pl;
  return up && !fplan(up, N, Hlow, Wlow, O, I, products, &pl) && rg_mfma_conv_up_planes64_supported(N, Hlow, Wlow, O, I, products);
}

extern "C" int rg_f32p_conv(int up, const void* x_planes, const void* w_planes, float* y, int N, int Hlow, int Wlow, int O, int I,
                            int products, float* stats_partial, const float* mask_f32, float mask_slope, void* ws, size_t ws_bytes,
                            void* stream) {
  FPlan pl;
  RG_REQUIRE(x_planes && w_planes && y, RG_EINVAL, "f32p_conv: null");
  if (!fplan(up, N, Hlow, Wlow, O, I, products, &pl) && up && rg_mfma_conv_up_planes64_supported(N, Hlow, Wlow, O, I, products))
    return rg_mfma_conv_up_planes64(x_planes, w_planes, y, N, Hlow, Wlow, O, I, products, rg_stream(stream), mask_f32, mask_slope);
  RG_REQUIRE(!mask_f32, RG_EUNSUPPORTED, "f32p_conv: a fused mask only with the 64-column transposed conv (rg_f32p_conv_mask_supported)");
  RG_REQUIRE(fplan(up, N, Hlow, Wlow, O, I, products, &pl), RG_EUNSUPPORTED, "f32p_conv: shape has no planes kernel");
  hipStream_t st = rg_stream(stream);
  G2Args a2{};
  GArgs& g = a2.g;
  g.A = (const uint16_t*)x_planes; g.B = (const uint16_t*)w_planes; g.C = y;
  g.M = N * Hlow * Wlow;
  size_t a_plane, rows_out;
  if (up) {          // x [N][Ho][Wo][O] low-res input, w planes wup[16][I][O]
    g.Ncols = I; g.Cin = O; g.taps = 4; g.Hs = Hlow; g.Ws = Wlow; g.ldc = I; g.b_col = O; g.b_tap = I * O;
    a_plane = (size_t)g.M * O * 2; rows_out = (size_t)g.M * 4;
  } else {           // x [N][2 Hlow][2 Wlow][I], w planes wdn[O][16][I]
    g.Ncols = O; g.Cin = I; g.taps = 16; g.Hs = 2 * Hlow; g.Ws = 2 * Wlow; g.ldc = O; g.b_col = 16 * I; g.b_tap = I;
    a_plane = (size_t)g.M * 4 * I * 2; rows_out = (size_t)g.M;
  }
  g.lgW = rg_ilog2(Wlow); g.lgH = rg_ilog2(Hlow);
  const size_t b_plane = (size_t)O * 16 * I * 2;
  a2.a_plane = (unsigned)a_plane; a2.b_plane = (unsigned)b_plane;
  a2.a_bytes = (unsigned)(3 * a_plane); a2.b_bytes = (unsigned)(3 * b_plane);
  a2.korder = rg_option("korder", 1);
  a2.nsplit = pl.nsplit;
  const size_t need = pl.nsplit > 1 ? (size_t)pl.nsplit * rows_out * g.Ncols * sizeof(float) : 0;
  RG_REQUIRE(need == 0 || (ws && ws_bytes >= need), RG_EWORKSPACE, "f32p_conv: workspace too small (%zu < %zu)", ws_bytes, need);
  a2.slab = (float*)ws; a2.slab_stride = (long long)rows_out * g.Ncols;
  g.stats = pl.nsplit == 1 ? stats_partial : nullptr;
  g.tiles_n = g.Ncols / pl.bn;
  a2.tiles_m = (g.M + pl.bm - 1) / pl.bm;
  a2.lgcpt = rg_ilog2(g.Cin >> 6); a2.cmask = (g.Cin >> 6) - 1;
  const int nclass = up ? 4 : 1;
  dim3 grid((unsigned)(a2.tiles_m * g.tiles_n), (unsigned)nclass, (unsigned)pl.nsplit);
  a2.xcd_swizzle = (rg_option("xcd", 1) && grid.x % 8 == 0 && grid.x >= 16 && a_plane > b_plane) ? 1 : 0;
  if (up) { if (products == 6) launch_planes<MODE_UP, 6>(a2, pl.bm, grid, st); else launch_planes<MODE_UP, 3>(a2, pl.bm, grid, st); }
  else { if (products == 6) launch_planes<MODE_DOWN, 6>(a2, pl.bm, grid, st); else launch_planes<MODE_DOWN, 3>(a2, pl.bm, grid, st); }
  RG_LAUNCH_CHECK("f32p_conv");
  if (pl.nsplit > 1) return rg_reduce_slabs((const float*)ws, y, rows_out * g.Ncols, pl.nsplit, 0, 0, 0, st);
  return RG_OK;
}
