// rg_wgrad8f.hip -- the fp32 mode's weight gradients from operands split once per tensor into bf16 planes (rg_conv8f.hip has the
// scheme): wgrad8_kernel (rg_wgrad8.hip, the SAME source, template parameter NP) over K-concatenated plane pairs of `low` and
// `high`, fp32 accumulation, fp32 result (split-K slabs fp32, reduced in fixed order).  (bf16 library only.)
#define RG_WGRAD8_KERNEL_ONLY 1
#include "rg_wgrad8.hip"

namespace {

struct WPlan { int nsplit, per; bool narrow; };      // narrow: the 128 x 512 tile (O = 128: wgrad8n_kernel)

bool wplan(int N, int Ho, int Wo, int O, int I, int products, WPlan* pl) {
  const long long K = (long long)N * Ho * Wo;
  if (!(products == 3 || products == 6) || N <= 0 || !rg_is_pow2(Ho) || !rg_is_pow2(Wo)) return false;
  if (!(I % 8 == 0 && K >= 256 && K % 64 == 0)) return false;
  if (O % 256 == 0 && (16 * I) % 256 == 0) pl->narrow = false;
  else if (O % 128 == 0 && (16 * I) % 512 == 0) pl->narrow = true;
  else return false;
  if (3ull * K * O * 2 >= 0x7fffff00ull || 3ull * K * 4 * I * 2 >= 0x7fffff00ull) return false;
  return true;
}
// split-K over the flat (plane pair, pixel tile) index: one block per CU where the tiles allow it
void wsplit(long long Kflat_tiles, int O, int I, WPlan* pl) {
  const int tiles = pl->narrow ? (O / 128) * (16 * I / 512) : (O / 256) * (16 * I / 256);
  const int target = rg_option("wgrad8_blocks", 256);
  int ns = (target + tiles - 1) / tiles;
  if (ns < 1) ns = 1;
  long long per = (Kflat_tiles + ns - 1) / ns;
  if (per < 4) per = 4;
  per = (per + 1) & ~1ll;
  pl->per = (int)per;
  pl->nsplit = (int)((Kflat_tiles + per - 1) / per);
}

}  // namespace

extern "C" int rg_f32p_wgrad_supported(int N, int Ho, int Wo, int O, int I, int products) {
  WPlan pl;
  return wplan(N, Ho, Wo, O, I, products, &pl) ? 1 : 0;
}
extern "C" size_t rg_f32p_wgrad_workspace_bytes(int N, int Ho, int Wo, int O, int I, int products, int two) {
  WPlan pl;
  if (!wplan(N, Ho, Wo, O, I, products, &pl)) return 0;
  const long long K = (long long)N * Ho * Wo;
  wsplit((K / 64) * (two ? 2 : 1) * products, O, I, &pl);
  return pl.nsplit > 1 ? (size_t)pl.nsplit * O * 16 * I * sizeof(float) : 0;
}

// dw[O][16][I] (+)= wgrad(low0, high0) (+ wgrad(low1, high1)); low*: planes [3][N][Ho][Wo][O], high*: planes [3][N][2Ho][2Wo][I]
extern "C" int rg_f32p_wgrad(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N, int Ho,
                             int Wo, int O, int I, int products, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  WPlan pl;
  RG_REQUIRE(low0 && high0 && dw && (low1 == nullptr) == (high1 == nullptr), RG_EINVAL, "f32p_wgrad: bad args");
  RG_REQUIRE(wplan(N, Ho, Wo, O, I, products, &pl), RG_EUNSUPPORTED, "f32p_wgrad: shape has no planes kernel");
  const bool two = low1 != nullptr;
  const int Kseg = N * Ho * Wo;
  wsplit((long long)(Kseg / 64) * (two ? 2 : 1) * products, O, I, &pl);
  const size_t need = pl.nsplit > 1 ? (size_t)pl.nsplit * O * 16 * I * sizeof(float) : 0;
  RG_REQUIRE(need == 0 || (ws && ws_bytes >= need), RG_EWORKSPACE, "f32p_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
  hipStream_t st = rg_stream(stream);
  W8Args g{};
  g.low[0] = (const uint16_t*)low0; g.high[0] = (const uint16_t*)high0;
  g.low[1] = (const uint16_t*)(two ? low1 : low0); g.high[1] = (const uint16_t*)(two ? high1 : high0);
  g.low_plane = (unsigned)((size_t)Kseg * O * 2); g.high_plane = (unsigned)((size_t)Kseg * 4 * I * 2);
  g.low_bytes = 3 * g.low_plane; g.high_bytes = 3 * g.high_plane;
  g.Kseg[0] = Kseg; g.Kseg[1] = two ? Kseg : 0;
  g.out = pl.nsplit > 1 ? (float*)ws : dw; g.O = O; g.I = I;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_o = pl.narrow ? O / 128 : O / 256; g.tiles_c = pl.narrow ? 16 * I / 512 : 16 * I / 256;
  g.nsplit = pl.nsplit; g.kt_per_split = pl.per;
  g.accumulate = pl.nsplit == 1 ? accumulate : 0;
  const dim3 grid((unsigned)(g.tiles_o * g.tiles_c * pl.nsplit));
  if (pl.narrow) {
    if (products == 6) hipLaunchKernelGGL((wgrad8n_kernel<6>), grid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL((wgrad8n_kernel<3>), grid, dim3(512), 0, st, g);
  } else {
    if (products == 6) hipLaunchKernelGGL((wgrad8_kernel<false, 6>), grid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL((wgrad8_kernel<false, 3>), grid, dim3(512), 0, st, g);
  }
  RG_LAUNCH_CHECK("f32p_wgrad");
  if (pl.nsplit > 1) return rg_reduce_slabs((const float*)ws, dw, (size_t)O * 16 * I, pl.nsplit, accumulate, 0, 0, st);
  return RG_OK;
}
