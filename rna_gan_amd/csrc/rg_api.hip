// rg_api.hip -- C-ABI dispatchers: argument validation, algorithm choice (MFMA vs generic),
// error string.  See include/rnagan_hip.h for the contract.
#include "rg_common.h"
#include "rg_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

static thread_local char g_err[512] = "";

void rg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// bumped whenever an entry point is added or a signature changes; rna_gan_amd/_abi.py (ABI_VERSION) refuses any other value
extern "C" int rg_version(void) { return 600; }   // 6.00: round 6 (rg_probe_*: on-box ceilings)
extern "C" const char* rg_last_error(void) { return g_err; }

// ---- kernel-selection knobs: override table in front of the RNAGAN_* environment variables
static const char* const g_opt_names[] = {"conv8", "conv8_blocks", "conv_tile", "xcd", "class_fast", "wgrad_blocks",
                                          "wgrad8", "conv_v1", "stream_tile", "narrow8", "conv8_mfma", "conv8_epi", "wgrad8_blocks", "korder", "convp", "convp_blocks",
                                          "fp8_mx", "wgrad8n", "f32mma", "convd", "convd_blocks", "skinny128", "slab16", "wslab16", "bn_rev", "wgrad8_mfma", "narrow32", "upimg", "upimg_blocks"};
constexpr int G_NOPT = sizeof(g_opt_names) / sizeof(g_opt_names[0]);
static int g_opt_override[G_NOPT];      // value + 1; 0 = not set
static int g_opt_env[G_NOPT];           // cached environment value + 1; 0 = not read yet; -1 = variable absent

int rg_option(const char* name, int dflt) {
  for (int i = 0; i < G_NOPT; ++i) {
    if (strcmp(name, g_opt_names[i]) != 0) continue;
    if (g_opt_override[i] > 0) return g_opt_override[i] - 1;
    if (g_opt_env[i] == 0) {
      char var[64] = "RNAGAN_";
      size_t n = strlen(var);
      for (const char* c = name; *c && n + 1 < sizeof(var); ++c) var[n++] = (char)toupper((unsigned char)*c);
      var[n] = 0;
      const char* e = getenv(var);
      g_opt_env[i] = e ? atoi(e) + 1 : -1;
      if (g_opt_env[i] == 0) g_opt_env[i] = -1;      // negative values are not representable: treated as absent
    }
    return g_opt_env[i] > 0 ? g_opt_env[i] - 1 : dflt;
  }
  return dflt;
}

extern "C" int rg_set_option(const char* name, int value) {
  RG_REQUIRE(name, RG_EINVAL, "set_option: null name");
  for (int i = 0; i < G_NOPT; ++i)
    if (strcmp(name, g_opt_names[i]) == 0) { g_opt_override[i] = value < 0 ? 0 : value + 1; return RG_OK; }
  rg_set_error("set_option: unknown option '%s'", name);
  return RG_EINVAL;
}

static bool want_mfma(int algo, int dtype) { return algo != RG_ALGO_GENERIC && dtype == RG_H16; }

// ------------------------------------------------------------------------------------------------
extern "C" int rg_pack_conv_weight(const float* w, void* wdn, void* wup, int O, int I, int dtype, void* stream) {
  RG_REQUIRE(w && O > 0 && I > 0, RG_EINVAL, "pack_conv_weight: bad args");
  RG_REQUIRE(dtype == RG_H16, RG_EUNSUPPORTED, "pack_conv_weight: only bf16 packs exist (fp32 kernels read w)");
  return rg_mfma_pack_conv_weight(w, wdn, wup, O, I, rg_stream(stream));
}

extern "C" size_t rg_conv_workspace_bytes(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo) {
  if (!want_mfma(algo, dtype)) return 0;
  if (!rg_mfma_conv_supported(N, Hlow, Wlow, up ? O : I, up ? I : O)) return 0;
  return rg_mfma_conv_ws_bytes(up, N, Hlow, Wlow, O, I);
}

extern "C" int rg_conv_stats_rows(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo) {
  if (N <= 0 || Hlow <= 0 || Wlow <= 0 || O <= 0 || I <= 0 || !want_mfma(algo, dtype)) return 0;
  if (!rg_mfma_conv_supported(N, Hlow, Wlow, up ? O : I, up ? I : O)) return 0;
  return rg_mfma_conv_stats_rows(up, N, Hlow, Wlow, O, I);
}

// ---- split-K launches whose slab reduction is left to the consumer (rg_bn_forward_slabs / rg_bn_act_bwd_slabs)
extern "C" int rg_conv_split(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo) {
  if (N <= 0 || Hlow <= 0 || Wlow <= 0 || O <= 0 || I <= 0 || !want_mfma(algo, dtype)) return 1;
  if (!rg_mfma_conv_supported(N, Hlow, Wlow, up ? O : I, up ? I : O)) return 1;
  return rg_mfma_conv_nsplit(up, N, Hlow, Wlow, O, I);
}

extern "C" int rg_conv_slab_dtype(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo) {
  if (rg_conv_split(up, N, Hlow, Wlow, O, I, dtype, algo) <= 1) return RG_F32;
  return rg_mfma_conv_slab16(up, N, Hlow, Wlow, O, I) ? RG_H16 : RG_F32;
}

extern "C" int rg_conv_down_partial(const void* x, const void* wdn, int N, int Hi, int Wi, int I, int O, int dtype, int algo,
                                    void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && wdn && ws && N > 0 && Hi > 0 && Wi > 0 && I > 0 && O > 0 && Hi % 2 == 0 && Wi % 2 == 0, RG_EINVAL,
             "conv_down_partial: bad args");
  RG_REQUIRE(rg_conv_split(0, N, Hi / 2, Wi / 2, O, I, dtype, algo) > 1, RG_EUNSUPPORTED,
             "conv_down_partial: this shape does not run split-K (rg_conv_split)");
  return rg_mfma_conv_down(x, wdn, nullptr, N, Hi, Wi, I, O, nullptr, ws, ws_bytes, rg_stream(stream), 1);
}

extern "C" int rg_conv_up_partial(const void* x, const void* wup, int N, int Ho, int Wo, int O, int I, int dtype, int algo,
                                  void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && wup && ws && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL, "conv_up_partial: bad args");
  RG_REQUIRE(rg_conv_split(1, N, Ho, Wo, O, I, dtype, algo) > 1, RG_EUNSUPPORTED,
             "conv_up_partial: this shape does not run split-K (rg_conv_split)");
  return rg_mfma_conv_up(x, wup, nullptr, N, Ho, Wo, O, I, nullptr, 1.f, nullptr, ws, ws_bytes, rg_stream(stream), nullptr,
                         nullptr, 1.f, 0, 1);
}

// ---- data-gradient convs that also produce the BatchNorm-backward sums of the block they feed (rg_bn_act_bwd_partials)
extern "C" int rg_conv_bnbwd_rows(int up, int N, int Hlow, int Wlow, int O, int I, int groups, int dtype, int algo) {
  if (N <= 0 || Hlow <= 0 || Wlow <= 0 || O <= 0 || I <= 0 || !want_mfma(algo, dtype)) return 0;
  if (!rg_mfma_conv_supported(N, Hlow, Wlow, up ? O : I, up ? I : O)) return 0;
  return rg_mfma_conv_bnbwd_rows(up, N, Hlow, Wlow, O, I, groups);
}

extern "C" int rg_conv_down_bnbwd(const void* x, const void* wdn, void* y, int N, int Hi, int Wi, int I, int O, const void* z_next,
                                  const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                                  int groups, float* sums_partial, int dtype, int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && wdn && y && z_next && mean && invstd && gamma && beta && sums_partial && N > 0 && Hi > 0 && Wi > 0 && I > 0 &&
                 O > 0 && Hi % 2 == 0 && Wi % 2 == 0, RG_EINVAL, "conv_down_bnbwd: bad args");
  RG_REQUIRE(rg_conv_bnbwd_rows(0, N, Hi / 2, Wi / 2, O, I, groups, dtype, algo) > 0, RG_EUNSUPPORTED,
             "conv_down_bnbwd: this shape has no fused form (rg_conv_bnbwd_rows)");
  RgBnBwdFuse bf{z_next, mean, invstd, gamma, beta, slope, groups, sums_partial};
  return rg_mfma_conv_down(x, wdn, y, N, Hi, Wi, I, O, nullptr, ws, ws_bytes, rg_stream(stream), 0, &bf);
}

extern "C" int rg_conv_up_bnbwd(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I, const void* z_next,
                                const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                                int groups, float* sums_partial, int dtype, int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && wup && y && z_next && mean && invstd && gamma && beta && sums_partial && N > 0 && Ho > 0 && Wo > 0 && I > 0 &&
                 O > 0, RG_EINVAL, "conv_up_bnbwd: bad args");
  RG_REQUIRE(rg_conv_bnbwd_rows(1, N, Ho, Wo, O, I, groups, dtype, algo) > 0, RG_EUNSUPPORTED,
             "conv_up_bnbwd: this shape has no fused form (rg_conv_bnbwd_rows)");
  RgBnBwdFuse bf{z_next, mean, invstd, gamma, beta, slope, groups, sums_partial};
  return rg_mfma_conv_up(x, wup, y, N, Ho, Wo, O, I, nullptr, 1.f, nullptr, ws, ws_bytes, rg_stream(stream), nullptr, nullptr,
                         1.f, 0, 0, &bf);
}

extern "C" int rg_conv_down(const void* x, const float* w, const void* wdn, void* y, int N, int Hi, int Wi, int I,
                            int O, float* stats_partial, int dtype, int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && I > 0 && O > 0 && Hi % 2 == 0 && Wi % 2 == 0, RG_EINVAL,
             "conv_down: bad args");
  if (want_mfma(algo, dtype) && wdn && rg_mfma_conv_supported(N, Hi / 2, Wi / 2, /*Kc=*/I, /*Ncols=*/O))
    return rg_mfma_conv_down(x, wdn, y, N, Hi, Wi, I, O, stats_partial, ws, ws_bytes, rg_stream(stream));
  RG_REQUIRE(!stats_partial, RG_EUNSUPPORTED, "conv_down: statistics come from the MFMA epilogue only");
  RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "conv_down: shape/dtype not supported by the MFMA kernel");
  RG_REQUIRE(w, RG_EINVAL, "conv_down: generic kernel needs the fp32 master weight");
  return rg_generic_conv_down(x, w, y, N, Hi, Wi, I, O, dtype, rg_stream(stream));
}

extern "C" int rg_conv_up(const void* x, const float* w, const void* wup, void* y, int N, int Ho, int Wo, int O, int I,
                          const void* mask_act, float mask_slope, float* stats_partial, int dtype, int algo, void* ws,
                          size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && y && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL, "conv_up: bad args");
  RG_REQUIRE(!(mask_act && stats_partial), RG_EINVAL, "conv_up: mask and statistics are exclusive");
  if (want_mfma(algo, dtype) && wup && rg_mfma_conv_supported(N, Ho, Wo, /*Kc=*/O, /*Ncols=*/I))
    return rg_mfma_conv_up(x, wup, y, N, Ho, Wo, O, I, mask_act, mask_slope, stats_partial, ws, ws_bytes,
                           rg_stream(stream));
  RG_REQUIRE(!stats_partial, RG_EUNSUPPORTED, "conv_up: statistics come from the MFMA epilogue only");
  RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "conv_up: shape/dtype not supported by the MFMA kernel");
  RG_REQUIRE(w, RG_EINVAL, "conv_up: generic kernel needs the fp32 master weight");
  return rg_generic_conv_up(x, w, y, N, Ho, Wo, O, I, mask_act, mask_slope, dtype, rg_stream(stream));
}

extern "C" size_t rg_conv_wgrad_workspace_bytes(int N, int Ho, int Wo, int O, int I, int dtype, int algo) {
  size_t a = rg_generic_wgrad_ws_bytes(N, Ho, Wo, O, I);
  size_t b = 0;
  if (want_mfma(algo, dtype) && rg_mfma_wgrad_supported(N, Ho, Wo, O, I)) {
    b = rg_mfma_wgrad_ws_bytes(N, Ho, Wo, O, I);
    size_t c = rg_mfma_wgrad2_ws_bytes(N, Ho, Wo, O, I);
    if (c > b) b = c;
  }
  return a > b ? a : b;
}

extern "C" int rg_conv_wgrad2(const void* low0, const void* high0, const void* low1, const void* high1, float* dw,
                              int N, int Ho, int Wo, int O, int I, int dtype, int accumulate, int algo, void* ws,
                              size_t ws_bytes, void* stream) {
  RG_REQUIRE(low0 && high0 && low1 && high1 && dw && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL,
             "conv_wgrad2: bad args");
  if (want_mfma(algo, dtype) && rg_mfma_wgrad_supported(N, Ho, Wo, O, I))
    return rg_mfma_conv_wgrad2(low0, high0, low1, high1, dw, N, Ho, Wo, O, I, accumulate, ws, ws_bytes,
                               rg_stream(stream));
  RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "conv_wgrad2: shape/dtype not supported by the MFMA kernel");
  int rc = rg_generic_conv_wgrad(low0, high0, dw, N, Ho, Wo, O, I, dtype, accumulate, ws, ws_bytes, rg_stream(stream));
  if (rc) return rc;
  return rg_generic_conv_wgrad(low1, high1, dw, N, Ho, Wo, O, I, dtype, 1, ws, ws_bytes, rg_stream(stream));
}

// rg_conv_wgrad / rg_conv_wgrad2 (low1 / high1 may be NULL: one segment) with the split-K reduction LEFT TO THE CALLER: a plan
// with nsplit > 1 leaves its fp32 slabs [nsplit][O][16][I] in `slab` (at least rg_conv_wgrad_workspace_bytes) and does not
// touch dw; *nsplit_out says how many (1: dw was written, nothing is pending).  The consumer is rg_adam_step_slabs.
extern "C" int rg_conv_wgrad_slabs(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N,
                                   int Ho, int Wo, int O, int I, int dtype, int algo, void* slab, size_t slab_bytes,
                                   int* nsplit_out, int* slab_dtype_out, void* stream) {
  RG_REQUIRE(low0 && high0 && dw && nsplit_out && slab_dtype_out && (low1 == nullptr) == (high1 == nullptr) && N > 0 && Ho > 0 &&
                 Wo > 0 && I > 0 && O > 0, RG_EINVAL, "conv_wgrad_slabs: bad args");
  *slab_dtype_out = RG_F32;
  if (!(want_mfma(algo, dtype) && rg_mfma_wgrad_supported(N, Ho, Wo, O, I))) {
    // a shape the matrix-core kernel does not take (small models): the generic kernel reduces its own split-K through `slab`
    // as a workspace and WRITES dw -- nothing pending, *nsplit_out = 1
    *nsplit_out = 1;
    RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "conv_wgrad_slabs: shape/dtype not supported by the MFMA kernel");
    int rc = rg_generic_conv_wgrad(low0, high0, dw, N, Ho, Wo, O, I, dtype, 0, slab, slab_bytes, rg_stream(stream));
    if (rc || !low1) return rc;
    return rg_generic_conv_wgrad(low1, high1, dw, N, Ho, Wo, O, I, dtype, 1, slab, slab_bytes, rg_stream(stream));
  }
  return rg_mfma_conv_wgrad2(low0, high0, low1, high1, dw, N, Ho, Wo, O, I, 0, slab, slab_bytes, rg_stream(stream), nsplit_out,
                             slab_dtype_out);
}

// Weight gradient of a 4 x 4 stride-2 layer AND the Adam step of that tensor in one launch (low1 / high1 may be NULL): p, m, v
// (fp32) and the optional bf16 operand image `shadow_bf16` are the tensor's tap-major [O][16][I] buffers, hyper the 8 constants
// of rg_adam_hyper_dev.  The gradient is never written.  rg_conv_wgrad_adam_supported says whether the shape has such a plan
// (bf16 operands, the matrix-core kernel without split-K); the caller's streaming step must leave the tensor out
// (rg_adam_step_slabs: a segment with nsplit = -1).
extern "C" int rg_conv_wgrad_adam_supported(int N, int Ho, int Wo, int O, int I, int two, int dtype, int algo) {
  return N > 0 && Ho > 0 && Wo > 0 && O > 0 && I > 0 && want_mfma(algo, dtype) && rg_mfma_wgrad_supported(N, Ho, Wo, O, I) &&
         rg_mfma_conv_wgrad_adam_supported(N, Ho, Wo, O, I, two != 0) ? 1 : 0;
}
extern "C" int rg_conv_wgrad_adam(const void* low0, const void* high0, const void* low1, const void* high1, float* p, float* m,
                                  float* v, const float* hyper, void* shadow_bf16, int N, int Ho, int Wo, int O, int I, int dtype,
                                  int algo, void* stream) {
  RG_REQUIRE(low0 && high0 && p && m && v && hyper && (low1 == nullptr) == (high1 == nullptr) && N > 0 && Ho > 0 && Wo > 0 &&
                 I > 0 && O > 0, RG_EINVAL, "conv_wgrad_adam: bad args");
  RG_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)shadow_bf16 & 7) == 0, RG_EINVAL,
             "conv_wgrad_adam: p, m, v must be 16-byte aligned (the bf16 image 8-byte)");
  RG_REQUIRE(rg_conv_wgrad_adam_supported(N, Ho, Wo, O, I, low1 != nullptr, dtype, algo), RG_EUNSUPPORTED,
             "conv_wgrad_adam: shape / dtype without a single-split matrix-core plan");
  return rg_mfma_conv_wgrad_adam(low0, high0, low1, high1, N, Ho, Wo, O, I, p, m, v, (uint16_t*)shadow_bf16, hyper,
                                 rg_stream(stream));
}

// Data parallel: the weight gradient of a layer whose plan has no split-K (rg_conv_wgrad_adam_supported) written ONCE, as bf16,
// straight into the tensor's slice of the all-reduce wire buffer (tap-major [O][16][I] like dw) -- no fp32 gradient, no cast pass
// over it; the segment table of rg_grad_to_wire skips the tensor (nsplit = -1).
extern "C" int rg_conv_wgrad_wire(const void* low0, const void* high0, const void* low1, const void* high1, void* wire_bf16, int N,
                                  int Ho, int Wo, int O, int I, int dtype, int algo, void* stream) {
  RG_REQUIRE(low0 && high0 && wire_bf16 && (low1 == nullptr) == (high1 == nullptr) && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0,
             RG_EINVAL, "conv_wgrad_wire: bad args");
  RG_REQUIRE(((uintptr_t)wire_bf16 & 15) == 0, RG_EINVAL, "conv_wgrad_wire: the wire slice must be 16-byte aligned");
  RG_REQUIRE(rg_conv_wgrad_adam_supported(N, Ho, Wo, O, I, low1 != nullptr, dtype, algo), RG_EUNSUPPORTED,
             "conv_wgrad_wire: shape / dtype without a single-split matrix-core plan");
  return rg_mfma_conv_wgrad_wire(low0, high0, low1, high1, N, Ho, Wo, O, I, (uint16_t*)wire_bf16, rg_stream(stream));
}

extern "C" int rg_conv_wgrad(const void* low, const void* high, float* dw, int N, int Ho, int Wo, int O, int I,
                             int dtype, int accumulate, int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(low && high && dw && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL, "conv_wgrad: bad args");
  if (want_mfma(algo, dtype) && rg_mfma_wgrad_supported(N, Ho, Wo, O, I))
    return rg_mfma_conv_wgrad2(low, high, nullptr, nullptr, dw, N, Ho, Wo, O, I, accumulate, ws, ws_bytes,
                               rg_stream(stream));
  RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "conv_wgrad: shape/dtype not supported by the MFMA kernel");
  return rg_generic_conv_wgrad(low, high, dw, N, Ho, Wo, O, I, dtype, accumulate, ws, ws_bytes, rg_stream(stream));
}

extern "C" int rg_first_down(const float* x_nchw, const float* w, const float* bias, void* y, int N, int H, int W,
                             int I, int O, float slope, int dtype, void* stream) {
  RG_REQUIRE(x_nchw && w && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && I > 0 && O > 0, RG_EINVAL,
             "first_down: bad args");
  if (rg_skinny_supported(I, O) && !(dtype == RG_F32 && rg_generic_f32_image_side(N, H, W, I, O)))
    return rg_skinny_first_down(x_nchw, w, bias, y, nullptr, N, H, W, I, O, slope, dtype, rg_stream(stream));
  return rg_generic_first_down(x_nchw, w, bias, y, N, H, W, I, O, slope, dtype, rg_stream(stream));
}

// first_down that also writes the packed sign bits of its (bf16, 64-channel) output: bits[pixel] bit c = y[pixel][c] > 0
extern "C" int rg_first_down_bits(const float* x_nchw, const float* w, const float* bias, void* y, void* bits, int N, int H,
                                  int W, int I, int O, float slope, int dtype, void* stream) {
  RG_REQUIRE(x_nchw && w && y && bits && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && I > 0 && O == 64 &&
                 dtype == RG_H16, RG_EINVAL, "first_down_bits: bad args (64 bf16 output channels)");
  if (rg_skinny_supported(I, O))
    return rg_skinny_first_down(x_nchw, w, bias, y, bits, N, H, W, I, O, slope, dtype, rg_stream(stream));
  int rc = rg_generic_first_down(x_nchw, w, bias, y, N, H, W, I, O, slope, dtype, rg_stream(stream));
  return rc != RG_OK ? rc : rg_skinny_sign_pack(y, bits, (long long)N * (H / 2) * (W / 2), O, dtype, rg_stream(stream));
}

// y = lrelu'(a) * conv2d(x, w) with lrelu'(a) given as a's packed sign bits (rg_first_down_bits): the tangent of
// discriminator layer 0 in the gradient penalty's forward-mode pass.  rg_first_down_masked_supported: 0 -> use
// rg_first_down(slope = 1) + rg_lrelu_bwd.
extern "C" int rg_first_down_masked_supported(int H, int W, int I, int O, int dtype) {
  return rg_skinny_supported(I, O) && rg_skinny_first_down_masked_supported(H, W, I, O, dtype);
}
extern "C" int rg_first_down_masked(const float* x_nchw, const float* w, void* y, const void* mask_bits, float mask_slope,
                                    int N, int H, int W, int I, int O, int dtype, void* stream) {
  RG_REQUIRE(x_nchw && w && y && mask_bits && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, RG_EINVAL,
             "first_down_masked: bad args");
  RG_REQUIRE(rg_first_down_masked_supported(H, W, I, O, dtype), RG_EUNSUPPORTED, "first_down_masked: shape");
  return rg_skinny_first_down_masked(x_nchw, w, y, mask_bits, mask_slope, N, H, W, I, O, dtype, rg_stream(stream));
}

extern "C" int rg_sign_pack(const void* a, void* bits, long long npix, int C, int dtype, void* stream) {
  RG_REQUIRE(a && bits && npix > 0, RG_EINVAL, "sign_pack: bad args");
  return rg_skinny_sign_pack(a, bits, npix, C, dtype, rg_stream(stream));
}

extern "C" int rg_conv_up_maskbits_supported(int N, int Ho, int Wo, int O, int I, int dtype, int algo) {
  return want_mfma(algo, dtype) && N > 0 && Ho > 0 && Wo > 0 && rg_mfma_conv_supported(N, Ho, Wo, O, I) &&
         rg_mfma_conv_up_maskbits_supported(N, Ho, Wo, O, I);
}

// conv_up with the consumer's LeakyReLU backward fused from PACKED sign bits (rg_first_down_bits / rg_sign_pack)
extern "C" int rg_conv_up_maskbits(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I,
                                   const void* mask_bits, float mask_slope, int dtype, int algo, void* ws, size_t ws_bytes,
                                   void* stream) {
  RG_REQUIRE(x && wup && y && mask_bits, RG_EINVAL, "conv_up_maskbits: bad args");
  RG_REQUIRE(rg_conv_up_maskbits_supported(N, Ho, Wo, O, I, dtype, algo), RG_EUNSUPPORTED, "conv_up_maskbits: shape");
  return rg_mfma_conv_up(x, wup, y, N, Ho, Wo, O, I, mask_bits, mask_slope, nullptr, ws, ws_bytes, rg_stream(stream),
                         nullptr, nullptr, 1.f, 1);
}

// rg_last_up whose input is the PRE-BatchNorm tensor z: lrelu(bn_train(z)) is applied while the rows are staged (same bf16
// rounding as rg_bn_act), for generator forwards that keep nothing for a backward pass.
extern "C" int rg_last_up_pre_supported(int Wo, int O, int I, int dtype) {
  return rg_skinny_supported(I, O) && rg_skinny_last_up_pre_supported(Wo, O, dtype);
}
extern "C" int rg_last_up_pre(const void* z, const float* w, const float* bias, float* y_nchw, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, float slope, int N, int Ho, int Wo,
                              int O, int I, int apply_tanh, int dtype, void* stream) {
  RG_REQUIRE(z && w && y_nchw && mean && invstd && gamma && beta && N > 0 && Ho > 0 && Wo > 0, RG_EINVAL,
             "last_up_pre: bad args");
  RG_REQUIRE(rg_last_up_pre_supported(Wo, O, I, dtype), RG_EUNSUPPORTED, "last_up_pre: shape");
  return rg_skinny_last_up(z, w, bias, y_nchw, N, Ho, Wo, O, I, apply_tanh, dtype, rg_stream(stream), mean, invstd, gamma,
                           beta, slope);
}

extern "C" int rg_last_up(const void* x, const float* w, const float* bias, float* y_nchw, int N, int Ho, int Wo,
                          int O, int I, int apply_tanh, int dtype, void* stream) {
  RG_REQUIRE(x && w && y_nchw && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL, "last_up: bad args");
  if (rg_skinny_supported(I, O))
    return rg_skinny_last_up(x, w, bias, y_nchw, N, Ho, Wo, O, I, apply_tanh, dtype, rg_stream(stream));
  return rg_generic_last_up(x, w, bias, y_nchw, N, Ho, Wo, O, I, apply_tanh, dtype, rg_stream(stream));
}

// rg_last_up (no activation) with the first consumer's pass over the output fused into the store phase (rg_skinny.hip LuPost)
extern "C" int rg_last_up_post_blocks(int N, int Ho, int Wo, int O, int I, int dtype) {
  return rg_skinny_supported(I, O) ? rg_skinny_last_up_post_blocks(N, Ho, Wo, O, dtype) : 0;
}
extern "C" int rg_last_up_post(const void* x, const float* w, float* y_nchw, int N, int Ho, int Wo, int O, int I, int dtype,
                               const float* tanh_img, float* part, void* stream) {
  RG_REQUIRE(x && w && y_nchw && N > 0 && Ho > 0 && Wo > 0 && (tanh_img || part), RG_EINVAL, "last_up_post: bad args");
  RG_REQUIRE(rg_last_up_post_blocks(N, Ho, Wo, O, I, dtype) > 0, RG_EUNSUPPORTED, "last_up_post: shape");
  RG_REQUIRE((((uintptr_t)tanh_img | (uintptr_t)part) & 15) == 0, RG_EINVAL, "last_up_post: 16-byte aligned buffers required");
  return rg_skinny_last_up(x, w, nullptr, y_nchw, N, Ho, Wo, O, I, 0, dtype, rg_stream(stream), nullptr, nullptr, nullptr,
                           nullptr, 1.f, tanh_img, part);
}
extern "C" int rg_last_up_part_chan_sum(const float* part, int nblocks, float* out3, int accumulate, void* stream) {
  RG_REQUIRE(part && out3 && nblocks > 0, RG_EINVAL, "last_up_part_chan_sum: bad args");
  return rg_skinny_lu_part_final(part, nblocks, out3, accumulate, 0, nullptr, nullptr, 0.f, rg_stream(stream));
}
extern "C" int rg_gp_coef_parts_scaled(const float* part, int nblocks, float* sq, float* loss, float* coef, float lambd,
                                       float in_scale, float out_scale, void* stream) {
  RG_REQUIRE(part && loss && coef && nblocks > 0 && in_scale > 0.f && out_scale > 0.f, RG_EINVAL, "gp_coef_parts: bad args");
  return rg_skinny_lu_part_final(part, nblocks, sq, 0, 1, loss, coef, lambd, rg_stream(stream), in_scale, out_scale);
}
// the 16-bit storage type of this build of the library: RG_BF16 (librnagan_hip.so) or RG_F16 (librnagan_hip_f16.so)
extern "C" int rg_storage_dtype(void) { return RG_H16; }
extern "C" int rg_gp_coef_parts(const float* part, int nblocks, float* sq, float* loss, float* coef, float lambd,
                                void* stream) {
  RG_REQUIRE(part && loss && coef && nblocks > 0, RG_EINVAL, "gp_coef_parts: bad args");
  return rg_skinny_lu_part_final(part, nblocks, sq, 0, 1, loss, coef, lambd, rg_stream(stream));
}

extern "C" size_t rg_skinny_wgrad_workspace_bytes(int N, int Ho, int Wo, int O, int I) {
  size_t a = rg_generic_wgrad_ws_bytes(N, Ho, Wo, O, I);
  size_t b = rg_skinny_supported(I, O) ? rg_skinny_wgrad_ws_bytes(N, Ho, Wo, O, I) : 0;
  return a > b ? a : b;
}

extern "C" int rg_skinny_wgrad(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                               int dtype, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(low && high_nchw && dw && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL,
             "skinny_wgrad: bad args");
  if (rg_skinny_supported(I, O) && !(dtype == RG_F32 && rg_generic_f32_image_side(N, 2 * Ho, 2 * Wo, I, O)))
    return rg_skinny_wgrad_impl(low, high_nchw, dw, N, Ho, Wo, O, I, dtype, accumulate, ws, ws_bytes,
                                rg_stream(stream));
  return rg_generic_skinny_wgrad(low, high_nchw, dw, N, Ho, Wo, O, I, dtype, accumulate, ws, ws_bytes,
                                 rg_stream(stream));
}

// rg_skinny_wgrad that ALSO produces the layer's bias gradient dbias[O] = sum over the pixels of `low` (the Conv2d(3, 64) /
// ConvTranspose2d(64, 3) pair keeps a bias, src/histopathology_gan.py:186-192) where the kernel can form it as a by-product of its
// pass over `low` (a column of ones in the patch operand: the 256 x 256 bf16 row kernel) -- *bias_done_out = 1; otherwise 0 and
// dbias is untouched (call rg_col_sum).  dbias may be NULL (= rg_skinny_wgrad).
extern "C" int rg_skinny_wgrad_bias(const void* low, const float* high_nchw, float* dw, float* dbias, int N, int Ho, int Wo, int O,
                                    int I, int dtype, int accumulate, int bias_accumulate, void* ws, size_t ws_bytes,
                                    int* bias_done_out, void* stream) {
  RG_REQUIRE(low && high_nchw && dw && bias_done_out && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL,
             "skinny_wgrad_bias: bad args");
  *bias_done_out = 0;
  if (rg_skinny_supported(I, O) && !(dtype == RG_F32 && rg_generic_f32_image_side(N, 2 * Ho, 2 * Wo, I, O)))
    return rg_skinny_wgrad_impl(low, high_nchw, dw, N, Ho, Wo, O, I, dtype, accumulate, ws, ws_bytes, rg_stream(stream),
                                O == 64 ? dbias : nullptr, bias_accumulate, bias_done_out);
  return rg_generic_skinny_wgrad(low, high_nchw, dw, N, Ho, Wo, O, I, dtype, accumulate, ws, ws_bytes, rg_stream(stream));
}

// rg_skinny_wgrad with the per-workgroup partial gradients LEFT to the caller's optimizer step (rg_adam_step_slabs): `slab`
// receives *nslab_out fp32 slabs of O * 48 elements in dw's layout (at most rg_skinny_wgrad_workspace_bytes); a second
// contribution to the same tensor is a second call with `slab` advanced by the first one's slabs.  *nslab_out = 0: no such form
// for this shape / dtype, nothing launched (call rg_skinny_wgrad).  bias_slab (optional): receives the bias-gradient partials
// [*nslab_out][O] of the same pass when *bias_done_out comes back 1 (see rg_skinny_wgrad_bias).
extern "C" int rg_skinny_wgrad_slabs(const void* low, const float* high_nchw, int N, int Ho, int Wo, int O, int I, int dtype,
                                     void* slab, size_t slab_bytes, int* nslab_out, float* bias_slab, int* bias_done_out,
                                     void* stream) {
  RG_REQUIRE(low && high_nchw && slab && nslab_out && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0 &&
                 (bias_slab == nullptr || bias_done_out != nullptr), RG_EINVAL, "skinny_wgrad_slabs: bad args");
  *nslab_out = 0;
  if (bias_done_out) *bias_done_out = 0;
  if (!rg_skinny_supported(I, O)) return RG_OK;
  return rg_skinny_wgrad_slabs_impl(low, high_nchw, N, Ho, Wo, O, I, dtype, slab, slab_bytes, nslab_out, bias_slab, bias_done_out,
                                    rg_stream(stream));
}

// ------------------------------------------------------------------------------------------------
extern "C" int rg_pack_conv_wup_from_bf16(const void* w_bf16, void* wup, int O, int I, void* stream) {
  RG_REQUIRE(w_bf16 && wup && O > 0 && I > 0, RG_EINVAL, "pack_conv_wup_from_bf16: bad args");
  return rg_mfma_transpose_bf16(w_bf16, wup, O, 16 * I, 0, rg_stream(stream));
}
// the transposed-conv weight images of up to 8 layers in one launch (each as rg_pack_conv_wup_from_bf16)
extern "C" int rg_pack_conv_wup_from_bf16_multi(int n, const void* const* w_bf16, void* const* wup, const int* O, const int* I,
                                                void* stream) {
  RG_REQUIRE(n >= 1 && n <= 8 && w_bf16 && wup && O && I, RG_EINVAL, "pack_conv_wup_from_bf16_multi: bad args");
  int Cc[8];
  for (int i = 0; i < n; ++i) Cc[i] = 16 * I[i];
  return rg_mfma_transpose_bf16_multi(n, w_bf16, wup, O, Cc, rg_stream(stream));
}
extern "C" int rg_pack_g0_weight_from_bf16(const void* w_bf16, void* wp, int E, int C, void* stream) {
  RG_REQUIRE(w_bf16 && wp && E > 0 && C > 0, RG_EINVAL, "pack_g0_weight_from_bf16: bad args");
  return rg_mfma_transpose_bf16(w_bf16, wp, E, 16 * C, 1, rg_stream(stream));
}
extern "C" int rg_pack_g0_weight(const float* w, void* wp, int E, int C, int dtype, void* stream) {
  RG_REQUIRE(w && wp && E > 0 && C > 0, RG_EINVAL, "pack_g0_weight: bad args");
  RG_REQUIRE(dtype == RG_H16, RG_EUNSUPPORTED, "pack_g0_weight: only bf16 packs exist");
  return rg_mfma_pack_g0_weight(w, wp, E, C, rg_stream(stream));
}

extern "C" size_t rg_g0_workspace_bytes(int N, int E, int C, int dtype, int algo) {
  if (!want_mfma(algo, dtype)) return 0;
  size_t a = (size_t)N * E * 2;                       // forward: bf16 copy of z
  size_t b = rg_mfma_g0_wgrad_ws_bytes(N, E, C);      // wgrad: k-contiguous bf16 images of z and gy
  return a > b ? a : b;
}

extern "C" int rg_g0_fwd(const float* z, const float* w, const void* wp, void* y, int N, int E, int C, int dtype,
                         int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && y && N > 0 && E > 0 && C > 0, RG_EINVAL, "g0_fwd: bad args");
  if (want_mfma(algo, dtype) && wp && rg_mfma_plain_supported(N, E, 16 * C)) {
    RG_REQUIRE(ws && ws_bytes >= (size_t)N * E * 2, RG_EWORKSPACE, "g0_fwd: workspace too small");
    int rc = rg_cast_pad(z, ws, N, E, E, RG_H16, stream);
    if (rc) return rc;
    return rg_mfma_gemm_plain(ws, wp, y, N, E, 16 * C, 16 * C, rg_stream(stream));
  }
  RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "g0_fwd: shape/dtype not supported by the MFMA kernel");
  RG_REQUIRE(w, RG_EINVAL, "g0_fwd: generic kernel needs the fp32 master weight");
  return rg_generic_g0_fwd(z, w, y, N, E, C, dtype, rg_stream(stream));
}

extern "C" int rg_g0_fwd_affine(const float* z, const void* wp, void* y, int N, int E, int C, const float* scale,
                                const float* shift, float slope, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && wp && y && scale && shift && N > 0 && E > 0 && C > 0, RG_EINVAL, "g0_fwd_affine: bad args");
  RG_REQUIRE(rg_mfma_plain_supported(N, E, 16 * C), RG_EUNSUPPORTED, "g0_fwd_affine: shape not supported by the MFMA kernel");
  RG_REQUIRE(ws && ws_bytes >= (size_t)N * E * 2, RG_EWORKSPACE, "g0_fwd_affine: workspace too small");
  int rc = rg_cast_pad(z, ws, N, E, E, RG_H16, stream);
  if (rc) return rc;
  return rg_mfma_gemm_plain(ws, wp, y, N, E, 16 * C, 16 * C, rg_stream(stream), scale, shift, slope);
}

extern "C" int rg_conv_up_affine(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I,
                                 const float* scale, const float* shift, float slope, void* ws, size_t ws_bytes,
                                 void* stream) {
  RG_REQUIRE(x && wup && y && scale && shift && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL,
             "conv_up_affine: bad args");
  RG_REQUIRE(rg_mfma_conv_supported(N, Ho, Wo, /*Kc=*/O, /*Ncols=*/I), RG_EUNSUPPORTED,
             "conv_up_affine: shape not supported by the MFMA kernel");
  return rg_mfma_conv_up(x, wup, y, N, Ho, Wo, O, I, nullptr, 1.f, nullptr, ws, ws_bytes, rg_stream(stream), scale, shift,
                         slope);
}

extern "C" int rg_conv_up_fp8(const void* x8, const void* wup8, void* y, int N, int Ho, int Wo, int O, int I,
                              const float* scale, const float* shift, float slope, int out_fp8, void* stream) {
  RG_REQUIRE(x8 && wup8 && y && scale && shift && N > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, RG_EINVAL, "conv_up_fp8: bad args");
  RG_REQUIRE(rg_is_pow2(Ho) && rg_is_pow2(Wo) && rg_mfma_fp8_supported(N * Ho * Wo, 4 * O, I, 4), RG_EUNSUPPORTED,
             "conv_up_fp8: shape not supported (O, I multiples of 128, at least 256 / 512 low-resolution pixels)");
  return rg_mfma_conv_up_fp8(x8, wup8, y, N, Ho, Wo, O, I, scale, shift, slope, out_fp8, rg_stream(stream));
}
extern "C" int rg_gemm_fp8(const void* a8, const void* b8, void* y, int M, int K, int Ncols, const float* scale,
                           const float* shift, float slope, int out_fp8, void* stream) {
  RG_REQUIRE(a8 && b8 && y && scale && shift && M > 0 && K > 0 && Ncols > 0, RG_EINVAL, "gemm_fp8: bad args");
  RG_REQUIRE(rg_mfma_fp8_supported(M, K, Ncols, 1), RG_EUNSUPPORTED, "gemm_fp8: shape not supported");
  return rg_mfma_gemm_fp8(a8, b8, y, M, K, Ncols, scale, shift, slope, out_fp8, rg_stream(stream));
}
extern "C" int rg_fp8_supported(int M, int K, int Ncols, int taps) { return rg_mfma_fp8_supported(M, K, Ncols, taps) ? 1 : 0; }

extern "C" int rg_g0_wgrad(const float* z, const void* gy, float* dw, int N, int E, int C, int dtype, int accumulate,
                           int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && gy && dw && N > 0 && E > 0 && C > 0, RG_EINVAL, "g0_wgrad: bad args");
  // output-streaming bound (|dw| = 16*E*C fp32 written once)
  if (want_mfma(algo, dtype) && !accumulate && rg_mfma_g0_wgrad_supported(N, E, C) && ws &&
      ws_bytes >= rg_mfma_g0_wgrad_ws_bytes(N, E, C))
    return rg_mfma_g0_wgrad(z, gy, dw, N, E, C, ws, ws_bytes, rg_stream(stream));
  return rg_generic_g0_wgrad(z, gy, dw, N, E, C, dtype, accumulate, rg_stream(stream));
}

extern "C" int rg_pack_linear_weight(const float* w, void* wp, int Nout, int K, int Nout_pad, int K_pad, void* stream) {
  RG_REQUIRE(w && wp && Nout > 0 && K > 0 && Nout_pad >= Nout && K_pad >= K, RG_EINVAL, "pack_linear_weight: bad args");
  return rg_mfma_pack_linear_weight(w, wp, Nout, K, Nout_pad, K_pad, rg_stream(stream));
}

extern "C" size_t rg_linear_workspace_bytes(int M, int K, int Nout, int algo) {
  if (algo == RG_ALGO_GENERIC) return rg_generic_linear_ws_bytes(M, K, Nout);     // split-K slabs of the fp32 kernel (0: no split)
  size_t kp = rg_align_up((size_t)K, 64);
  return rg_align_up((size_t)M * kp * 2, 256) + rg_mfma_linear_ws_bytes(M, (int)kp, Nout);
}

extern "C" int rg_linear_affine_act(const float* x, int ldx, const float* w, const void* wp, const float* scale,
                                    const float* shift, float* y, int ldy, int M, int K, int Nout, float slope,
                                    int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && y && M > 0 && K > 0 && Nout > 0 && ldx >= K && ldy >= Nout, RG_EINVAL, "linear: bad args");
  // AUTO falls back to the generic kernel for output widths the MFMA epilogue does not take (Nout % 8)
  if (algo != RG_ALGO_GENERIC && wp && (Nout % 8 == 0 || algo == RG_ALGO_MFMA)) {
    int kp = (int)rg_align_up((size_t)K, 64);
    size_t xb = rg_align_up((size_t)M * kp * 2, 256);
    RG_REQUIRE(ws && ws_bytes >= (size_t)M * kp * 2, RG_EWORKSPACE, "linear: workspace too small");
    RG_REQUIRE(ldx == K, RG_EINVAL, "linear(MFMA): x must be dense");
    int rc = rg_cast_pad(x, ws, M, K, kp, RG_H16, stream);
    if (rc) return rc;
    void* slab = ws_bytes > xb ? (char*)ws + xb : nullptr;
    return rg_mfma_linear(ws, wp, scale, shift, y, ldy, M, kp, Nout, slope, slab, ws_bytes > xb ? ws_bytes - xb : 0,
                          rg_stream(stream));
  }
  RG_REQUIRE(algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "linear: MFMA path needs the packed weight");
  RG_REQUIRE(w, RG_EINVAL, "linear: generic kernel needs the fp32 weight");
  return rg_generic_linear(x, ldx, w, scale, shift, y, ldy, M, K, Nout, slope, rg_stream(stream), ws, ws_bytes);
}

/* ---- resize-convolution block of DCGANUpGenerator (src/dcgan.py:45-56,76-84) ---- */
extern "C" size_t rg_upconv3_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  size_t b = rg_generic_upconv3_ws_bytes(N, H, W, Cin, Cout);
  if (rg_mfma_upconv3_supported(N, H, W, Cin, Cout)) b = std::max(b, rg_mfma_upconv3_fwd_ws_bytes(N, H, W, Cin, Cout));
  if (rg_mfma_upconv3_bwd_supported(N, H, W, Cin, Cout)) b = std::max(b, rg_mfma_upconv3_bwd_ws_bytes(N, H, W, Cin, Cout));
  if (rg_mfma_upconv3_wgrad_supported(N, H, W, Cin, Cout)) b = std::max(b, rg_mfma_upconv3_wgrad_ws_bytes(N, H, W, Cin, Cout));
  b = std::max(b, rg_upimg_wgrad_ws_bytes(N, H, W, Cin, Cout));
  if (rg_mfma_upconv3_image_supported(N, H, W, Cin, Cout))
    b = std::max(b, std::max(rg_mfma_upconv3_image_fwd_ws_bytes(N, H, W, Cin, Cout),
                             rg_mfma_upconv3_image_wgrad_ws_bytes(N, H, W, Cin, Cout)));
  return b;
}
extern "C" int rg_upconv3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                              int Cout, int out_nchw_f32, int dtype, int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, RG_EINVAL, "upconv3_fwd: bad args");
  // bf16 NHWC output with Cin % 64 == 0: matrix cores (pad image + 9-tap implicit GEMM); otherwise the functor kernel
  const bool mfma_ok = dtype == RG_H16 && (out_nchw_f32 ? rg_mfma_upconv3_image_supported(N, H, W, Cin, Cout)
                                                         : rg_mfma_upconv3_supported(N, H, W, Cin, Cout));
  RG_REQUIRE(mfma_ok || algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "upconv3_fwd: shape/dtype not supported by the MFMA kernel");
  if (mfma_ok && algo != RG_ALGO_GENERIC && out_nchw_f32 && rg_option("upimg", 1) && rg_upimg_fwd_supported(N, H, W, Cin, Cout))
    return rg_upimg_fwd(x, w, bias, (float*)y, N, H, W, Cin, Cout, rg_stream(stream));     // upsample + pad formed in LDS
  if (mfma_ok && algo != RG_ALGO_GENERIC && out_nchw_f32)
    return rg_mfma_upconv3_image_fwd(x, w, bias, (float*)y, N, H, W, Cin, Cout, ws, ws_bytes, rg_stream(stream));
  if (mfma_ok && algo != RG_ALGO_GENERIC)
    return rg_mfma_upconv3_fwd(x, w, bias, y, N, H, W, Cin, Cout, ws, ws_bytes, rg_stream(stream));
  return rg_generic_upconv3_fwd(x, w, bias, y, N, H, W, Cin, Cout, out_nchw_f32, dtype, rg_stream(stream));
}
extern "C" int rg_upconv3_bwd_data(const void* gy, int gy_nchw_f32, const float* w, void* gx, int N, int H, int W,
                                   int Cin, int Cout, int dtype, int algo, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(gy && w && gx && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, RG_EINVAL, "upconv3_bwd_data: bad args");
  const bool mfma_ok = dtype == RG_H16 && !gy_nchw_f32 && rg_mfma_upconv3_bwd_supported(N, H, W, Cin, Cout);
  RG_REQUIRE(mfma_ok || algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "upconv3_bwd_data: shape/dtype not supported by the MFMA kernel");
  if (dtype == RG_H16 && gy_nchw_f32 && algo != RG_ALGO_GENERIC && rg_option("upimg", 1) && rg_upimg_bwd_supported(N, H, W, Cin, Cout))
    return rg_upimg_bwd_data((const float*)gy, w, gx, N, H, W, Cin, Cout, rg_stream(stream));     // the image block, fused
  if (mfma_ok && algo != RG_ALGO_GENERIC)
    return rg_mfma_upconv3_bwd_data(gy, w, gx, N, H, W, Cin, Cout, ws, ws_bytes, rg_stream(stream));
  return rg_generic_upconv3_bwd_data(gy, gy_nchw_f32, w, gx, N, H, W, Cin, Cout, dtype, ws, ws_bytes, rg_stream(stream));
}
extern "C" int rg_upconv3_wgrad(const void* gy, int gy_nchw_f32, const void* x, float* dw, int N, int H, int W, int Cin,
                                int Cout, int dtype, int algo, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(gy && x && dw && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, RG_EINVAL, "upconv3_wgrad: bad args");
  const bool mfma_ok = dtype == RG_H16 && (gy_nchw_f32 ? rg_mfma_upconv3_image_supported(N, H, W, Cin, Cout)
                                                        : rg_mfma_upconv3_wgrad_supported(N, H, W, Cin, Cout));
  RG_REQUIRE(mfma_ok || algo != RG_ALGO_MFMA, RG_EUNSUPPORTED, "upconv3_wgrad: shape/dtype not supported by the MFMA kernel");
  if (mfma_ok && algo != RG_ALGO_GENERIC && gy_nchw_f32 && rg_option("upimg", 1) && rg_upimg_wgrad_supported(N, H, W, Cin, Cout) &&
      ws && ws_bytes >= rg_upimg_wgrad_ws_bytes(N, H, W, Cin, Cout))
    return rg_upimg_wgrad((const float*)gy, x, dw, N, H, W, Cin, Cout, accumulate, ws, ws_bytes, rg_stream(stream));
  if (mfma_ok && algo != RG_ALGO_GENERIC && gy_nchw_f32)
    return rg_mfma_upconv3_image_wgrad((const float*)gy, x, dw, N, H, W, Cin, Cout, accumulate, ws, ws_bytes, rg_stream(stream));
  if (mfma_ok && algo != RG_ALGO_GENERIC)
    return rg_mfma_upconv3_wgrad(gy, x, dw, N, H, W, Cin, Cout, accumulate, ws, ws_bytes, rg_stream(stream));
  return rg_generic_upconv3_wgrad(gy, gy_nchw_f32, x, dw, N, H, W, Cin, Cout, dtype, accumulate, ws, ws_bytes,
                                  rg_stream(stream));
}
