// rg_internal.h -- host-side functions shared between the translation units of librnagan_hip.so.
#pragma once
#include "rg_common.h"

// rg_api.hip: kernel-selection knob `name` (override set through rg_set_option, else RNAGAN_<NAME>, else dflt)
int rg_option(const char* name, int dflt);

// rg_generic.hip
int rg_reduce_slabs(const float* slab, float* dst, size_t n, int nsplit, int accumulate, int perm_mode, int Q,
                    hipStream_t st);
int rg_generic_conv_down(const void* x, const float* w, void* y, int N, int Hi, int Wi, int I, int O, int dtype,
                         hipStream_t st);
int rg_generic_conv_up(const void* x, const float* w, void* y, int N, int Ho, int Wo, int O, int I, const void* mask,
                       float mslope, int dtype, hipStream_t st);
bool rg_generic_f32_image_side(int N, int H, int W, int I, int O);
int rg_generic_first_down(const float* x, const float* w, const float* bias, void* y, int N, int H, int W, int I,
                          int O, float slope, int dtype, hipStream_t st);
int rg_generic_last_up(const void* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo, int O,
                       int I, int apply_tanh, int dtype, hipStream_t st);
size_t rg_generic_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I);
int rg_generic_conv_wgrad(const void* low, const void* high, float* dw, int N, int Ho, int Wo, int O, int I,
                          int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int rg_generic_skinny_wgrad(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                            int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int rg_generic_g0_fwd(const float* z, const float* w, void* y, int N, int E, int C, int dtype, hipStream_t st);
int rg_generic_g0_wgrad(const float* z, const void* gy, float* dw, int N, int E, int C, int dtype, int accumulate,
                        hipStream_t st);
int rg_generic_linear(const float* x, int ldx, const float* w, const float* scale, const float* shift, float* y,
                      int ldy, int M, int K, int Nout, float slope, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
size_t rg_generic_linear_ws_bytes(int M, int K, int Nout);

// BatchNorm-backward sums of the consuming block computed in a conv launch's epilogue (GArgs::bwd_z, rg_conv8.hip)
struct RgBnBwdFuse {
  const void* z; const float* mean; const float* invstd; const float* gamma; const float* beta; float slope;
  int groups; float* sums;
};

// rg_mfma.hip (bf16 MFMA implicit-GEMM kernels)
bool rg_mfma_conv_supported(int N, int Hq, int Wq, int Kc, int Ncols);
bool rg_mfma_plain_supported(int M, int K, int Ncols);
bool rg_mfma_wgrad_supported(int N, int Ho, int Wo, int O, int I);
size_t rg_mfma_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I);
int rg_mfma_pack_conv_weight(const float* w, void* wdn, void* wup, int O, int I, hipStream_t st);
int rg_mfma_pack_g0_weight(const float* w, void* wp, int E, int C, hipStream_t st);
int rg_mfma_pack_linear_weight(const float* w, void* wp, int Nout, int K, int Nout_pad, int K_pad, hipStream_t st);
int rg_mfma_conv_down(const void* x, const void* wdn, void* y, int N, int Hi, int Wi, int I, int O, float* stats,
                      void* ws, size_t ws_bytes, hipStream_t st, int defer_reduce = 0, const RgBnBwdFuse* bf = nullptr);
int rg_mfma_conv_bnbwd_rows(int up, int N, int Hlow, int Wlow, int O, int I, int groups);
int rg_mfma_conv_nsplit(int up, int N, int Hlow, int Wlow, int O, int I);
// split-K partial tiles of the 8-wave conv kernel as bf16 instead of fp32 (option `slab16`): half the slab bytes written by the
// conv launch and read by the fused reduction + BatchNorm kernel; the partial sums are rounded to bf16 before they are added
constexpr int RG_SLAB16_DEFAULT = 1;
constexpr int RG_BN_REV_DEFAULT = 4;     // option bn_rev: BatchNorm row passes that walk their rows from the END (bit 0: backward-kind applies, bit 1: forward
                                         // applies, bit 2: reductions).  Measured: reductions only
// (fp16 build: weight gradients carry the static loss scale and are the LARGE sums of the backward pass -- 2e5 was seen on the
// critic head in the penalty step at S = 4096 -- so their partial sums stay fp32 there: fp16 ends at 65 504)
#ifdef RG_HALF_F16
constexpr int RG_WSLAB16_DEFAULT = 0;
#else
constexpr int RG_WSLAB16_DEFAULT = 1;
#endif
constexpr int RG_WGRAD8_MFMA_DEFAULT = 32;   // option wgrad8_mfma: 32 = v_mfma_f32_32x32x16, 16 = v_mfma_f32_16x16x32 in wgrad8_kernel
constexpr int RG_F32MMA_DEFAULT = 2;     // option f32mma: 0 vector ALU, 1 f32 matrix cores, 2 (default since the end of round 5) the same with the
                                         // 128 x 128-tile structured launches as six bf16 matrix-core products per fp32 product    // deferred split-K weight-gradient slabs (rg_conv_wgrad_slabs) as bf16: option wslab16
int rg_mfma_conv_slab16(int up, int N, int Hlow, int Wlow, int O, int I);
int rg_mfma_conv_stats_rows(int up, int N, int Hlow, int Wlow, int O, int I);
int rg_mfma_conv_up(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I, const void* mask,
                    float mslope, float* stats, void* ws, size_t ws_bytes, hipStream_t st, const float* scale = nullptr,
                    const float* shift = nullptr, float slope = 1.f, int mask_packed = 0, int defer_reduce = 0,
                    const RgBnBwdFuse* bf = nullptr);
// packed-mask form of the transposed conv's fused LeakyReLU backward (rg_convp.hip): shapes that take it
bool rg_mfma_conv_up_maskbits_supported(int N, int Ho, int Wo, int O, int I);
size_t rg_mfma_conv_ws_bytes(int up, int N, int Hlow, int Wlow, int O, int I);
bool rg_mfma_upconv3_supported(int N, int H, int W, int Cin, int Cout);
size_t rg_mfma_upconv3_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_mfma_upconv3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                        int Cout, void* ws, size_t ws_bytes, hipStream_t st);
bool rg_mfma_upconv3_bwd_supported(int N, int H, int W, int Cin, int Cout);
size_t rg_mfma_upconv3_bwd_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_mfma_upconv3_bwd_data(const void* gy, const float* w, void* gx, int N, int H, int W, int Cin, int Cout, void* ws,
                             size_t ws_bytes, hipStream_t st);
bool rg_mfma_upconv3_wgrad_supported(int N, int H, int W, int Cin, int Cout);
size_t rg_mfma_upconv3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_mfma_upconv3_wgrad(const void* gy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st);
// rg_upimg.hip: the resize-convolution generator's image block without the materialised upsample + pad image
bool rg_upimg_fwd_supported(int N, int H, int W, int Cin, int Cout);
int rg_upimg_fwd(const void* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, hipStream_t st);
bool rg_upimg_bwd_supported(int N, int H, int W, int Cin, int Cout);
int rg_upimg_bwd_data(const float* gy, const float* w, void* gx, int N, int H, int W, int Cin, int Cout, hipStream_t st);
bool rg_upimg_wgrad_supported(int N, int H, int W, int Cin, int Cout);
size_t rg_upimg_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_upimg_wgrad(const float* gy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout, int accumulate, void* ws,
                   size_t ws_bytes, hipStream_t st);
bool rg_mfma_upconv3_image_supported(int N, int H, int W, int Cin, int Cout);
size_t rg_mfma_upconv3_image_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_mfma_upconv3_image_fwd(const void* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                              int Cout, void* ws, size_t ws_bytes, hipStream_t st);
size_t rg_mfma_upconv3_image_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_mfma_upconv3_image_wgrad(const float* gy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout,
                                int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
size_t rg_generic_upconv3_ws_bytes(int N, int H, int W, int Cin, int Cout);
int rg_generic_upconv3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                           int Cout, int out_nchw, int dtype, hipStream_t st);
int rg_generic_upconv3_bwd_data(const void* gy, int gy_nchw, const float* w, void* gx, int N, int H, int W, int Cin,
                                int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t st);
int rg_generic_upconv3_wgrad(const void* gy, int gy_nchw, const void* x, float* dw, int N, int H, int W, int Cin,
                             int Cout, int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int rg_mfma_transpose_bf16(const void* src, void* dst, int R, int Cc, int permute, hipStream_t st);
int rg_mfma_transpose_bf16_multi(int n, const void* const* src, void* const* dst, const int* R, const int* Cc, hipStream_t st);
size_t rg_mfma_g0_wgrad_ws_bytes(int N, int E, int C);
bool rg_mfma_g0_wgrad_supported(int N, int E, int C);
int rg_mfma_g0_wgrad(const float* z, const void* gy, float* dw, int N, int E, int C, void* ws, size_t ws_bytes,
                     hipStream_t st);
int rg_mfma_gemm_plain(const void* a, const void* bt, void* c, int M, int K, int Ncols, int ldc, hipStream_t st,
                       const float* scale = nullptr, const float* shift = nullptr, float slope = 1.f);
int rg_mfma_linear(const void* a, const void* bt, const float* scale, const float* shift, float* y, int ldy, int M,
                   int Kpad, int Nout, float slope, void* ws, size_t ws_bytes, hipStream_t st);
size_t rg_mfma_linear_ws_bytes(int M, int Kpad, int Nout);
int rg_mfma_conv_wgrad(const void* low, const void* high, float* dw, int N, int Ho, int Wo, int O, int I,
                       int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

int rg_mfma_conv_wgrad2(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N,
                        int Ho, int Wo, int O, int I, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                        int* nsplit_out = nullptr, int* slab_dtype_out = nullptr);
size_t rg_mfma_wgrad2_ws_bytes(int N, int Ho, int Wo, int O, int I);

bool rg_mfma_conv_up_planes64_supported(int N, int Ho, int Wo, int O, int I, int products);
int rg_mfma_conv_up_planes64(const void* xp, const void* wp, float* y, int N, int Ho, int Wo, int O, int I, int products,
                             hipStream_t st, const float* maskf = nullptr, float mslope = 1.f);
bool rg_mfma_fp8_supported(int M, int K, int Ncols, int taps);
int rg_mfma_conv_up_fp8(const void* x8, const void* wup8, void* y, int N, int Ho, int Wo, int O, int I, const float* scale,
                        const float* shift, float slope, int out_fp8, hipStream_t st);
int rg_mfma_gemm_fp8(const void* a8, const void* b8, void* y, int M, int K, int Ncols, const float* scale, const float* shift,
                     float slope, int out_fp8, hipStream_t st);

// rg_conv8.hip (8-wave ping-pong gather GEMM; args = G2Args of rg_gather.h)
int rg_conv8_launch(int mode, const void* args, int bm, unsigned gx, unsigned gy, unsigned gz, hipStream_t st);
int rg_conv8n_launch(const void* args, unsigned tiles_m, hipStream_t st);
// rg_convp.hip: 128 -> 64 channel transposed conv with the input patch resident in LDS
bool rg_convp_supported(int M, int Ncols, int Cin, int Hs, int Ws);
constexpr int RG_CONVP_DEFAULT = 1;
// rg_convd.hip: 64 -> 128 channel stride-2 conv (128-pixel-wide input) with the input's parity planes resident in LDS
bool rg_convd_supported(int M, int Ncols, int Cin, int Hs, int Ws);
int rg_convd_stats_rows(int M);
int rg_convd_launch(const void* args, hipStream_t st);
int rg_convp_tiles(int M);
int rg_convp_launch(const void* args, hipStream_t st);

// rg_wgrad8.hip (8-wave ping-pong weight gradient)
bool rg_wgrad8_supported(int K, int O, int I);
int rg_wgrad8_split(int K, int O, int I, int* kt_per_split);
bool rg_wgrad8n_supported(int K, int O, int I);
int rg_wgrad8n_split(int K, int O, int I, int* kt_per_split);
int rg_wgrad8n_launch(const void* low0, const void* high0, const void* low1, const void* high1, float* out, int Kseg,
                      int two, int O, int I, int Ho, int Wo, int nsplit, int kt_per_split, int accumulate, hipStream_t st,
                      int slab16 = 0);
int rg_wgrad8_launch(const void* low0, const void* high0, const void* low1, const void* high1, float* out, int Kseg,
                     int two, int O, int I, int Ho, int Wo, int nsplit, int kt_per_split, int accumulate, hipStream_t st,
                     int slab16 = 0);
int rg_wgrad8_wire_launch(const void* low0, const void* high0, const void* low1, const void* high1, uint16_t* out16, int Kseg,
                          int two, int O, int I, int Ho, int Wo, int kt_per_split, hipStream_t st);
int rg_mfma_conv_wgrad_wire(const void* low0, const void* high0, const void* low1, const void* high1, int N, int Ho, int Wo,
                            int O, int I, uint16_t* out16, hipStream_t st);
int rg_wgrad8_adam_launch(const void* low0, const void* high0, const void* low1, const void* high1, int Kseg, int two, int O,
                          int I, int Ho, int Wo, int kt_per_split, float* p, float* m, float* v, uint16_t* shadow,
                          const float* hyper, hipStream_t st);
// weight gradient + Adam step of the tensor in one launch (plans of the 256 x 256 ping-pong kernel without split-K)
bool rg_mfma_conv_wgrad_adam_supported(int N, int Ho, int Wo, int O, int I, bool two);
int rg_mfma_conv_wgrad_adam(const void* low0, const void* high0, const void* low1, const void* high1, int N, int Ho, int Wo,
                            int O, int I, float* p, float* m, float* v, uint16_t* shadow, const float* hyper, hipStream_t st);

// rg_skinny.hip (image-side 3-channel layers)
bool rg_skinny_supported(int I, int O);
int rg_skinny_first_down(const float* x, const float* w, const float* bias, void* y, void* bits, int N, int H, int W, int I,
                         int O, float slope, int dtype, hipStream_t st);
int rg_skinny_sign_pack(const void* a, void* bits, long long npix, int C, int dtype, hipStream_t st);
bool rg_skinny_first_down_masked_supported(int H, int W, int I, int O, int dtype);
int rg_skinny_first_down_masked(const float* x, const float* w, void* y, const void* mask_bits, float mslope, int N, int H,
                                int W, int I, int O, int dtype, hipStream_t st);
int rg_skinny_last_up(const void* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo, int O, int I,
                      int apply_tanh, int dtype, hipStream_t st, const float* pre_mean = nullptr,
                      const float* pre_invstd = nullptr, const float* pre_gamma = nullptr, const float* pre_beta = nullptr,
                      float pre_slope = 1.f, const float* post_tb_img = nullptr, float* post_part = nullptr);
bool rg_skinny_last_up_pre_supported(int Wo, int O, int dtype);
int rg_skinny_last_up_post_blocks(int N, int Ho, int Wo, int O, int dtype);
int rg_skinny_lu_part_final(const float* part, int nb, float* out, int accumulate, int mode, float* loss, float* coef,
                            float lambd, hipStream_t st, float in_scale = 1.f, float out_scale = 1.f);
size_t rg_skinny_wgrad_ws_bytes(int N, int Ho, int Wo, int O, int I);
int rg_skinny_wgrad_slabs_impl(const void* low, const float* high_nchw, int N, int Ho, int Wo, int O, int I, int dtype,
                               void* slab, size_t slab_bytes, int* nslab_out, float* bias_slab, int* bias_done_out,
                               hipStream_t st);
int rg_skinny_wgrad_impl(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                         int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, float* dbias = nullptr,
                         int bias_accumulate = 0, int* bias_done_out = nullptr);
