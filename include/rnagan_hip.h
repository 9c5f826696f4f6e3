/* rnagan_hip.h -- C ABI of librnagan_hip.so: the MI355X (gfx950) kernels behind the RNA-GAN
 * WGAN-GP training path.
 *
 * The reference (gevaertlab/RNA-GAN) has no native code; its hot path reaches native kernels only
 * through PyTorch ATen/cuDNN/cuBLAS calls made from Python.  Each entry point below replaces one
 * such implicit native op (SURVEY.md 2.2 "native-op inventory", K1..K15) and cites the reference
 * call site (paths relative to /root/reference) whose arithmetic it provides.  The Python side
 * (rna_gan_amd/_abi.py, ctypes) binds exactly these symbols; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library never
 *     allocates, frees or retains memory.  Workspace is passed in (rg_*_workspace_bytes).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *     Safe under HIP graph capture.  Re-entrant; no global mutable state except the thread-local
 *     error string.
 *   - return value: 0 = RG_OK, <0 = error (rg_last_error() gives the text).  Launch errors are
 *     detected with hipGetLastError() only.
 *   - activations: NHWC ([N][H][W][C], C fastest) in `dtype` = RG_F32 or RG_BF16; arithmetic and
 *     all per-channel statistics are fp32.  Image-side boundary tensors (discriminator input,
 *     generator output) are NCHW fp32 exactly as in PyTorch.
 *   - a 4x4 / stride-2 / pad-1 conv weight is w[O][I][4][4] fp32 (PyTorch memory layout), where
 *     O = channels on the LOW-resolution side and I = channels on the HIGH-resolution side: this is
 *     nn.Conv2d.weight (out,in,kh,kw) in the discriminator and nn.ConvTranspose2d.weight
 *     (in,out,kh,kw) in the generator, so "down"/"up" serve both networks and each other's backward.
 *   - `algo`: RG_ALGO_AUTO picks the MFMA implicit-GEMM kernels when dtype is bf16 and the shape
 *     is tile-aligned, else the generic tiled fp32 kernels; RG_ALGO_GENERIC / RG_ALGO_MFMA force one
 *     (MFMA returns RG_EUNSUPPORTED for shapes it cannot take).
 */
#ifndef RNAGAN_HIP_H
#define RNAGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RG_OK 0
#define RG_EINVAL (-1)
#define RG_EUNSUPPORTED (-2)
#define RG_EWORKSPACE (-3)
#define RG_EHIP (-4)

#define RG_F32 0
#define RG_BF16 1
#define RG_F16 2   /* IEEE fp16 storage: librnagan_hip_f16.so only (the same sources built with -DRG_HALF_F16; see "fp16 build" below) */

#define RG_ALGO_AUTO 0
#define RG_ALGO_GENERIC 1
#define RG_ALGO_MFMA 2

int rg_version(void);
const char* rg_last_error(void);
/* Kernel-selection knobs for A/B measurements inside one process (tools/, tests/): name = the part of the matching
 * environment variable after "RNAGAN_", lower case ("conv8", "conv8_blocks", "conv_tile", "xcd", "class_fast",
 * "wgrad_blocks", "wgrad8", ..., "f32mma": 0 sends the RG_F32 conv / dense launches back to the vector-ALU GEMM instead of the
 * f32 matrix-core one, 1 keeps every such launch on the f32 matrix instruction, 2 (the default) lets their 128 x 128-tile launches
 * form every fp32 product as six bf16 matrix-core products of the operands' exact three-way bf16 splits (fp32-grade accuracy,
 * ~11 % faster fp32 iteration); "wslab16": 0 keeps the deferred
 * weight-gradient slabs fp32; "bn_rev": which BatchNorm row passes walk their rows from the end (bit 2, the default: reductions); "convd": 0 sends the 64 -> 128 channel stride-2 conv back from the parity-plane-resident kernel to the
 * implicit-GEMM one, "convd_blocks": its persistent grid; "slab16": 0 keeps the split-K partial tiles of the conv launches fp32;
 * "skinny128": 0 sends the image-side layers of 256 x 256 images back
 * from the control-flow-free row kernels to the general row-staged ones; "wgrad8_mfma": 32 (default) / 16 = the matrix instruction
 * shape of the 256 x 256-tile weight-gradient kernel, bit-identical results; "narrow32": 0 sends 3 x 3 convs of <= 32 output columns
 * back to the kernel that issues MFMAs for all 64 columns of its tile; "upimg": 0 sends the resize-convolution generator's image
 * block back to the path that materialises its upsampled + padded input, "upimg_blocks": the fused kernel's grid).  An option set here
 * overrides the environment; value < 0 clears the override.  Returns RG_EINVAL for an
 * unknown name.  Not thread-safe against concurrent launches. */
int rg_set_option(const char* name, int value);

/* ---------------------------------------------------------------------------------------------
 * Stride-2 4x4 convolution family (K1/K2/K3 of SURVEY 2.2)
 * ------------------------------------------------------------------------------------------- */

/* Weight layout of this family: fp32 masters and gradients are TAP-MAJOR, w[O][4][4][I] (O = channels on the
 * low-resolution side, I = on the high-resolution side), i.e. nn.Conv2d's weight[O][I][kh][kw] /
 * nn.ConvTranspose2d's weight[O][I][kh][kw] with the dimensions permuted (0,2,3,1).  The host side keeps the
 * nn.Parameter as a strided view of this storage, so state_dicts and checkpoints still show the PyTorch shape.
 * With this order the bf16 GEMM operand of rg_conv_down is an element-wise cast of the master and rg_conv_wgrad
 * writes dw without a permuting pass.
 *
 * bf16 operand images for the MFMA kernels (either output may be NULL):
 *   wdn[O][16][I]  (B operand of rg_conv_down: k = (tap, i) contiguous in i) = cast of w
 *   wup[16][I][O]  (B operand of rg_conv_up:   k = (tap, o) contiguous in o) = transpose of w as [O][16*I]
 * both in `dtype`.  Runs after every optimizer step (src/wgan_loss.py:127,261,388). */
int rg_pack_conv_weight(const float* w, void* wdn, void* wup, int O, int I, int dtype, void* stream);
/* The same wup / G.0 operand images built from the bf16 copy of the masters that rg_adam_step_dev(shadow_bf16) keeps
 * (2-byte instead of 4-byte reads; w_bf16 is [O][16][I] resp. [E][C][16], the master's order). */
int rg_pack_conv_wup_from_bf16(const void* w_bf16, void* wup, int O, int I, void* stream);
/* the same for n <= 8 layers in ONE launch (arrays of n pointers / sizes on the host): every layer needs O % 64 == 0,
 * 16 * I % 128 == 0 and 16-byte aligned buffers, else RG_EUNSUPPORTED (use the single-layer call) */
int rg_pack_conv_wup_from_bf16_multi(int n, const void* const* w_bf16, void* const* wup, const int* O, const int* I, void* stream);
int rg_pack_g0_weight_from_bf16(const void* w_bf16, void* wp, int E, int C, void* stream);

/* y[N][Hi/2][Wi/2][O] = conv2d(x[N][Hi][Wi][I], w[O][4][4][I], stride 2, pad 1).
 * nn.Conv2d forward in the discriminator (D(.) at src/wgan_loss.py:119,241,253,379) and the
 * data-gradient of nn.ConvTranspose2d in the generator (loss.backward(), :126).
 * `wdn` may be NULL when the generic kernel runs (it reads `w`).
 * stats_partial (may be NULL): when the layer is followed by a train-mode BatchNorm, the MFMA epilogue also writes
 * per-tile column sums of y and y^2 (of the bf16 values as stored), fp32 [rows][2][O] with
 * rows = rg_conv_stats_rows(...) (0: this shape takes split-K or the generic kernel -- pass NULL and use
 * rg_bn_forward).  rg_bn_forward_partials finishes them: the separate statistics pass over y disappears. */
int rg_conv_stats_rows(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo);
int rg_conv_down(const void* x, const float* w, const void* wdn, void* y, int N, int Hi, int Wi, int I,
                 int O, float* stats_partial, int dtype, int algo, void* ws, size_t ws_bytes, void* stream);
/* workspace of rg_conv_down (up = 0) / rg_conv_up (up = 1): split-K partial slabs for layers whose tile
 * grid cannot fill the chip (few rows, long K); Hlow/Wlow = low-resolution side.  May be 0. */
size_t rg_conv_workspace_bytes(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo);

/* y[N][2Ho][2Wo][I] = conv_transpose2d(x[N][Ho][Wo][O], w[O][4][4][I], stride 2, pad 1).
 * nn.ConvTranspose2d forward in the generator (G(.) at src/wgan_loss.py:113,247,371) and the
 * data-gradient of nn.Conv2d in the discriminator (.backward() :126,260,387; autograd.grad :34-41).
 * mask_act (may be NULL): an activation tensor with y's shape and dtype; then y *= (mask_act > 0 ? 1 : mask_slope),
 * i.e. the LeakyReLU backward of the layer below is applied in the epilogue (the data-gradient of discriminator
 * layer 1 feeding layer 0's LeakyReLU) instead of in a separate pass over y. */
int rg_conv_up(const void* x, const float* w, const void* wup, void* y, int N, int Ho, int Wo, int O, int I,
               const void* mask_act, float mask_slope, float* stats_partial, int dtype, int algo, void* ws,
               size_t ws_bytes, void* stream);

/* dw[O][kh][kw][I] (+)= sum_{n,ho,wo} low[n][ho][wo][o] * high[n][2ho-1+kh][2wo-1+kw][i].
 * Weight gradient of both layer kinds (every .backward()).  Deterministic: split-K partial slabs
 * in `ws` are summed in a fixed order. */
size_t rg_conv_wgrad_workspace_bytes(int N, int Ho, int Wo, int O, int I, int dtype, int algo);
int rg_conv_wgrad(const void* low, const void* high, float* dw, int N, int Ho, int Wo, int O, int I, int dtype,
                  int accumulate, int algo, void* ws, size_t ws_bytes, void* stream);
/* Two contributions of the same layer in one launch: dw (+)= wgrad(low0, high0) + wgrad(low1, high1).
 * Used where the reference's autograd accumulates two backward passes into one .grad: D(real) + D(fake)
 * in the discriminator step (wgan_loss.py:241-260) and primal + tangent in the penalty step (:379-387).
 * Same workspace as rg_conv_wgrad. */
int rg_conv_wgrad2(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N,
                   int Ho, int Wo, int O, int I, int dtype, int accumulate, int algo, void* ws, size_t ws_bytes,
                   void* stream);
/* rg_conv_wgrad (low1 = high1 = NULL) / rg_conv_wgrad2 with the split-K reduction LEFT TO THE CALLER'S OPTIMIZER STEP
 * (.backward() directly followed by optimizer.step(), src/wgan_loss.py:126-127, :260-261, :387-388): a plan with
 * *nsplit_out > 1 leaves its fp32 partial slabs [nsplit][O][16][I] in `slab` (>= rg_conv_wgrad_workspace_bytes, caller-owned,
 * must stay untouched until rg_adam_step_slabs has consumed it; *slab_dtype_out says whether the partial tiles are fp32 or
 * bf16 -- option wslab16, default bf16: each partial sum rounded once, added in fp32) and does not write dw; *nsplit_out == 1: dw was written
 * (not accumulated) and nothing is pending -- also the outcome for every shape / dtype the matrix-core kernel does not take
 * (the generic kernel runs with `slab` as its workspace). */
int rg_conv_wgrad_slabs(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N, int Ho,
                        int Wo, int O, int I, int dtype, int algo, void* slab, size_t slab_bytes, int* nsplit_out,
                        int* slab_dtype_out, void* stream);
/* The same pair of calls (weight gradient, then optimizer.step()) as ONE launch for a layer whose plan has no split-K: the
 * gradient tile is never written -- the kernel's epilogue applies the Adam update (the arithmetic of rg_adam_step_dev, bit for
 * bit) to the tensor's fp32 master p, moments m, v and optional bf16 operand image, all tap-major [O][16][I] like dw.
 * hyper = the 8 constants rg_adam_hyper_dev wrote for THIS step.  26 bytes per parameter instead of 4 + 30.
 * rg_conv_wgrad_adam_supported: 1 when the shape / dtype has such a plan (bf16, 256 | O, 16 | I, no split-K at this pixel
 * count); the caller's streaming step then leaves the tensor out (rg_adam_step_slabs, seg_nsplit = -1). */
int rg_conv_wgrad_adam_supported(int N, int Ho, int Wo, int O, int I, int two, int dtype, int algo);
int rg_conv_wgrad_adam(const void* low0, const void* high0, const void* low1, const void* high1, float* p, float* m, float* v,
                       const float* hyper, void* shadow_bf16, int N, int Ho, int Wo, int O, int I, int dtype, int algo,
                       void* stream);

/* Image-side layers (I = 3 channels, NCHW fp32 on the high-resolution side; HBM-bound).  Their weights keep
 * the PyTorch layout w[O][I][4][4] (48 values per O).
 * rg_first_down: y[N][H/2][W/2][O] = lrelu_slope(conv2d(x_nchw, w) + bias); bias may be NULL,
 *   slope = 1 disables the activation.  Discriminator layer 0 forward
 *   (Conv2d(3,64,4,2,1)+LeakyReLU, histopathology_gan.py:186-192) and data-gradient of the
 *   generator's last ConvTranspose2d.
 * rg_last_up: y_nchw[N][I][2Ho][2Wo] = act(conv_transpose2d(x, w) + bias), act = tanh or none.
 *   Generator last layer forward (ConvTranspose2d(64,3,4,2,1)+Tanh, dcgan.py:82) and
 *   data-gradient of discriminator layer 0 (gives d/d xhat for the penalty, wgan_loss.py:34-41).
 * rg_skinny_wgrad: weight gradient of those two layers. */
int rg_first_down(const float* x_nchw, const float* w, const float* bias, void* y, int N, int H, int W, int I,
                  int O, float slope, int dtype, void* stream);
int rg_last_up(const void* x, const float* w, const float* bias, float* y_nchw, int N, int Ho, int Wo, int O,
               int I, int apply_tanh, int dtype, void* stream);
/* Packed LeakyReLU mask of the discriminator's layer 0 (histopathology_gan.py:186-192) for the data-gradient conv of
 * layer 1 (.backward() at src/wgan_loss.py:126,260,387 and autograd.grad :34-41): bits[pixel] (uint64, pixel = NHWC row of
 * y) has bit c set when y[pixel][c] > 0 (the bf16 value as stored).  rg_first_down_bits = rg_first_down + those bits
 * (O = 64, bf16; written by the same kernel); rg_sign_pack computes them from an existing activation [npix][64].
 * rg_conv_up_maskbits = rg_conv_up with mask_act given in this packed form (8 B instead of 128 B per output pixel, and the
 * kernel that keeps the input patch resident in LDS); rg_conv_up_maskbits_supported tells whether the shape takes it
 * (O = 128 -> I = 64 channels, Wo in {16, 32, 64}); otherwise use rg_conv_up with the activation itself. */
int rg_first_down_bits(const float* x_nchw, const float* w, const float* bias, void* y, void* bits, int N, int H, int W,
                       int I, int O, float slope, int dtype, void* stream);
int rg_sign_pack(const void* a, void* bits, long long npix, int C, int dtype, void* stream);
/* y = lrelu'(a) * conv2d(x_nchw, w) (no bias), lrelu'(a) taken from a's packed sign bits: the tangent of discriminator
 * layer 0 in the penalty's forward-mode pass (second-order term of src/wgan_loss.py:32-44) in one kernel.
 * rg_first_down_masked_supported == 0: use rg_first_down(slope 1) + rg_lrelu_bwd instead. */
int rg_first_down_masked_supported(int H, int W, int I, int O, int dtype);
int rg_first_down_masked(const float* x_nchw, const float* w, void* y, const void* mask_bits, float mask_slope, int N,
                         int H, int W, int I, int O, int dtype, void* stream);
int rg_conv_up_maskbits_supported(int N, int Ho, int Wo, int O, int I, int dtype, int algo);
int rg_conv_up_maskbits(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I,
                        const void* mask_bits, float mask_slope, int dtype, int algo, void* ws, size_t ws_bytes,
                        void* stream);
/* rg_last_up (no bias, no activation) with the FIRST CONSUMER's pass over its fp32 NCHW output fused into the store phase:
 *   tanh_img (optional, shape of y): y *= 1 - tanh_img^2 -- in the generator-loss step the data gradient of the
 *     discriminator's layer 0 is the cotangent of the generator's Tanh output (.backward() at src/wgan_loss.py:126 through
 *     nn.Tanh, src/dcgan.py:82): rg_tanh_bwd's arithmetic without materialising the unmultiplied gradient;
 *   part (optional): float[rg_last_up_post_blocks(...)][4], one row per workgroup = sums over the rows it wrote of channel
 *     0, 1, 2 of y and of y^2.  rg_last_up_part_chan_sum adds the rows up into the bias gradient of the generator's last
 *     ConvTranspose2d (out3[c] (+)= sum); rg_gp_coef_parts into the penalty's squared norm ||d D(xhat) / d xhat||^2 over the
 *     WHOLE batch (src/wgan_loss.py:34-43: gradients.norm(2) of the flattened (N, -1) tensor), from which it writes loss =
 *     (norm - 1)^2 and coef = lambd * 2 (norm - 1) / norm as rg_gp_coef does (sq, optional, receives the squared norm).
 *   Fixed summation order: deterministic.  rg_last_up_post_blocks == 0: no such kernel for the shape (use rg_last_up +
 *   rg_tanh_bwd / rg_nchw_chan_sum / rg_sqnorm). */
int rg_last_up_post_blocks(int N, int Ho, int Wo, int O, int I, int dtype);
int rg_last_up_post(const void* x, const float* w, float* y_nchw, int N, int Ho, int Wo, int O, int I, int dtype,
                    const float* tanh_img, float* part, void* stream);
int rg_last_up_part_chan_sum(const float* part, int nblocks, float* out3, int accumulate, void* stream);
int rg_gp_coef_parts(const float* part, int nblocks, float* sq, float* loss, float* coef, float lambd, void* stream);
/* the same when the gradient whose squared norm the partials hold carries a loss scale (fp16 build): norm = sqrt(sum) / in_scale,
 * loss from that norm, coef = lambd * 2 (norm - 1) / norm * out_scale / in_scale -- the tangent direction coef * g' then carries
 * out_scale.  Scales are powers of two (exact); with in_scale = out_scale = 1 this is rg_gp_coef_parts bit for bit. */
int rg_gp_coef_parts_scaled(const float* part, int nblocks, float* sq, float* loss, float* coef, float lambd, float in_scale,
                            float out_scale, void* stream);
/* rg_last_up with the generator's last train-mode BatchNorm + LeakyReLU applied to its input on the fly (z = the pre-BatchNorm
 * conv output; mean / invstd from rg_bn_finalize_partials or rg_bn_stats_finalize): the no-grad generator forwards of the
 * D-loss and penalty steps (src/wgan_loss.py:247,371) skip the normalisation pass over their largest activation.  Same bf16
 * rounding as rg_bn_act followed by rg_last_up.  rg_last_up_pre_supported == 0: use those two. */
int rg_last_up_pre_supported(int Wo, int O, int I, int dtype);
int rg_last_up_pre(const void* z, const float* w, const float* bias, float* y_nchw, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, float slope, int N, int Ho, int Wo, int O, int I, int apply_tanh,
                   int dtype, void* stream);
size_t rg_skinny_wgrad_workspace_bytes(int N, int Ho, int Wo, int O, int I);
int rg_skinny_wgrad(const void* low, const float* high_nchw, float* dw, int N, int Ho, int Wo, int O, int I,
                    int dtype, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* rg_skinny_wgrad with the reduction of its per-workgroup partial gradients LEFT to the optimizer step that follows at once
 * (src/wgan_loss.py:126-127, :260-261, :387-388): `slab` receives *nslab_out fp32 slabs of O * 48 elements in dw's layout
 * (at most rg_skinny_wgrad_workspace_bytes), to be handed to rg_adam_step_slabs as that tensor's segment; a second contribution
 * to the same tensor (D(real) + D(fake), primal + tangent) is a second call with `slab` advanced by the first call's slabs.
 * *nslab_out = 0: this shape / dtype has no such form and nothing was launched. */
int rg_skinny_wgrad_slabs(const void* low, const float* high_nchw, int N, int Ho, int Wo, int O, int I, int dtype, void* slab,
                          size_t slab_bytes, int* nslab_out, float* bias_slab, int* bias_done_out, void* stream);
/* rg_skinny_wgrad that also forms the layer's bias gradient dbias[O] = sum over the pixels of `low` (what autograd adds to
 * Conv2d(3, 64).bias.grad, histopathology_gan.py:186-192) as a by-product of its pass over `low` where the kernel can (a column of
 * ones in the patch operand of the 256 x 256 bf16 row kernel): *bias_done_out = 1; otherwise 0 and dbias is untouched (rg_col_sum
 * then).  bias_accumulate: add to dbias.  rg_skinny_wgrad_slabs' bias_slab / bias_done_out: the same by-product as partials
 * [*nslab_out][O] for the optimizer step (a segment of rg_adam_step_slabs). */
int rg_skinny_wgrad_bias(const void* low, const float* high_nchw, float* dw, float* dbias, int N, int Ho, int Wo, int O, int I,
                         int dtype, int accumulate, int bias_accumulate, void* ws, size_t ws_bytes, int* bias_done_out,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers: generator layer 0 (K4), discriminator head (K5), betaVAE encoder (K8)
 * ------------------------------------------------------------------------------------------- */

/* Generator layer 0, ConvTranspose2d(E,C,4,1,0) on a 1x1 input (dcgan.py:38-40):
 *   y[N][4][4][C] = sum_e z[n][e] * w[e][c][kh][kw].   w is [E][C][4][4] fp32;
 *   wp = packed [16*C][E] in dtype (rg_pack_g0_weight), NULL for the generic kernel. */
int rg_pack_g0_weight(const float* w, void* wp, int E, int C, int dtype, void* stream);
int rg_g0_fwd(const float* z, const float* w, const void* wp, void* y, int N, int E, int C, int dtype, int algo,
              void* ws, size_t ws_bytes, void* stream);
size_t rg_g0_workspace_bytes(int N, int E, int C, int dtype, int algo);
int rg_g0_wgrad(const float* z, const void* gy, float* dw, int N, int E, int C, int dtype, int accumulate,
                int algo, void* ws, size_t ws_bytes, void* stream);

/* Discriminator head, Conv2d(C,1,4,1,0)+LeakyReLU on a 4x4 map -> (N,):
 *   h[n] = sum a[n][kh][kw][c] * w[0][c][kh][kw];  out[n] = lrelu(h[n]). */
int rg_head_fwd(const void* a, const float* w, float* h, float* out, int N, int C, float slope, int dtype,
                void* stream);
/* gh[n] = coef * lrelu'(h[n])  (coef = d loss / d out[n]: -1/N, +1/N or 1; wgan_loss.py:24-29,33) */
int rg_head_grad(const float* h, float* gh, int N, float coef, float slope, void* stream);
int rg_head_bwd_data(const float* gh, const float* w, void* ga, int N, int C, int dtype, void* stream);
int rg_head_wgrad(const float* gh, const void* a, float* dw, int N, int C, int dtype, int accumulate,
                  void* stream);

/* y[M][ldy] = act((x[M][K] . w[Nout][K]^T) * scale[j] + shift[j]) : Linear + BatchNorm1d(eval)
 * folded + LeakyReLU(0.01) of the betaVAE encoder, and z_mu (betaVAE.py:26-42,102-107).
 * x is fp32 [M][ldx]; w fp32 [Nout][K] (PyTorch Linear layout); y fp32. slope=1: no activation.
 * wp: optional bf16 copy of w padded to [Nout_pad][K_pad] (rg_pack_linear_weight) for the MFMA
 * kernel (weight streaming is the bound: 2 B/weight instead of 4). */
int rg_pack_linear_weight(const float* w, void* wp, int Nout, int K, int Nout_pad, int K_pad, void* stream);
size_t rg_linear_workspace_bytes(int M, int K, int Nout, int algo);
int rg_linear_affine_act(const float* x, int ldx, const float* w, const void* wp, const float* scale,
                         const float* shift, float* y, int ldy, int M, int K, int Nout, float slope, int algo,
                         void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d in TRAIN mode + LeakyReLU: forward, backward, forward-mode tangent and the joint
 * (double) backward needed by the gradient penalty (K6/K7).  z, a, g* are [M][C] views of NHWC
 * tensors (M = N*H*W).  All column reductions are two-stage and deterministic; `ws` must hold
 * rg_colreduce_workspace_bytes(M, C, nq) bytes (nq = number of sums, <= 3).
 * ------------------------------------------------------------------------------------------- */
size_t rg_colreduce_workspace_bytes(int M, int C, int nq);

/* sum[c] = sum_m z, sumsq[c] = sum_m z^2 */
int rg_bn_stats(const void* z, float* sum, float* sumsq, int M, int C, int dtype, void* ws, size_t ws_bytes,
                void* stream);
/* mean, invstd = 1/sqrt(biased var + eps); if running_mean != NULL also the PyTorch running-stat
 * update (momentum, unbiased variance) and ++(*num_batches_tracked) (int64). */
int rg_bn_finalize(const float* sum, const float* sumsq, int M, int C, float eps, float momentum, float* mean,
                   float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                   void* stream);
/* rg_bn_stats + rg_bn_finalize in one pass: the finishing step of the column reduction writes mean / invstd and
 * updates the running statistics directly (one launch less per BatchNorm forward; same arithmetic). */
/* The whole train-mode BatchNorm2d + LeakyReLU forward: a = lrelu(gamma * (z - mean) * invstd + beta) with the batch
 * statistics of z (written to mean / invstd, running statistics updated when given): statistics pass (whose finishing
 * step does the finalize) + pointwise pass.  RNAGAN_BN_FUSED=1 selects a single-launch form for small tensors in this
 * and in rg_bn_act_bwd / rg_bn_tangent / rg_bn_double_bwd (measured slower on MI355X, kept for experiments). */
int rg_bn_forward(const void* z, int M, int C, float eps, float momentum, const float* gamma, const float* beta,
                  float slope, float* mean, float* invstd, float* running_mean, float* running_var,
                  int64_t* num_batches_tracked, void* a, int dtype, void* ws, size_t ws_bytes, void* stream);
/* rg_bn_forward with the statistics taken from the partial sums a conv epilogue wrote (rg_conv_down / rg_conv_up
 * stats_partial, G rows): finishing kernel + pointwise pass. */
int rg_bn_forward_partials(const float* partial, int G, const void* z, int M, int C, float eps, float momentum,
                           const float* gamma, const float* beta, float slope, float* mean, float* invstd,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, void* a, int dtype,
                           void* ws, size_t ws_bytes, void* stream);     /* ws: 32 * 2 * C floats (two-level finish) */
int rg_bn_stats_finalize(const void* z, int M, int C, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, int dtype, void* ws,
                         size_t ws_bytes, void* stream);
/* a = lrelu((z-mean)*invstd*gamma + beta) */
int rg_bn_act(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
              void* a, int M, int C, float slope, int dtype, void* stream);
/* gy = ga*lrelu'(y); s_gy = sum gy; s_gyxh = sum gy*xhat; gz = gamma*invstd*(gy - s_gy/M - xhat*s_gyxh/M);
 * dgamma (+)= s_gyxh, dbeta (+)= s_gy when dgamma != NULL. */
int rg_bn_act_bwd(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, void* gz, float* s_gy, float* s_gyxh, float* dgamma, float* dbeta,
                  int accumulate, int M, int C, float slope, int dtype, void* ws, size_t ws_bytes, void* stream);
/* mean / invstd (+ running statistics) from the conv epilogue's column sums, without the normalisation pass. */
int rg_bn_finalize_partials(const float* partial, int G, int M, int C, float eps, float momentum, float* mean,
                            float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, void* ws,
                            size_t ws_bytes, void* stream);
/* ---- generator layer 0: weight gradient + optimizer step in one pass --------------------------------------------------
 * rg_g0_wgrad (.backward() into nn.ConvTranspose2d(E, C, 4, 1, 0).weight, src/wgan_loss.py:126) followed by the Adam step
 * on that tensor (optimizer_generator.step(), :127) as ONE streaming kernel: the gradient dw[e][c][tap] = sum_n z[n][e] *
 * gz0[n][tap][c] (K = the batch) is formed in registers and torch.optim.Adam's update applied in place to p / m / v
 * ([E][C][4][4] fp32, 16-byte aligned), `hyper` = the 8 floats of rg_adam_hyper_dev, shadow_bf16 (may be NULL) = the bf16
 * image of p that rg_adam_step_dev keeps.  26 B per parameter instead of 4 (gradient written) + 30.  Single-process runs
 * only: a data-parallel step needs the gradient in memory for its all-reduce.  dtype = element type of gz0. */
int rg_g0_wgrad_adam_supported(int N, int E, int C, int dtype);
int rg_g0_wgrad_adam(const float* z, const void* gz0, float* p, float* m, float* v, const float* hyper, void* shadow_bf16,
                     int N, int E, int C, int dtype, void* stream);

/* The same fusion for an nn.Linear weight W[O][I] (fp32, row pitch I): dW[o][i] = sum_n g[n][o] * x[n][i], the batch being the
 * only contraction, formed on MFMA and consumed by torch.optim.Adam's update (weight decay included) in one pass over p / m / v
 * -- the betaVAE's layers in src/betaVAE_training.py (optimizer.step() at src/betaVAE.py:226 after loss.backward() :225).
 * gT [>= O][ldn] and xT [>= I][ldn]: bf16, samples contiguous, zero padded to ldn (a multiple of 64) -- the images
 * rg_transpose_pack_bf16 writes.  p / m / v: the weight's segment of the flat parameter / moment buffers; hyper: rg_adam_hyper_dev.
 * The gradient itself is not written anywhere.  wpack_bf16 (optional): the bf16 operand image [>= O][Kp] of the UPDATED weight in
 * rg_pack_linear_weight's layout (its zero padding is the caller's: only the O x I entries are written), so that the next
 * forward needs no packing pass. */
int rg_linear_wgrad_adam(const void* gT, const void* xT, int ldn, int N, float* p, float* m, float* v, const float* hyper, int O,
                         int I, void* wpack_bf16, int Kp, void* stream);

/* ---- Inception-v3 feature extractor of the FID metric (src/fid.py:33-94: torchvision inception_v3 up to Mixed_7c) --------
 * NHWC fp32.  A BasicConv2d (Conv2d(bias=False) + BatchNorm2d(eval, eps 1e-3) + ReLU) = rg_im2col_nhwc (not needed for 1x1
 * stride 1) + rg_linear_affine_act with the folded BatchNorm affine and slope 0; ldx / ldy are row strides in elements, so a
 * branch reads / writes a channel slice of a wider activation (torch.cat of the block outputs is free).
 *   rg_im2col_nhwc        cols[(n,ho,wo)][(i,j,c)] = x[n][ho*sh-ph+i][wo*sw-pw+j][c] or 0 outside the image
 *   rg_pool2d_nhwc        mode 0: F.max_pool2d(k, stride) ; mode 1: F.avg_pool2d(k, stride, pad) (padding counted)
 *   rg_nchw_to_nhwc_affine y[n][p][c] = x[n][c][p] * scale[c] + shift[c]  (x * 2 - 1 and torchvision's transform_input)
 *   rg_spatial_mean_nhwc  adaptive_avg_pool2d(., (1, 1)) */
int rg_im2col_nhwc(const float* x, int ldx, float* cols, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                   int pw, void* stream);
int rg_pool2d_nhwc(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int k, int stride, int pad, int mode,
                   void* stream);
int rg_nchw_to_nhwc_affine(const float* x_nchw, float* y_nhwc, int N, int C, int H, int W, const float* scale, const float* shift,
                           void* stream);
int rg_spatial_mean_nhwc(const float* x, float* y, int N, int HW, int C, void* stream);

/* ---- data-gradient conv + the BatchNorm backward it feeds, without the reduction pass (bf16 path) ------------------------
 * In .backward() through a [Conv -> BatchNorm2d(train) -> LeakyReLU] stack the data gradient ga produced by one layer's
 * conv (rg_conv_up for nn.Conv2d, rg_conv_down for nn.ConvTranspose2d) is consumed by the BatchNorm backward of the layer
 * below, whose first pass reduces gy = ga * lrelu'(y) and gy * xhat over all rows.  rg_conv_*_bnbwd compute these column
 * sums in the conv's epilogue (the tile is in registers; the consumer's z tile is read there) and write one partial row
 * [2][C] per block tile into sums_partial (rg_conv_bnbwd_rows(...) rows; 0: this shape splits K or runs another kernel --
 * use rg_conv_* + rg_bn_act_bwd).  rg_bn_act_bwd_partials = rg_bn_act_bwd[_g2] from those partial rows: finisher + pointwise
 * pass.  groups = 2: two batch halves with their own statistics (mean / invstd [2][C]); G = partial rows per group,
 * nblk = 1 (rg_conv_down_bnbwd) or 4 (rg_conv_up_bnbwd: rows are parity-class-major). */
int rg_conv_bnbwd_rows(int up, int N, int Hlow, int Wlow, int O, int I, int groups, int dtype, int algo);
int rg_conv_down_bnbwd(const void* x, const void* wdn, void* y, int N, int Hi, int Wi, int I, int O, const void* z_next,
                       const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, int groups,
                       float* sums_partial, int dtype, int algo, void* ws, size_t ws_bytes, void* stream);
int rg_conv_up_bnbwd(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I, const void* z_next,
                     const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, int groups,
                     float* sums_partial, int dtype, int algo, void* ws, size_t ws_bytes, void* stream);
int rg_bn_act_bwd_partials(const float* partial, int G, int nblk, const void* z, const void* ga, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, void* gz, float* s_gy, float* s_gyxh,
                           float* dgamma, float* dbeta, int accumulate, int M, int C, int groups, float slope, int dtype,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- split-K conv + train-mode BatchNorm without the intermediate passes (bf16 path) -------------------------------
 * The deep Conv2d / ConvTranspose2d layers at small batch run split-K (rg_conv_split(...) > 1): every launch leaves
 * nsplit fp32 slabs [nsplit][rows][C] in the workspace.  rg_conv_down_partial / rg_conv_up_partial run ONLY that launch
 * (slabs in `ws`, slab s at ws + s * rows * C floats, rows in output NHWC order); the consumer sums them:
 *   rg_bn_forward_slabs : z = bf16(sum_s slab_s) (written), batch statistics of z (per batch group, running statistics
 *                         updated group after group), a = lrelu(BN(z)) -- nn.Conv2d -> nn.BatchNorm2d(train) ->
 *                         nn.LeakyReLU of the torchgan DCGAN blocks, = rg_conv_* + rg_bn_forward[_g2];
 *   rg_bn_act_bwd_slabs : ga = bf16(sum_s slab_s) = the data gradient arriving at the block (written when ga_out != NULL),
 *                         then exactly rg_bn_act_bwd[_g2] (.backward() of the same block).
 * One launch each: the workgroups of a 128-column slice exchange their partial column sums through `scratch` and meet at
 * a counter in `sync` (agent-scope release/acquire hand-off, deterministic summation order).  `sync`: a caller-owned
 * buffer of rg_slab_bn_sync_words() 32-bit words, ZEROED ONCE when allocated and private to one stream (every launch
 * leaves it zero); `scratch`: rg_slab_bn_scratch_bytes(M, C, groups) bytes, contents irrelevant.  M = rows per batch group
 * (groups = 1 or 2, z is [groups * M][C]).  rg_slab_bn_supported: C % 128 == 0, nsplit in {2, 4, 8}, row count divisible
 * into 64 / 128 / 256-row blocks with at most 256 workgroups (one per CU: all co-resident -- required by the hand-off). */
int rg_conv_split(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo);
/* element type of the slabs that launch leaves: RG_F32, or RG_BF16 where the 8-wave kernel stores its partial tiles as bf16
 * (option "slab16", default on: half the slab bytes written and re-read; each partial sum is rounded to bf16 before the
 * consumer adds them in fp32).  Pass it to the consumer as slab_dtype; the slab stride stays in ELEMENTS. */
int rg_conv_slab_dtype(int up, int N, int Hlow, int Wlow, int O, int I, int dtype, int algo);
int rg_conv_down_partial(const void* x, const void* wdn, int N, int Hi, int Wi, int I, int O, int dtype, int algo,
                         void* ws, size_t ws_bytes, void* stream);
int rg_conv_up_partial(const void* x, const void* wup, int N, int Ho, int Wo, int O, int I, int dtype, int algo,
                       void* ws, size_t ws_bytes, void* stream);
int rg_slab_bn_supported(long long M, int C, int groups, int nsplit);
size_t rg_slab_bn_scratch_bytes(long long M, int C, int groups);
size_t rg_slab_bn_sync_words(void);
int rg_bn_forward_slabs(const void* slab, int nsplit, size_t slab_stride, int slab_dtype, void* z, void* a, long long M, int C, int groups,
                        float eps, float momentum, const float* gamma, const float* beta, float slope, float* mean,
                        float* invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                        void* scratch, size_t scratch_bytes, void* sync, void* stream);
/* the forward-mode tangent of the same block (rg_bn_tangent) with zt arriving as slabs (the penalty's tangent forward):
 * zt_out = bf16(sum_s slab_s) (always written), at, s_zt, s_xhzt as rg_bn_tangent */
int rg_bn_tangent_slabs(const void* slab, int nsplit, size_t slab_stride, int slab_dtype, const void* z, void* zt_out, void* at, long long M,
                        int C, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                        float* s_zt, float* s_xhzt, void* scratch, size_t scratch_bytes, void* sync, void* stream);
int rg_bn_act_bwd_slabs(const void* slab, int nsplit, size_t slab_stride, int slab_dtype, const void* z, void* ga_out, void* gz, long long M,
                        int C, int groups, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        float slope, float* s_gy, float* s_gyxh, float* dgamma, float* dbeta, int accumulate, void* scratch,
                        size_t scratch_bytes, void* sync, void* stream);

/* Two batch groups in one call (the D-loss step runs D(real) and D(fake) -- src/wgan_loss.py:241-253 -- as one double
 * batch through the conv layers; BatchNorm must treat the halves as the two separate forward calls they are in the
 * reference): z / a / ga / gz are [2*M][C] (first half first), mean / invstd / s_gy / s_gyxh [2][C].  rg_bn_forward_g2 =
 * rg_bn_forward_partials (partial != NULL: 2*G rows [.][2][C] laid out [nblk][2 halves][G/nblk] -- nblk = 1 for
 * rg_conv_down's row-tile order, 4 for rg_conv_up's class-major order) or rg_bn_forward (partial == NULL) on the first half,
 * then on the second (running statistics and num_batches_tracked updated in that order); rg_bn_act_bwd_g2 = rg_bn_act_bwd on both
 * halves with dgamma / dbeta summed.  Workspace: twice rg_colreduce_workspace_bytes(M, C, 2). */
int rg_bn_finalize_partials_g2(const float* partial, int G, int nblk, int M, int C, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               void* ws, size_t ws_bytes, void* stream);
int rg_bn_forward_g2(const float* partial, int G, int nblk, const void* z, int M, int C, float eps, float momentum,
                     const float* gamma, const float* beta, float slope, float* mean, float* invstd, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, void* a, int dtype, void* ws, size_t ws_bytes,
                     void* stream);
int rg_bn_act_bwd_g2(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, void* gz, float* s_gy, float* s_gyxh, float* dgamma, float* dbeta, int accumulate,
                     int M, int C, float slope, int dtype, void* ws, size_t ws_bytes, void* stream);
/* tangent of a = lrelu(bn(z)) along zt with batch statistics varying:
 * at = lrelu'(y)*gamma*invstd*(zt - s_zt/M - xhat*s_xhzt/M) */
int rg_bn_tangent(const void* z, const void* zt, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, void* at, float* s_zt, float* s_xhzt, int M, int C, float slope, int dtype,
                  void* ws, size_t ws_bytes, void* stream);
/* joint reverse through (a, at): see DESIGN.md "GP second-order pass" for the formula.
 * qa may be NULL (zero primal cotangent). */
int rg_bn_double_bwd(const void* z, const void* qa, const void* zt, const void* ga1, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, const float* s_gy,
                     const float* s_gyxh, const float* s_zt, const float* s_xhzt, void* pz, float* dgamma,
                     float* dbeta, int accumulate, int M, int C, float slope, int dtype, void* ws,
                     size_t ws_bytes, void* stream);

/* out = g * lrelu'(a) with the mask taken from the sign of the activation OUTPUT a */
int rg_lrelu_bwd(const void* g, const void* a, void* out, size_t n, float slope, int dtype, void* stream);
/* out[c] (+)= sum_m g[m][c]   (bias gradient) */
int rg_col_sum(const void* g, float* out, int M, int C, int dtype, int accumulate, void* ws, size_t ws_bytes,
               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image-side pointwise ops and reductions (NCHW fp32) (K7, K10, K11)
 * ------------------------------------------------------------------------------------------- */
int rg_tanh_bwd(const float* gy, const float* y, float* gz, size_t n, void* stream);
/* out[c] (+)= sum_{n,h,w} g[n][c][h][w] */
int rg_nchw_chan_sum(const float* g, float* out, int N, int C, int HW, int accumulate, void* ws, size_t ws_bytes,
                     void* stream);
/* xhat = eps*real + (1-eps)*fake (wgan_loss.py:377) */
int rg_interp(const float* real, const float* fake, float* out, size_t n, float eps, void* stream);
/* graph-replayable variant: eps read from device memory */
int rg_interp_dev(const float* real, const float* fake, float* out, size_t n, const float* eps, void* stream);
/* out[0] = sum x^2 (fp32 result, pairwise/blocked accumulation; deterministic) */
size_t rg_reduce_workspace_bytes(size_t n);
int rg_sqnorm(const float* x, float* out, size_t n, void* ws, size_t ws_bytes, void* stream);
/* from sq = ||g||^2: loss = (sqrt(sq)-1)^2 ; coef = lambd*2*(sqrt(sq)-1)/sqrt(sq) (wgan_loss.py:43) */
int rg_gp_coef(const float* sq, float* loss, float* coef, float lambd, void* stream);
int rg_gp_coef_scaled(const float* sq, float* loss, float* coef, float lambd, float in_scale, float out_scale, void* stream);
/* out = x * coef[0] (coef on device: no host sync inside the step) */
int rg_scale_by(const float* x, const float* coef, float* out, size_t n, void* stream);
/* out[0] = sign * mean(a - b) (b may be NULL) (wgan_loss.py:24-29) */
int rg_mean_diff(const float* a, const float* b, float* out, int n, float sign, void* stream);
/* noise = (u+z - mean_col)/std_col, unbiased std over the batch (wgan_loss.py:105-106) */
int rg_latent_prep(const float* u, const float* z, float* out, int N, int E, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer (K12/K13): torch.optim.Adam semantics on flat fp32 buffers
 * (histopathology_gan.py:252,257; stepped at wgan_loss.py:127,261,388), optional weight clamp
 * (wgan_loss.py:213-215).  step is 1-based.  Hyper-parameters are doubles, as torch holds them: the
 * kernel's constants (1-beta, lr/bias_correction1, 1/sqrt(bias_correction2)) are formed in double and
 * rounded once, which reproduces torch's fp32 results to the last bits.
 * ------------------------------------------------------------------------------------------- */
int rg_adam_step(float* p, const float* g, float* m, float* v, size_t n, int step, double lr, double beta1,
                 double beta2, double eps, void* stream);
int rg_clamp(float* p, size_t n, float lo, float hi, void* stream);
/* Same update with the step-dependent constants read from DEVICE memory, so the launch can live in
 * a captured HIP graph and be replayed every step: hyper[0..7] = {beta1, beta2, 1-beta1, 1-beta2, eps,
 * lr/bias_correction1, 1/sqrt(bias_correction2), weight_decay}, produced by rg_adam_hyper_dev.
 * shadow_bf16 (may be NULL): bf16[n], receives the rounded updated parameters in the same launch -- for the
 * tap-major conv weights that IS the wdn operand of rg_conv_down, so no separate pack pass reads the masters.
 * grad_bf16 (may be NULL): bf16[n] gradient read INSTEAD of g -- the all-reduced bf16 wire buffer of a data-parallel
 * run, which then needs no widening pass back to fp32. */
int rg_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, void* shadow_bf16,
                     const void* grad_bf16, void* stream);
/* rg_adam_step_dev over [p, p + n) cut into nseg <= 24 consecutive segments (seg_off / seg_n in elements, tiling the range in
 * order, every offset a multiple of 4): a segment with seg_slab[i] != NULL takes its gradient as the sum, in slab order, of
 * seg_nsplit[i] slabs of seg_n[i] elements, fp32 or bf16 by seg_dtype[i] (what rg_conv_wgrad_slabs left in the caller's buffer
 * and reported) instead of reading g --
 * the split-K reduction launches of the backward pass and the write + re-read of the reduced gradient disappear (one launch
 * for the whole buffer; deterministic summation order).  A segment with seg_slab[i] == NULL and seg_nsplit[i] == -1 is SKIPPED
 * (its tensor is stepped by rg_conv_wgrad_adam).  Segment tables are HOST arrays. */
int rg_adam_step_slabs(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, void* shadow_bf16, int nseg,
                       const unsigned long long* seg_off, const unsigned long long* seg_n, const void* const* seg_slab,
                       const int* seg_nsplit, const int* seg_dtype, void* stream);
/* Data parallel (one process per GPU, bf16 gradient wire; the reference has no distributed path -- SURVEY 8 e): the flat fp32
 * gradient g[n] goes onto the bf16 wire buffer of the all-reduce with the segment table of rg_adam_step_slabs: a plain segment is
 * rounded, a slab segment is summed (slab order, fp32) and rounded once, a segment with seg_nsplit = -1 is left alone (its
 * weight-gradient launch wrote it: rg_conv_wgrad_wire).  Replaces the per-layer reduction launches + one cast pass. */
int rg_grad_to_wire(const float* g, void* wire_bf16, size_t n, int nseg, const unsigned long long* seg_off,
                    const unsigned long long* seg_n, const void* const* seg_slab, const int* seg_nsplit, const int* seg_dtype,
                    void* stream);
/* rg_conv_wgrad / rg_conv_wgrad2 (low1 = high1 = NULL: one segment) of a layer whose plan has no split-K
 * (rg_conv_wgrad_adam_supported) written as bf16 into the tensor's 16-byte aligned slice of that wire buffer. */
int rg_conv_wgrad_wire(const void* low0, const void* high0, const void* low1, const void* high1, void* wire_bf16, int N, int Ho,
                       int Wo, int O, int I, int dtype, int algo, void* stream);
/* ++(*step_dev) and recompute hyper[0..6] from it on the device (double arithmetic, one thread): with
 * this launch in front of rg_adam_step_dev the whole optimizer step replays from a graph untouched.
 * hyper[7] = weight_decay (torch.optim.Adam's L2 term g += wd * p; 0 on the GAN path, betaVAE training sets it,
 * src/betaVAE_training.py:163). */
int rg_adam_hyper_dev(int* step_dev, double lr, double beta1, double beta2, double eps, double weight_decay, float* hyper,
                      void* stream);
/* ... and hyper[8] = grad_scale_inv: every Adam kernel multiplies the gradient it reads (fp32 gradient, slab sums, wire, the tile
 * of the fused weight-gradient launches) by it before the update -- the unscale of a loss-scaled backward pass (fp16 build:
 * the backward seeds carry a static power-of-two scale, the mechanism sketched at src/betaVAE.py:184,230-236).  `hyper` is
 * therefore >= 9 floats for BOTH entry points; rg_adam_hyper_dev writes hyper[8] = 1 (exact: results unchanged). */
int rg_adam_hyper_dev2(int* step_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                       double grad_scale_inv, float* hyper, void* stream);
/* The 16-bit storage type of this build of the library: RG_BF16 (librnagan_hip.so) or RG_F16 (librnagan_hip_f16.so: the same
 * sources compiled with -DRG_HALF_F16 -- activations, operand images, split-K slabs and the data-parallel wire are IEEE fp16,
 * the MFMAs the _f16 forms; every entry point takes RG_F16 where this header says RG_BF16; no rg_probe_* entry points).
 * BASELINE.json configs[3] names fp16 storage. */
int rg_storage_dtype(void);

/* ---------------------------------------------------------------------------------------------
 * betaVAE TRAINING (src/betaVAE.py:63-107 model, :145-163 loss, :166-284 train loop).
 * A Linear layer's three GEMMs all take the "NT" form C[M][Nout] = A[M][K] . B[Nout][K]^T:
 *   forward  y  = x  . W^T      A = x [N][in]        B = W   [out][in]
 *   data     dx = dy . W        A = dy [N][out]      B = W^T [in][out]     (rg_transpose_pack_bf16 of W)
 *   weight   dW = dy^T . x      A = dy^T [out][N]    B = x^T [in][N]       (both transposed packs; K = batch)
 * rg_gemm_nt_bf16: bf16 operands with K padded to K_pad (multiple of 64, zero filled), fp32 accumulate/output,
 *   epilogue y = lrelu(acc * scale[j] + shift[j], slope) (scale/shift may be NULL, slope 1 = none).  Nout need not
 *   be a multiple of 8; the columns up to the next multiple of 8 (when inside ldy) are zero-filled.
 * rg_transpose_pack_bf16: dst bf16 [C_pad][R_pad] = src^T (fp32 [R][C]), zero padded; R_pad % 64 == 0.
 * rg_transpose_f32: fp32 parity mode (the functor GEMM rg_linear_affine_act takes fp32 operands).
 * ------------------------------------------------------------------------------------------- */
int rg_transpose_f32(const float* src, float* dst, int R, int C, void* stream);
int rg_transpose_pack_bf16(const float* src, void* dst, int R, int C, int R_pad, int C_pad, void* stream);
size_t rg_gemm_nt_bf16_workspace_bytes(int M, int K_pad, int Nout);
int rg_gemm_nt_bf16(const void* a, const void* b, const float* scale, const float* shift, float* y, int ldy, int M,
                    int K_pad, int Nout, float slope, void* ws, size_t ws_bytes, void* stream);
/* y[N][ld] = Dropout(x[N][F]) with the caller's keep-mask (uint8, NULL = keep all) and scale 1/(1-p); columns
 * F..ld-1 are zero (src/betaVAE.py:27: nn.Dropout() in front of the encoder) */
int rg_vae_dropout(const float* x, const unsigned char* mask, float* y, int N, int F, int ld, float scale, void* stream);
/* z = mu + eps * exp(0.5 logvar) (betaVAE.py:96-100) and its backward joined with the loss gradients:
 * gmu = gmu_loss + gz ; glv = glv_loss + gz * eps * 0.5 exp(0.5 logvar) */
int rg_vae_reparam(const float* mu, const float* logvar, const float* eps, float* z, size_t n, void* stream);
int rg_vae_reparam_bwd(const float* gz, const float* logvar, const float* eps, const float* gmu_loss,
                       const float* glv_loss, float* gmu, float* glv, size_t n, void* stream);
int rg_tanh_inplace(float* x, size_t n, void* stream);
int rg_add_inplace(float* y, const float* x, size_t n, void* stream);     /* y += x (the two heads' gradients meet) */
/* betaVAEloss (betaVAE.py:145-163): losses[3] = {total, reconstruction (MSE over N*F), kl}; total = recons + beta*kl
 * when training else recons.  Also writes d total / d x_recons ([N][ld]), d total / d z_mean, d total / d z_logvar
 * ([N][Z]).  x and x_recons are [N][ld] with zero pad columns F..ld-1.  Deterministic two-stage reduction. */
size_t rg_vae_loss_workspace_bytes(void);
int rg_vae_loss(const float* x, const float* x_recons, int N, int F, int ld, const float* z_mean, const float* z_logvar,
                int Z, float beta, int training, float* losses, float* g_recons, float* g_mean, float* g_logvar, void* ws,
                size_t ws_bytes, void* stream);

/* bf16 -> fp32 widening of a flat buffer (gradient all-reduce decompression) */
int rg_widen_bf16(const void* src, float* dst, size_t n, void* stream);
/* fp32 -> dtype cast with optional row padding: dst[M][ldd] = src[M][K] (pad columns zeroed) */
int rg_cast_pad(const float* src, void* dst, int M, int K, int ldd, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hardware self-tests (used by tests/ on the GPU box): check the MFMA / transposed-LDS-read lane
 * maps this library relies on with exact integer data.  Return 0 when the layouts match.
 * ------------------------------------------------------------------------------------------- */
int rg_selftest_layouts(int* detail, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Resize-convolution block of DCGANUpGenerator (src/dcgan.py:45-56 and the last block :76-84):
 *   y[N][2H][2W][Cout] = Conv2d(Cin, Cout, 3, 1, 0)(ReflectionPad2d(1)(Upsample(x2, bilinear)(x[N][H][W][Cin]))) + bias
 * w[Cout][Cin][3][3] and bias[Cout] in PyTorch layout.  out_nchw_f32 / gy_nchw_f32 = 1: the image-side tensor is
 * NCHW fp32 (the generator's output block).  bf16 NHWC blocks run on the matrix cores: forward (Cin % 64 == 0) =
 * padded upsampled image materialised in the workspace + 9-tap stride-1 implicit GEMM with the bias in its epilogue;
 * data gradient (Cout % 64 == 0) = 9-tap full correlation onto the padded grid + adjoint of pad/upsample; weight
 * gradient = the pixel-contracting kernel of the 4x4 layers with 9 stride-1 taps.  Everything else (fp32, the NCHW
 * image block with Cout = 3): functor-GEMM kernels on the vector ALUs.
 * ------------------------------------------------------------------------------------------- */
size_t rg_upconv3_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int rg_upconv3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin, int Cout,
                   int out_nchw_f32, int dtype, int algo, void* ws, size_t ws_bytes, void* stream);
/* gx[N][H][W][Cin] = d/dx of the block for the output gradient gy */
int rg_upconv3_bwd_data(const void* gy, int gy_nchw_f32, const float* w, void* gx, int N, int H, int W, int Cin,
                        int Cout, int dtype, int algo, void* ws, size_t ws_bytes, void* stream);
/* dw[Cout][Cin][3][3] (+)= weight gradient (the bias gradient is a column sum of gy: rg_col_sum / rg_nchw_chan_sum) */
int rg_upconv3_wgrad(const void* gy, int gy_nchw_f32, const void* x, float* dw, int N, int H, int W, int Cin, int Cout,
                     int dtype, int algo, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Split forms for synchronised (global-batch) statistics in a data-parallel run (--sync-stats): *_sums
 * writes the RANK-LOCAL column sums, the caller all-reduces them (tiny RCCL all-reduces), *_apply finishes with the
 * global sums and the global row count.  Forward statistics use rg_bn_stats + rg_bn_finalize(count = global) +
 * rg_bn_act; the penalty norm is one scalar all-reduce of rg_sqnorm's result.
 * ------------------------------------------------------------------------------------------- */
int rg_bn_bwd_sums(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float* s_gy, float* s_gyxh, int M, int C, float slope, int dtype, void* ws,
                   size_t ws_bytes, void* stream);
int rg_bn_bwd_apply(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, const float* s_gy, const float* s_gyxh, void* gz, int M, int C, int M_total,
                    float slope, int dtype, void* stream);
int rg_bn_tangent_sums(const void* z, const void* zt, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, float* s_zt, float* s_xhzt, int M, int C, float slope, int dtype, void* ws,
                       size_t ws_bytes, void* stream);
int rg_bn_tangent_apply(const void* z, const void* zt, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const float* s_zt, const float* s_xhzt, void* at, int M, int C, int M_total,
                        float slope, int dtype, void* stream);
/* raw3[3][C] = column sums of (gy*zt', qy, qy*xhat) of rg_bn_double_bwd's reduction */
int rg_bn_dbl_sums(const void* z, const void* qa, const void* zt, const void* ga1, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, float* raw3, int M, int C, float slope, int dtype, void* ws,
                   size_t ws_bytes, void* stream);
/* dgamma / dbeta receive this rank's contribution (summed later by the gradient all-reduce); ws >= 5*C floats */
int rg_bn_dbl_apply(const void* z, const void* qa, const void* zt, const void* ga1, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, const float* s_gy, const float* s_gyxh, const float* s_zt,
                    const float* s_xhzt, const float* raw3_global, const float* raw3_local, void* pz, float* dgamma,
                    float* dbeta, int accumulate, int M, int C, int M_total, float slope, int dtype, void* ws,
                    size_t ws_bytes, void* stream);
int rg_latent_stats(const float* u, const float* z, float* s, float* ss, int N, int E, void* stream);
int rg_latent_apply(const float* u, const float* z, const float* s, const float* ss, float* out, int N, int E, int N_total,
                    void* stream);

/* Generated images for export (src/gan_utils.py:236-241): y_nhwc[N][H][W][C] = x_nchw * 0.5 + 0.5 (the inverse of
 * the input normalisation, transforms.Normalize(-mean/std, 1/std) with mean = std = 0.5) in NHWC order. */
int rg_export_images_nhwc(const float* x_nchw, float* y_nhwc, int N, int C, int H, int W, void* stream);

/* Generator-only inference with eval-mode BatchNorm (torchgan's sample grid: generator.eval(); SURVEY 8 f1 / BASELINE
 * configs[4]): the folded BatchNorm affine and the LeakyReLU are applied to the fp32 accumulators in the conv epilogue,
 *   y = lrelu(conv(x) * scale[c] + shift[c], slope),  scale = gamma / sqrt(running_var + eps),  shift = beta - running_mean * scale
 * so an inference layer is ONE kernel (no statistics, no apply pass).  bf16 MFMA path only (RG_EUNSUPPORTED otherwise).
 * rg_g0_fwd_affine: the generator's first layer, scale / shift of length 16*C in the layer's (tap, c) column order. */
int rg_conv_up_affine(const void* x, const void* wup, void* y, int N, int Ho, int Wo, int O, int I, const float* scale,
                      const float* shift, float slope, void* ws, size_t ws_bytes, void* stream);
int rg_g0_fwd_affine(const float* z, const void* wp, void* y, int N, int E, int C, const float* scale, const float* shift,
                     float slope, void* ws, size_t ws_bytes, void* stream);

/* fp8 (OCP e4m3) operands for generator-only inference (BASELINE configs[4]: "Generator-only tile synthesis, batch 4096
 * fp8"): activations x8[N][Ho][Wo][O] and weights wup8[16][I][O] (rg_conv_up) / b8[Ncols][K] (plain GEMM: the generator's
 * first layer) are fp8 bytes, fp32 accumulation (v_mfma_f32_16x16x32_fp8_fp8, 128-deep k-tiles on the pipeline of
 * rg_conv8.hip), epilogue y = lrelu(acc * scale[c] + shift[c], slope) -- scale carries the folded BatchNorm AND the
 * per-column weight quantisation scale -- written as bf16 (out_fp8 = 0) or fp8 (1, the next fp8 layer's input).
 * rg_fp8_supported: 1 when a (M rows, K = taps * channels, Ncols) problem has an fp8 kernel (channels multiples of 128,
 * Ncols a multiple of 128, M >= 256 / 512); the caller keeps the layer in bf16 otherwise.
 * rg_cast_fp8: dst[i] = fp8(src[i] * mul).  rg_selftest_fp8: lane map of the fp8 MFMA and converter known answers. */
int rg_conv_up_fp8(const void* x8, const void* wup8, void* y, int N, int Ho, int Wo, int O, int I, const float* scale,
                   const float* shift, float slope, int out_fp8, void* stream);
int rg_gemm_fp8(const void* a8, const void* b8, void* y, int M, int K, int Ncols, const float* scale, const float* shift,
                float slope, int out_fp8, void* stream);
int rg_fp8_supported(int M, int K, int Ncols, int taps);
int rg_cast_fp8(const float* src, void* dst, size_t n, float mul, void* stream);
int rg_selftest_fp8(int* detail, void* stream);

/* Input contract of the discriminator (src/histopathology_gan.py:106-109: ToTensor + Normalize(0.5, 0.5); dataset
 * output src/read_data.py:339-342,366-370): dst[i] = ((float)src_u8[i] / 255 - mean) / std, element order unchanged
 * (uint8 CHW tiles stay CHW).  The same three fp32 operations as the host transform: bit-identical to it.  Lets the
 * loader hand over uint8 tiles (a quarter of the PCIe bytes) and normalise on the device. */
int rg_u8_to_norm(const void* src_u8, float* dst, size_t n, float mean, float stdv, void* stream);

/* ---- the fp32 mode on the bf16 matrix cores, operands split ONCE PER TENSOR (rna_gan_amd/csrc/rg_conv8f.hip, rg_wgrad8f.hip) ----
 * The reference computes in fp32 (src/betaVAE.py:184,223; the GAN CLI never enables AMP).  An fp32 value is the exact sum of three
 * bf16 numbers v = h + m + l; rg_split_planes writes them as a plane-major buffer planes[3][n] (bf16, n % 8 == 0) in one
 * bandwidth-bound pass, and the stride-2 conv / transposed conv / weight gradient run the product's 8-wave bf16 kernels over
 * K-concatenated plane pairs with fp32 accumulation and an fp32 result:
 *   products = 6: hh hm mh hl lh mm -- every term down to 2^-24 |a b| (the f32 instruction's accuracy class)
 *   products = 3: hh hm mh          -- 2^-16 |a b| per product
 * x_planes: planes of the NHWC activation the bf16 entry point would take (rg_conv_down: [N][2Hlow][2Wlow][I]; rg_conv_up:
 * [N][Hlow][Wlow][O]); w_planes: planes of wdn[O][16][I] (down) / wup[16][I][O] (up); y fp32 NHWC.  stats_partial as rg_conv_down
 * (rg_f32p_conv_stats_rows rows; 0 = the launch splits K and writes none).  rg_f32p_wgrad: dw[O][16][I] (+)= one or two
 * (low, high) segments, all four operands as planes.  *_supported: 0 = use the RG_F32 entry points.  bf16 library only. */
int rg_split_planes(const float* src, void* planes_bf16, size_t n, void* stream);
int rg_f32p_conv_supported(int up, int N, int Hlow, int Wlow, int O, int I, int products);
size_t rg_f32p_conv_workspace_bytes(int up, int N, int Hlow, int Wlow, int O, int I, int products);
int rg_f32p_conv_stats_rows(int up, int N, int Hlow, int Wlow, int O, int I, int products);
int rg_f32p_conv(int up, const void* x_planes, const void* w_planes, float* y, int N, int Hlow, int Wlow, int O, int I,
                 int products, float* stats_partial, const float* mask_f32, float mask_slope, void* ws, size_t ws_bytes,
                 void* stream);
/* mask_f32 (optional, y's shape): y *= (mask > 0 ? 1 : mask_slope), the consumer's LeakyReLU backward as in rg_conv_up -- only
 * where rg_f32p_conv_mask_supported says 1 (the 64-column transposed conv); elsewhere the caller runs rg_lrelu_bwd behind it. */
int rg_f32p_conv_mask_supported(int up, int N, int Hlow, int Wlow, int O, int I, int products);
int rg_f32p_wgrad_supported(int N, int Ho, int Wo, int O, int I, int products);
size_t rg_f32p_wgrad_workspace_bytes(int N, int Ho, int Wo, int O, int I, int products, int two);
int rg_f32p_wgrad(const void* low0, const void* high0, const void* low1, const void* high1, float* dw, int N, int Ho, int Wo,
                  int O, int I, int products, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* On-box ceilings for the roofline object of bench.py ("re-measure both on the box (MFMA-loop and stream-copy
 * microbenchmarks) and report against both nominal and measured peak").  Measurement kernels only -- no product path calls
 * them; the caller times back-to-back launches with events on `stream`.  (rna_gan_amd/csrc/rg_probe.hip)
 * rg_probe_mfma_bare: `blocks` workgroups of 4 * waves_per_simd waves, each wave `iters` times the MFMAs of one 128 x 64 x 64
 *   wave k-tile (v_mfma_f32_16x16x32_bf16: mfma_shape 16, v_mfma_f32_32x32x16_bf16: 32) on random register operands.
 * rg_probe_lds_mfma: the product's 8-wave conv k-loop (conv8_kernel, 256 x 256 x 64 tile) over LDS-resident stages with its
 *   LDS-DMA issue compiled out; a[blocks * 256][128], b[256][128] bf16 operands, c[blocks * 256][256] bf16 result; iters even.
 * rg_probe_copy: float4 stream copy of nbytes (multiple of 16); variant 0 plain / 1 non-temporal loads and stores, `blocks`
 *   workgroups (0: 2048).  rg_probe_fill_bf16: n bf16 values uniform in [-1, 1).
 * *flops_out = algorithmic FLOPs of the launch. */
int rg_probe_mfma_bare(int mfma_shape, int waves_per_simd, int blocks, int iters, float* scratch, double* flops_out, void* stream);
int rg_probe_lds_mfma(int mfma_shape, int blocks, int iters, const void* a, const void* b, void* c, double* flops_out,
                      void* stream);
int rg_probe_copy(const void* src, void* dst, size_t nbytes, int variant, int blocks, void* stream);
int rg_probe_fill_bf16(void* p, size_t n, unsigned seed, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* RNAGAN_HIP_H */
