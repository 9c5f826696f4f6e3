"""ctypes binding of librnagan_hip.so (include/rnagan_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librnagan_hip.so")

RG_F32, RG_BF16, RG_F16 = 0, 1, 2
ALGO_AUTO, ALGO_GENERIC, ALGO_MFMA = 0, 1, 2

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_z = C.c_size_t

# name -> (restype, [argtypes])   -- must list every symbol declared in include/rnagan_hip.h
PROTOTYPES = {
    "rg_version": (_i, []),
    "rg_last_error": (C.c_char_p, []),
    "rg_set_option": (_i, [C.c_char_p, _i]),
    "rg_pack_conv_weight": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "rg_conv_stats_rows": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_down": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _z, _p]),
    "rg_conv_up": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p, _i, _i, _p, _z, _p]),
    "rg_conv_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_conv_wgrad2": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_conv_wgrad_slabs": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _p, _p]),
    "rg_grad_to_wire": (_i, [_p, _p, _z, _i, _p, _p, _p, _p, _p, _p]),
    "rg_conv_wgrad_wire": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rg_conv_wgrad_adam_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_wgrad_adam": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rg_first_down": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rg_first_down_bits": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rg_sign_pack": (_i, [_p, _p, C.c_longlong, _i, _i, _p]),
    "rg_first_down_masked_supported": (_i, [_i, _i, _i, _i, _i]),
    "rg_first_down_masked": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _i, _i, _i, _p]),
    "rg_conv_up_maskbits_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_up_maskbits": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _i, _i, _p, _z, _p]),
    "rg_last_up": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rg_skinny_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "rg_skinny_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_skinny_wgrad_slabs": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p, _p, _p, _p]),
    "rg_skinny_wgrad_bias": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _p]),
    "rg_pack_g0_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "rg_pack_conv_wup_from_bf16": (_i, [_p, _p, _i, _i, _p]),
    "rg_pack_conv_wup_from_bf16_multi": (_i, [_i, _p, _p, _p, _p, _p]),
    "rg_pack_g0_weight_from_bf16": (_i, [_p, _p, _i, _i, _p]),
    "rg_g0_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_g0_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "rg_g0_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_head_fwd": (_i, [_p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "rg_head_grad": (_i, [_p, _p, _i, _f, _f, _p]),
    "rg_head_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "rg_head_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "rg_pack_linear_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "rg_linear_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "rg_linear_affine_act": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _z, _p]),
    "rg_colreduce_workspace_bytes": (_z, [_i, _i, _i]),
    "rg_bn_stats": (_i, [_p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rg_bn_finalize": (_i, [_p, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p]),
    "rg_bn_forward": (_i, [_p, _i, _i, _f, _f, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _p, _z, _p]),
    "rg_bn_forward_partials": (_i, [_p, _i, _p, _i, _i, _f, _f, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _p, _z, _p]),
    "rg_bn_stats_finalize": (_i, [_p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _i, _p, _z, _p]),
    "rg_bn_act": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "rg_bn_act_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _z, _p]),
    "rg_bn_act_bwd_g2": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _z, _p]),
    "rg_bn_finalize_partials": (_i, [_p, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _z, _p]),
    "rg_linear_wgrad_adam": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _i, _p]),
    "rg_last_up_post_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "rg_last_up_post": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "rg_last_up_part_chan_sum": (_i, [_p, _i, _p, _i, _p]),
    "rg_gp_coef_parts": (_i, [_p, _i, _p, _p, _p, _f, _p]),
    "rg_last_up_pre_supported": (_i, [_i, _i, _i, _i]),
    "rg_last_up_pre": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rg_bn_forward_g2": (_i, [_p, _i, _i, _p, _i, _i, _f, _f, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _p, _z, _p]),
    "rg_bn_finalize_partials_g2": (_i, [_p, _i, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _z, _p]),
    "rg_bn_tangent": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _z, _p]),
    "rg_bn_double_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i,
                              _p, _z, _p]),
    "rg_lrelu_bwd": (_i, [_p, _p, _p, _z, _f, _i, _p]),
    "rg_col_sum": (_i, [_p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rg_tanh_bwd": (_i, [_p, _p, _p, _z, _p]),
    "rg_nchw_chan_sum": (_i, [_p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rg_interp": (_i, [_p, _p, _p, _z, _f, _p]),
    "rg_reduce_workspace_bytes": (_z, [_z]),
    "rg_sqnorm": (_i, [_p, _p, _z, _p, _z, _p]),
    "rg_gp_coef": (_i, [_p, _p, _p, _f, _p]),
    "rg_scale_by": (_i, [_p, _p, _p, _z, _p]),
    "rg_mean_diff": (_i, [_p, _p, _p, _i, _f, _p]),
    "rg_latent_prep": (_i, [_p, _p, _p, _i, _i, _p]),
    "rg_adam_step": (_i, [_p, _p, _p, _p, _z, _i, _d, _d, _d, _d, _p]),
    "rg_clamp": (_i, [_p, _z, _f, _f, _p]),
    "rg_upconv3_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "rg_upconv3_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_upconv3_bwd_data": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_upconv3_wgrad": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_export_images_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "rg_bn_bwd_sums": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _z, _p]),
    "rg_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "rg_bn_tangent_sums": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _z, _p]),
    "rg_bn_tangent_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "rg_bn_dbl_sums": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _z, _p]),
    "rg_bn_dbl_apply": (_i, [_p] * 17 + [_i, _i, _i, _i, _f, _i, _p, _z, _p]),
    "rg_latent_stats": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "rg_latent_apply": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "rg_adam_step_dev": (_i, [_p, _p, _p, _p, _z, _p, _p, _p, _p]),
    "rg_adam_step_slabs": (_i, [_p, _p, _p, _p, _z, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "rg_interp_dev": (_i, [_p, _p, _p, _z, _p, _p]),
    "rg_adam_hyper_dev": (_i, [_p, _d, _d, _d, _d, _d, _p, _p]),
    "rg_adam_hyper_dev2": (_i, [_p, _d, _d, _d, _d, _d, _d, _p, _p]),
    "rg_storage_dtype": (_i, []),
    "rg_gp_coef_scaled": (_i, [_p, _p, _p, _f, _f, _f, _p]),
    "rg_gp_coef_parts_scaled": (_i, [_p, _i, _p, _p, _p, _f, _f, _f, _p]),
    "rg_transpose_f32": (_i, [_p, _p, _i, _i, _p]),
    "rg_transpose_pack_bf16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "rg_gemm_nt_bf16_workspace_bytes": (_z, [_i, _i, _i]),
    "rg_gemm_nt_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _z, _p]),
    "rg_vae_dropout": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "rg_vae_reparam": (_i, [_p, _p, _p, _p, _z, _p]),
    "rg_vae_reparam_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "rg_tanh_inplace": (_i, [_p, _z, _p]),
    "rg_add_inplace": (_i, [_p, _p, _z, _p]),
    "rg_vae_loss_workspace_bytes": (_z, []),
    "rg_vae_loss": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _f, _i, _p, _p, _p, _p, _p, _z, _p]),
    "rg_widen_bf16": (_i, [_p, _p, _z, _p]),
    "rg_cast_pad": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "rg_selftest_layouts": (_i, [_p, _p]),
    "rg_u8_to_norm": (_i, [_p, _p, _z, _f, _f, _p]),
    "rg_conv_up_fp8": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _f, _i, _p]),
    "rg_gemm_fp8": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _f, _i, _p]),
    "rg_fp8_supported": (_i, [_i, _i, _i, _i]),
    "rg_cast_fp8": (_i, [_p, _p, _z, _f, _p]),
    "rg_selftest_fp8": (_i, [_p, _p]),
    "rg_g0_wgrad_adam_supported": (_i, [_i, _i, _i, _i]),
    "rg_g0_wgrad_adam": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rg_im2col_nhwc": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rg_pool2d_nhwc": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rg_nchw_to_nhwc_affine": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "rg_spatial_mean_nhwc": (_i, [_p, _p, _i, _i, _i, _p]),
    "rg_conv_bnbwd_rows": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_down_bnbwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _i, _p, _i, _i, _p, _z, _p]),
    "rg_conv_up_bnbwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _i, _p, _i, _i, _p, _z, _p]),
    "rg_bn_act_bwd_partials": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _z,
                                    _p]),
    "rg_conv_split": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_slab_dtype": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "rg_conv_down_partial": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_conv_up_partial": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_slab_bn_supported": (_i, [C.c_longlong, _i, _i, _i]),
    "rg_slab_bn_scratch_bytes": (_z, [C.c_longlong, _i, _i]),
    "rg_slab_bn_sync_words": (_z, []),
    "rg_bn_forward_slabs": (_i, [_p, _i, _z, _i, _p, _p, C.c_longlong, _i, _i, _f, _f, _p, _p, _f, _p, _p, _p, _p, _p, _p, _z, _p, _p]),
    "rg_bn_tangent_slabs": (_i, [_p, _i, _z, _i, _p, _p, _p, C.c_longlong, _i, _p, _p, _p, _p, _f, _p, _p, _p, _z, _p, _p]),
    "rg_bn_act_bwd_slabs": (_i, [_p, _i, _z, _i, _p, _p, _p, C.c_longlong, _i, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _p, _z,
                                 _p, _p]),
    "rg_conv_up_affine": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _f, _p, _z, _p]),
    "rg_g0_fwd_affine": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _f, _p, _z, _p]),
    "rg_split_planes": (_i, [_p, _p, _z, _p]),
    "rg_f32p_conv_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_f32p_conv_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_f32p_conv_stats_rows": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_f32p_conv": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _z, _p]),
    "rg_f32p_conv_mask_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_f32p_wgrad_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "rg_f32p_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "rg_f32p_wgrad": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rg_probe_mfma_bare": (_i, [_i, _i, _i, _i, _p, _p, _p]),
    "rg_probe_lds_mfma": (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    "rg_probe_copy": (_i, [_p, _p, _z, _i, _i, _p]),
    "rg_probe_fill_bf16": (_i, [_p, _z, C.c_uint, _p]),
}

# must equal rg_version() of the library (rna_gan_amd/csrc/rg_api.hip): bumped together with PROTOTYPES
ABI_VERSION = 600

_libs = {}
LIB_PATH_F16 = os.path.join(_HERE, "librnagan_hip_f16.so")
# entry points that exist only in the bf16 build (measurement probes; the fp32 mode's bf16-plane kernels)
BF16_ONLY_PREFIXES = ("rg_probe_", "rg_split_planes", "rg_f32p_")


def load(half: str = "bf16"):
    """Load the HIP library (once per build).  half = "bf16": librnagan_hip.so; "f16": librnagan_hip_f16.so (the same sources,
    IEEE fp16 as the 16-bit storage type).  Raises RuntimeError if it cannot be loaded."""
    if half in _libs:
        return _libs[half]
    if half not in ("bf16", "f16"):
        raise ValueError("half must be 'bf16' or 'f16'")
    path = LIB_PATH if half == "bf16" else LIB_PATH_F16
    if not os.path.exists(path):
        raise RuntimeError(
            "rna_gan_amd: %s not found. Build it with `python -m rna_gan_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback." % path)
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise RuntimeError("rna_gan_amd: cannot load %s: %s" % (path, e))
    for name, (res, args) in PROTOTYPES.items():
        if half == "f16" and name.startswith(BF16_ONLY_PREFIXES):
            continue
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError("rna_gan_amd: symbol %s missing from %s: the library is older than this package -- rebuild "
                               "it with `python -m rna_gan_amd.build`" % (name, path))
        fn.restype = res
        fn.argtypes = args
    got = lib.rg_version()
    if got != ABI_VERSION:
        raise RuntimeError("rna_gan_amd: %s reports ABI version %d, this package binds version %d (a stale or swapped "
                           "build): rebuild it with `python -m rna_gan_amd.build`" % (path, got, ABI_VERSION))
    want = RG_BF16 if half == "bf16" else RG_F16
    if lib.rg_storage_dtype() != want:
        raise RuntimeError("rna_gan_amd: %s stores dtype code %d, expected %d (a swapped build)" % (path, lib.rg_storage_dtype(), want))
    _libs[half] = lib
    return lib


_LAST_ERR_LIBS = ("bf16", "f16")


def check(rc: int, what: str):
    if rc != 0:
        # (the error string is thread-local per library: take it from whichever loaded build has one)
        msgs = []
        for h in _LAST_ERR_LIBS:
            if h in _libs:
                m = _libs[h].rg_last_error()
                if m:
                    msgs.append(m.decode() if len(_libs) == 1 else "[%s build] %s" % (h, m.decode()))
        raise RuntimeError("rna_gan_amd: %s failed (%d): %s" % (what, rc, " | ".join(msgs) if msgs else "?"))
