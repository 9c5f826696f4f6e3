"""Explicit forward / backward / gradient-penalty sequencing of the RNA-GAN DCGAN pair.

This module replaces what the reference gets from PyTorch autograd on its hot path
(reference: src/wgan_loss.py:107-129 G step, :213-263 D step, :369-389 GP step, with the
functional losses :24-44).  It contains NO arithmetic of its own: every tensor operation goes
through an ``ops`` backend object whose methods map 1:1 onto the C-ABI entry points declared in
include/rnagan_hip.h (product backend: rna_gan_amd.ops_hip.HipOps).  The same sequencing can be
driven by the torch twin in oracle/ops_ref.py, which is how the algorithm (in particular the
second-order gradient-penalty pass) is validated on CPU against the autograd oracle.

Layout conventions: see include/rnagan_hip.h / DESIGN.md ("Data layout in HBM").

Gradient penalty without autograd (DESIGN.md "GP second-order pass"):
  L = lambd * (||g|| - 1)^2,  g = d(sum_n D(xhat)_n)/d xhat.
  dL/dtheta = d/dtheta <v, g(theta)> with v = dL/dg held constant
            = d/dtheta [ directional derivative of sum_n D(xhat)_n along v ]      (linearity)
  so: (1) primal forward, (2) first backward (data gradients only) -> g, v,
      (3) tangent forward along v (forward mode through conv / train-mode BN / LeakyReLU),
      (4) one joint reverse sweep over (primal, tangent).  The cotangent of every tangent
          quantity equals the matching first-backward gradient (already in memory), so the joint
          sweep only propagates the PRIMAL cotangent: 6 conv passes in total, the algorithmic
          minimum for this loss.
"""
from __future__ import annotations

from typing import List, Optional

import torch


class ConvW:
    """Handle of one 4x4 conv weight: fp32 master (+ optional bias[.]) and its gradient.

    O = channels on the LOW-resolution side, I = channels on the HIGH-resolution side: this is
    nn.Conv2d's (out,in,kh,kw) and nn.ConvTranspose2d's (in,out,kh,kw), so one handle type serves
    both networks.  ``layout`` says how ``w`` and ``dw`` are stored:
      "OIHW": ``w[O][I][4][4]``, the PyTorch layout (image-side layers, G.0, the head, and any layer of a plain
              nn.Module handed to the CPU twin);
      "OHWI": ``w[O][4][4][I]``, tap-major -- what the HIP conv_down / conv_up / conv_wgrad kernels take
              (models.tap_major_ re-homes a module's parameters this way; the nn.Parameter stays a strided
              view with the PyTorch shape).
    ``packs`` caches backend-private re-layouts (bf16 GEMM operand images); they are rebuilt when
    ``version`` changes (after an optimizer step / state_dict load).
    """

    __slots__ = ("w", "bias", "dw", "dbias", "packs", "packs_version", "version", "layout", "shadow",
                 "shadow_version", "_fp8", "fuse_step", "pending_wgrad", "factor_stage", "owner", "defer_slabs",
                 "pending_slabs", "_slab_ws", "wire_slot", "pending_bias", "_bias_ws")

    def __init__(self, w, bias=None, dw=None, dbias=None, layout="OIHW"):
        self.w = w
        self.bias = bias
        self.dw = dw
        self.dbias = dbias
        self.packs = None
        self.packs_version = -1
        self.version = 0
        self.layout = layout
        self.shadow = None            # bf16 [O][16][I] image of a tap-major master maintained by the fused Adam
        self.shadow_version = -1      # master version the shadow reflects
        self._fp8 = None              # backend-private fp8 inference image (key, bytes, column scales)
        self.owner = None             # "G" / "D": which network the layer belongs to (bench.py's per-network roofline rows)
        # 4 x 4 conv layers (HIP backend, bf16, single process): defer_slabs -- set by the train_op runner for a gradient pass whose
        # optimizer step follows immediately -- lets a split-K weight-gradient launch leave its fp32 slabs in _slab_ws
        # (pending_slabs = (buffer, nsplit)) instead of reducing them into dw: rna_gan_amd.optim.Adam sums them inside its step
        self.defer_slabs = False
        self.pending_slabs = None
        self._slab_ws = None
        self.pending_bias = None      # image-side layers: (buffer, count) of deferred BIAS-gradient partials [count][64] (same pass)
        self._bias_ws = None
        # data parallel (bf16 wire): the layer's slice of the all-reduce wire buffer, set by the train_op runner together with
        # defer_slabs -- a weight-gradient launch without split-K writes its bf16 tile there (pending_slabs = (None, -1, 0)), a
        # split one leaves its slabs for rg_grad_to_wire; no fp32 gradient of the layer is formed in that pass
        self.wire_slot = None
        # generator layer 0 only (HIP backend): fuse_step -- set by the train_op runner for the duration of one gradient pass
        # whose optimizer step follows immediately -- lets g0_wgrad leave its operands in pending_wgrad instead of writing dw;
        # the fused Adam then forms the gradient and applies the step in one kernel (rg_g0_wgrad_adam)
        self.fuse_step = False
        self.pending_wgrad = None
        # data parallel: (z slot, gz0 slot) = this rank's slices of the gathered factor buffers; g0_wgrad then copies its
        # operands there (the ranks all-gather the FACTORS of the rank-64 gradient instead of all-reducing the gradient)
        self.factor_stage = None

    @classmethod
    def from_param(cls, weight, grad=None):
        """Handle of a 4x4 conv nn.Parameter (logical shape [O][I][4][4]): tap-major if its storage is."""
        w = weight.data if hasattr(weight, "data") else weight
        t = w.permute(0, 2, 3, 1)
        if t.is_contiguous() and (not w.is_contiguous() or w.shape[1] == 1):
            return cls(t, None, None if grad is None else grad.permute(0, 2, 3, 1), None, "OHWI")
        return cls(w, None, grad, None, "OIHW")

    @property
    def O(self):
        return self.w.shape[0]

    @property
    def I(self):
        return self.w.shape[3] if self.layout == "OHWI" else self.w.shape[1]

    def oihw(self):
        """The master in PyTorch order (a view)."""
        return self.w.permute(0, 3, 1, 2) if self.layout == "OHWI" else self.w

    def store_grad_oihw(self, g, accumulate):
        d = self.dw.permute(0, 3, 1, 2) if self.layout == "OHWI" else self.dw
        if accumulate:
            d.add_(g)
        else:
            d.copy_(g)


class BNP:
    """BatchNorm2d parameters/buffers of one layer (train-mode statistics are per call)."""

    __slots__ = ("gamma", "beta", "running_mean", "running_var", "nbt", "dgamma", "dbeta", "eps", "momentum")

    def __init__(self, gamma, beta, running_mean, running_var, nbt, dgamma=None, dbeta=None,
                 eps=1e-5, momentum=0.1):
        self.gamma, self.beta = gamma, beta
        self.running_mean, self.running_var, self.nbt = running_mean, running_var, nbt
        self.dgamma, self.dbeta = dgamma, dbeta
        self.eps, self.momentum = eps, momentum


class DiscNet:
    """Discriminator as the engine sees it (torchgan DCGANDiscriminator recipe)."""

    def __init__(self, conv0: ConvW, blocks: List, head: ConvW, slope: float, last_slope: float):
        self.conv0, self.blocks, self.head = conv0, blocks, head   # blocks: [(ConvW, BNP)]
        self.slope, self.last_slope = slope, last_slope

    def convs(self):
        return [self.conv0, self.head] + [b[0] for b in self.blocks]

    def bump(self):
        for cw in self.convs():
            cw.version += 1


class GenNet:
    """Generator as the engine sees it (torchgan DCGANGenerator recipe)."""

    def __init__(self, g0: ConvW, bn0: BNP, blocks: List, last: ConvW, slope: float):
        self.g0, self.bn0, self.blocks, self.last = g0, bn0, blocks, last   # blocks: [(ConvW, BNP)]
        self.slope = slope

    def convs(self):
        return [self.g0, self.last] + [b[0] for b in self.blocks]

    def bump(self):
        for cw in self.convs():
            cw.version += 1


class UpGenNet:
    """DCGANUpGenerator as the engine sees it (src/dcgan.py:8-99): G.0 block as GenNet, then R resize-convolution
    blocks [bilinear x2 + ReflectionPad(1) + Conv3x3(+bias) + BN + LReLU] and a last resize-convolution WITHOUT an
    activation.  The 3x3 weights keep the PyTorch layout w[Cout][Cin][3][3]."""

    def __init__(self, g0: ConvW, bn0: BNP, blocks: List, last: ConvW, slope: float):
        self.g0, self.bn0, self.blocks, self.last = g0, bn0, blocks, last   # blocks: [(ConvW, BNP)]
        self.slope = slope

    def convs(self):
        return [self.g0, self.last] + [b[0] for b in self.blocks]

    def bump(self):
        for cw in self.convs():
            cw.version += 1


class _Ctx:
    pass


def _bnb(zs, means, invstds, blocks, l, slope, groups):
    """(z, mean, invstd, gamma, beta, slope, groups) of discriminator block l: what a data-gradient conv needs to produce that
    block's BatchNorm-backward sums in its own epilogue (HIP backend; the twin ignores it)."""
    bn = blocks[l - 1][1]
    return (zs[l], means[l], invstds[l], bn.gamma, bn.beta, slope, groups)


def _refresh(ops, blocks):
    """Before a network's first conv of a pass: stale bf16 weight images of all its layers rebuilt in one launch (HIP backend)."""
    fn = getattr(ops, "refresh_packs", None)
    if fn is not None:
        fn([b[0] for b in blocks])


def _bn_forward(ops, z, bn: BNP, slope, update_running=True, partials=None):
    """partials: BatchNorm column sums written by the epilogue of the conv that produced z (or None)."""
    if update_running:
        return ops.bn_forward(z, bn.gamma, bn.beta, slope, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.nbt,
                              partials=partials)
    return ops.bn_forward(z, bn.gamma, bn.beta, slope, bn.eps, bn.momentum, partials=partials)


# --------------------------------------------------------------------------------------------
# Discriminator
# --------------------------------------------------------------------------------------------
def disc_forward(ops, D: DiscNet, x_nchw, update_running=True):
    """D(x): Conv+LReLU, R x [Conv+BN(train)+LReLU], Conv(4x4 valid)+LReLU -> (N,).
    Returns (out, ctx); ctx keeps what the backward passes need."""
    ctx = _Ctx()
    ctx.x = x_nchw
    _refresh(ops, D.blocks)
    a = ops.first_down(x_nchw, D.conv0, D.conv0.bias, D.slope)
    ctx.a = [a]
    ctx.z, ctx.mean, ctx.invstd = [None], [None], [None]
    for cw, bn in D.blocks:
        # defer=1: a split-K launch leaves its slabs to the BatchNorm op that follows (one fused reduce + statistics + apply)
        z, st = ops.conv_down(a, cw, want_stats=True, defer=1)
        a, mean, invstd = _bn_forward(ops, z, bn, D.slope, update_running, st)
        ctx.z.append(z); ctx.mean.append(mean); ctx.invstd.append(invstd); ctx.a.append(a)
    ctx.h, out = ops.head_fwd(a, D.head, D.last_slope)
    return out, ctx


def disc_backward(ops, D: DiscNet, ctx, coef: float, wgrad, accumulate: bool,
                  need_input_grad: bool, keep_for_gp: bool = False, input_post=None, partner=None):
    """Backward of sum_n coef * D(x)_n (coef: a float, or an (N,) tensor of per-sample cotangents as torch autograd
    hands them over).  wgrad: also produce parameter gradients (written with ``accumulate`` semantics into the
    .dw/.dbias/.dgamma/.dbeta buffers).  Returns d/dx (NCHW fp32) if requested.  keep_for_gp stores the per-layer
    first-backward gradients on ctx.  input_post (with need_input_grad): {"tanh_img": img or None} -- the first consumer's
    pass over d/dx fused into the kernel that writes it (ops.last_up_post): d/dx multiplied by 1 - img^2 (the cotangent of
    the generator's Tanh) and / or per-workgroup partial sums (channel sums, sum of squares) left in ctx.gx_parts; when the
    backend has no such kernel for the shape, ctx.gx_parts stays None and d/dx is returned unmodified.
    wgrad == "defer": the small parameter gradients (head, BatchNorm, layer-0 bias) are produced now, the conv weight
    gradients are NOT -- their operands gz stay on ctx.gz_keep for a later call with partner=ctx, whose per-layer weight
    gradient is then ONE two-segment launch over both chains (written, whatever ``accumulate`` says for the rest): the
    data-parallel D-loss step's prefix / rest pair (disc_loss_prefix_dgrad / disc_loss_rest_pairw)."""
    R = len(D.blocks)
    defer_w = isinstance(wgrad, str) and wgrad == "defer"
    if defer_w:
        ctx.gz_keep = [None] * (R + 1)
    ctx.gx_parts = None
    if torch.is_tensor(coef):       # N-length vector: host-side plumbing
        gh = (coef.reshape(-1).float() * torch.where(ctx.h > 0, torch.ones_like(ctx.h),
                                                     torch.full_like(ctx.h, D.last_slope))).contiguous()
    else:
        gh = ops.head_grad(ctx.h, coef, D.last_slope)
    if wgrad:
        with ops.side(gh):
            ops.head_wgrad(gh, ctx.a[R], D.head.dw, accumulate)
    ga = ops.head_bwd_data(gh, D.head)
    if keep_for_gp:
        ctx.gh = gh
        ctx.ga1 = [None] * (R + 1)
        ctx.gz1 = [None] * (R + 1)
        ctx.s_gy = [None] * (R + 1)
        ctx.s_gyxh = [None] * (R + 1)
    for l in range(R, 0, -1):
        cw, bn = D.blocks[l - 1]
        gz, s_gy, s_gyxh = ops.bn_act_bwd(ctx.z[l], ga, ctx.mean[l], ctx.invstd[l], bn.gamma, bn.beta,
                                          D.slope, bn.dgamma if wgrad else None,
                                          bn.dbeta if wgrad else None, accumulate, keep_ga=keep_for_gp)
        if keep_for_gp:
            ctx.ga1[l], ctx.gz1[l], ctx.s_gy[l], ctx.s_gyxh[l] = ga, gz, s_gy, s_gyxh
        if defer_w:
            ctx.gz_keep[l] = gz
        elif wgrad and partner is not None:
            with ops.side(gz, partner.gz_keep[l]):
                ops.conv_wgrad2(gz, ctx.a[l - 1], partner.gz_keep[l], partner.a[l - 1], cw, False)
        elif wgrad:
            with ops.side(gz):
                ops.conv_wgrad(gz, ctx.a[l - 1], cw, accumulate)
        # the data gradient of layer 1 feeds layer 0's LeakyReLU: its backward is fused into the epilogue
        # (defer=1: the next op on ga is the BatchNorm backward of the layer below, which reduces split-K slabs itself)
        ga = ops.conv_up(gz, cw, defer=1, bn_bwd=_bnb(ctx.z, ctx.mean, ctx.invstd, D.blocks, l - 1, D.slope, 1)) if l > 1 \
            else ops.conv_up(gz, cw, ctx.a[0], D.slope)
    gz0 = ga if R > 0 else ops.lrelu_bwd(ga, ctx.a[0], D.slope)
    if keep_for_gp:
        ctx.gz1[0] = gz0
    if defer_w:
        ctx.gz_keep[0] = gz0
        ops.col_sum(gz0, D.conv0.dbias, accumulate)
    elif wgrad and partner is not None:
        with ops.side(gz0, partner.gz_keep[0]):
            # (dbias: the kernel forms the bias gradient -- the column sums of gz0 -- in the same pass where it can)
            done = ops.skinny_wgrad(gz0, ctx.x, D.conv0.dw, False, dbias=D.conv0.dbias, dbias_accumulate=accumulate)
            ops.skinny_wgrad(partner.gz_keep[0], partner.x, D.conv0.dw, True)
        if not done:
            ops.col_sum(gz0, D.conv0.dbias, accumulate)
    elif wgrad:
        with ops.side(gz0):
            done = ops.skinny_wgrad(gz0, ctx.x, D.conv0.dw, accumulate, dbias=D.conv0.dbias, dbias_accumulate=accumulate)
        if not done:
            ops.col_sum(gz0, D.conv0.dbias, accumulate)
    gx = None
    if need_input_grad and input_post is not None:
        fused = ops.last_up_post(gz0, D.conv0, input_post.get("tanh_img"))
        if fused is not None:
            gx, ctx.gx_parts = fused
    if need_input_grad and gx is None:
        gx = ops.last_up(gz0, D.conv0, None, False)
    if wgrad:
        ops.join()
    return gx


def disc_backward_pair(ops, D: DiscNet, ctx_a, coef_a: float, ctx_b, coef_b: float):
    """Parameter gradients of sum_n coef_a*D(x_a)_n + coef_b*D(x_b)_n (written, not accumulated): the
    two backward chains run in lock step so that each layer's weight gradient is ONE two-segment
    launch (rg_conv_wgrad2) instead of two launches + two reductions.  Used by the D-loss step."""
    R = len(D.blocks)
    gha = ops.head_grad(ctx_a.h, coef_a, D.last_slope)
    ghb = ops.head_grad(ctx_b.h, coef_b, D.last_slope)
    with ops.side(gha, ghb):
        ops.head_wgrad(gha, ctx_a.a[R], D.head.dw, False)
        ops.head_wgrad(ghb, ctx_b.a[R], D.head.dw, True)
    ga_a = ops.head_bwd_data(gha, D.head)
    ga_b = ops.head_bwd_data(ghb, D.head)
    for l in range(R, 0, -1):
        cw, bn = D.blocks[l - 1]
        gz_a, _, _ = ops.bn_act_bwd(ctx_a.z[l], ga_a, ctx_a.mean[l], ctx_a.invstd[l], bn.gamma, bn.beta, D.slope,
                                    bn.dgamma, bn.dbeta, False, keep_ga=False)
        gz_b, _, _ = ops.bn_act_bwd(ctx_b.z[l], ga_b, ctx_b.mean[l], ctx_b.invstd[l], bn.gamma, bn.beta, D.slope,
                                    bn.dgamma, bn.dbeta, True, keep_ga=False)
        with ops.side(gz_a, gz_b):
            ops.conv_wgrad2(gz_a, ctx_a.a[l - 1], gz_b, ctx_b.a[l - 1], cw, False)
        if l > 1:
            # (no deferred slabs here: the two chains share one workspace and chain a's BatchNorm backward runs -- and uses it --
            # before chain b's would consume what chain b's conv left there)
            ga_a, ga_b = ops.conv_up(gz_a, cw), ops.conv_up(gz_b, cw)
        else:
            ga_a = ops.conv_up(gz_a, cw, ctx_a.a[0], D.slope)
            ga_b = ops.conv_up(gz_b, cw, ctx_b.a[0], D.slope)
    if R == 0:
        ga_a, ga_b = ops.lrelu_bwd(ga_a, ctx_a.a[0], D.slope), ops.lrelu_bwd(ga_b, ctx_b.a[0], D.slope)
    gz0_a, gz0_b = ga_a, ga_b
    with ops.side(gz0_a, gz0_b):
        ops.skinny_wgrad(gz0_a, ctx_a.x, D.conv0.dw, False)
        ops.skinny_wgrad(gz0_b, ctx_b.x, D.conv0.dw, True)
    ops.col_sum(gz0_a, D.conv0.dbias, False)
    ops.col_sum(gz0_b, D.conv0.dbias, True)
    ops.join()


def disc_gradient_penalty(ops, D: DiscNet, xhat, lambd: float, update_running=True):
    """lambd*(||d sum D(xhat)/d xhat||_2 - 1)^2 and its parameter gradients (written, not
    accumulated; lambd may carry the data-parallel 1/world factor -- it only scales the gradients).  Reference: src/wgan_loss.py:32-44 + :379-387.  Returns the UNWEIGHTED penalty
    as a 1-element device tensor (the reference returns loss.item() of the unweighted value)."""
    out, ctx = disc_forward(ops, D, xhat, update_running)
    loss, st = disc_gp_first(ops, D, ctx, lambd)
    disc_gp_second(ops, D, ctx, st, accumulate=False, need_input_grad=False)
    return loss


def disc_gp_first(ops, D: DiscNet, ctx, lambd):
    """Steps (2) of the penalty on the context of a primal forward D(xhat): first backward (data gradients only) ->
    g = d sum D(xhat) / d xhat, the UNWEIGHTED penalty (||g|| - 1)^2 (1-element device tensor) and the tangent
    direction v = lambd * 2 (||g|| - 1) / ||g|| * g.  lambd: float, or a 1-element device tensor is NOT supported (the
    coefficient kernel takes a host scalar)."""
    # the squared norm comes out of the kernel that writes g as per-workgroup partial sums (no pass over g) unless the
    # statistics are synchronised over the ranks (then the scalar itself is all-reduced)
    fuse = None if ops.stat_reduce is not None else {"tanh_img": None}
    # fp16 storage (HIP backend): the seed carries ops.gp_seed_scale so that the data gradients stay in fp16's normal range;
    # gp_coef / gp_coef_parts divide the norm by it and put ops.gp_tangent_scale on the tangent direction instead -- the joint
    # reverse sweep is bilinear in (first-backward gradients, tangents), so its parameter gradients arrive scaled by the product
    # (= the step's loss scale, which the optimizer removes).  1.0 everywhere else.
    seed = float(getattr(ops, "gp_seed_scale", 1.0))
    g = disc_backward(ops, D, ctx, seed, wgrad=False, accumulate=False, need_input_grad=True, keep_for_gp=True,
                      input_post=fuse)
    if ctx.gx_parts is not None:
        loss, coef = ops.gp_coef_parts(ctx.gx_parts, lambd)
    else:
        sq = ops.stat_allreduce(ops.sqnorm(g))      # whole-batch norm: summed over the ranks when statistics are synchronised
        loss, coef = ops.gp_coef(sq, lambd)
    return loss, (g, ops.scale_by(g, coef))


def disc_gp_second(ops, D: DiscNet, ctx, st, accumulate: bool, need_input_grad: bool):
    """Steps (3)+(4): tangent forward along v and the joint reverse sweep.  Parameter gradients of lambd * penalty are
    written (accumulate=False) or added (True) into the .dw / .dbias / .dgamma / .dbeta buffers; with need_input_grad
    the gradient with respect to xhat (= the primal cotangent at the input, NCHW fp32) is returned."""
    R = len(D.blocks)
    g, v = st
    xhat = ctx.x
    # (3) tangent forward along v
    at = ops.first_down_tangent(v, D.conv0, ctx.a[0], D.slope)
    ats, zts, s_zt, s_xhzt = [at], [None], [None], [None]
    for l in range(1, R + 1):
        cw, bn = D.blocks[l - 1]
        zt = ops.conv_down(at, cw, defer=1)         # split-K slabs are reduced by the tangent BatchNorm op (next)
        at, szt, sxz = ops.bn_tangent(ctx.z[l], zt, ctx.mean[l], ctx.invstd[l], bn.gamma, bn.beta, D.slope)
        zts.append(zt); ats.append(at); s_zt.append(szt); s_xhzt.append(sxz)
    # (4) joint reverse.  Head: t = sum_n lrelu'(h_n) * hdot_n  ->  dW_head = sum_n gh_n * at_R[n]
    with ops.side():
        ops.head_wgrad(ctx.gh, ats[R], D.head.dw, accumulate)
    qa = None
    for l in range(R, 0, -1):
        cw, bn = D.blocks[l - 1]
        pz = ops.bn_double_bwd(ctx.z[l], qa, zts[l], ctx.ga1[l], ctx.mean[l], ctx.invstd[l],
                               bn.gamma, bn.beta, D.slope, ctx.s_gy[l], ctx.s_gyxh[l],
                               s_zt[l], s_xhzt[l], bn.dgamma, bn.dbeta, accumulate)
        # dW = wgrad(pz, a_prev) + wgrad(gz1, at_prev): one launch, one split-K reduction
        with ops.side(pz):
            ops.conv_wgrad2(pz, ctx.a[l - 1], ctx.gz1[l], ats[l - 1], cw, accumulate)
        qa = ops.conv_up(pz, cw) if l > 1 else ops.conv_up(pz, cw, ctx.a[0], D.slope)
    if R == 0:
        # no BatchNorm block: the penalty is piecewise constant in the first layer's parameters except through the head
        raise NotImplementedError("gradient penalty needs at least one Conv+BN block (in_size >= 32)")
    p0 = qa
    with ops.side(p0, v):
        done = ops.skinny_wgrad(p0, xhat, D.conv0.dw, accumulate, dbias=D.conv0.dbias, dbias_accumulate=accumulate)
        ops.skinny_wgrad(ctx.gz1[0], v, D.conv0.dw, True)
    if not done:
        ops.col_sum(p0, D.conv0.dbias, accumulate)
    gx = ops.last_up(p0, D.conv0, None, False) if need_input_grad else None
    ops.join()
    return gx


def disc_forward_eval(ops, D: DiscNet, x_nchw):
    """D(x) with BatchNorm in EVAL mode (running statistics) -> (N,).  The reference never evaluates its
    discriminator (every call site is in train mode); provided so that ``discriminator.eval(); discriminator(x)``
    behaves like the nn.Module it replaces."""
    a = disc_features_eval(ops, D, x_nchw)
    _, out = ops.head_fwd(a, D.head, D.last_slope)
    return out


# --------------------------------------------------------------------------------------------
# Generator
# --------------------------------------------------------------------------------------------
def gen_forward(ops, G: GenNet, noise, update_running=True, keep=True):
    """G(z): ConvT(E->C0,k4,s1,p0)+BN+LReLU, R x [ConvT(k4,s2,p1)+BN+LReLU], ConvT+bias+Tanh."""
    ctx = _Ctx()
    ctx.noise = noise
    _refresh(ops, G.blocks)
    z = ops.g0_fwd(noise, G.g0)
    a, mean, invstd = _bn_forward(ops, z, G.bn0, G.slope, update_running)
    ctx.z, ctx.mean, ctx.invstd, ctx.a = [z], [mean], [invstd], [a]
    for l, (cw, bn) in enumerate(G.blocks):
        fused_last = not keep and l == len(G.blocks) - 1        # (that path takes the epilogue statistics, not slabs)
        z, st = ops.conv_up(a, cw, want_stats=True, defer=0 if fused_last else 1)
        if not keep and l == len(G.blocks) - 1:
            # nothing is kept for a backward pass: the last BatchNorm + LeakyReLU is applied inside the image layer
            img = ops.last_up_bn(z, st, bn, G.slope, G.last, G.last.bias, True, update_running)
            if img is not None:
                ctx.img = img
                return img, ctx
        a, mean, invstd = _bn_forward(ops, z, bn, G.slope, update_running, st)
        if keep:
            ctx.z.append(z); ctx.mean.append(mean); ctx.invstd.append(invstd); ctx.a.append(a)
    img = ops.last_up(a, G.last, G.last.bias, True)
    ctx.img = img
    return img, ctx


def _class_partials(st, m_half: int):
    """conv_up's column sums [rows][2][C] (class-major: [4 classes][row tiles]) if every class block splits into the two
    batch halves at a partial-row boundary, else None.  m_half: low-resolution pixels of one half."""
    if st is None:
        return None
    rows = st.shape[0]
    if rows % 8:
        return None
    per_class = rows // 4
    if (2 * m_half) % per_class or m_half % ((2 * m_half) // per_class):
        return None
    return st


def gen_forward_pair(ops, G: GenNet, noise2, update_running=True):
    """G(z) for TWO noise batches (noise2 = [z_a; z_b], nothing kept for a backward pass) as one double batch through the
    GEMM / conv layers, BatchNorm per half exactly as two consecutive forward calls would do it (own batch statistics, running
    statistics updated by the first half first).  The D-loss and the penalty step both need a fake batch from the SAME
    generator weights (src/wgan_loss.py:247 and :371): their two forwards become one.  Returns images [2n, 3, H, W]."""
    def run(bn):
        return (bn.running_mean, bn.running_var, bn.nbt) if update_running else (None, None, None)
    _refresh(ops, G.blocks)
    z = ops.g0_fwd(noise2, G.g0)
    a, _, _ = ops.bn_forward2(z, G.bn0.gamma, G.bn0.beta, G.slope, G.bn0.eps, G.bn0.momentum, *run(G.bn0))
    for l, (cw, bn) in enumerate(G.blocks):
        z, st = ops.conv_up(a, cw, want_stats=True, defer=0 if l == len(G.blocks) - 1 else 2)
        st = _class_partials(st, a.numel() // a.shape[-1] // 2)
        if l == len(G.blocks) - 1:
            img = ops.last_up_bn2(z, st, bn, G.slope, G.last, G.last.bias, True, update_running)
            if img is not None:
                return img
        a, _, _ = ops.bn_forward2(z, bn.gamma, bn.beta, G.slope, bn.eps, bn.momentum, *run(bn), partials=st, nblk=4)
    return ops.last_up(a, G.last, G.last.bias, True)


def upgen_forward(ops, G: UpGenNet, noise, update_running=True, keep=True):
    """DCGANUpGenerator forward (src/dcgan.py:85-99 through the blocks of :36-56,:76-84)."""
    ctx = _Ctx()
    ctx.noise = noise
    z = ops.g0_fwd(noise, G.g0)
    a, mean, invstd = _bn_forward(ops, z, G.bn0, G.slope, update_running)
    ctx.z, ctx.mean, ctx.invstd, ctx.a = [z], [mean], [invstd], [a]
    for cw, bn in G.blocks:
        z = ops.upconv3(a, cw, cw.bias)
        a, mean, invstd = _bn_forward(ops, z, bn, G.slope, update_running)
        if keep:
            ctx.z.append(z); ctx.mean.append(mean); ctx.invstd.append(invstd); ctx.a.append(a)
    ctx.a_last = a
    img = ops.upconv3(a, G.last, G.last.bias, out_nchw=True)       # no activation (src/dcgan.py:76-84)
    ctx.img = img
    return img, ctx


def upgen_backward(ops, G: UpGenNet, ctx, gimg, accumulate: bool, need_input_grad: bool = False):
    """Parameter gradients of the up-generator for d(loss)/d(img) = gimg (NCHW fp32) (+ d/d noise on request)."""
    R = len(G.blocks)
    ops.upconv3_wgrad(gimg, ctx.a[R], G.last, accumulate, gy_nchw=True)
    ops.nchw_chan_sum(gimg, G.last.dbias, accumulate)
    ga = ops.upconv3_bwd_data(gimg, G.last, gy_nchw=True)
    for l in range(R, 0, -1):
        cw, bn = G.blocks[l - 1]
        gz, _, _ = ops.bn_act_bwd(ctx.z[l], ga, ctx.mean[l], ctx.invstd[l], bn.gamma, bn.beta,
                                  G.slope, bn.dgamma, bn.dbeta, accumulate)
        ops.upconv3_wgrad(gz, ctx.a[l - 1], cw, accumulate)
        ops.col_sum(gz, cw.dbias, accumulate)      # the Conv2d keeps its bias in front of the BatchNorm (:50-51)
        ga = ops.upconv3_bwd_data(gz, cw)
    gz0, _, _ = ops.bn_act_bwd(ctx.z[0], ga, ctx.mean[0], ctx.invstd[0], G.bn0.gamma, G.bn0.beta,
                               G.slope, G.bn0.dgamma, G.bn0.dbeta, accumulate)
    ops.g0_wgrad(ctx.noise, gz0, G.g0.dw, accumulate)
    return ops.g0_bwd_data(gz0, G.g0) if need_input_grad else None


def _gen_fwd(ops, G, noise, **kw):
    return upgen_forward(ops, G, noise, **kw) if isinstance(G, UpGenNet) else gen_forward(ops, G, noise, **kw)


def _gen_bwd(ops, G, ctx, gimg, accumulate, need_input_grad=False):
    if isinstance(G, UpGenNet):
        return upgen_backward(ops, G, ctx, gimg, accumulate, need_input_grad)
    return gen_backward(ops, G, ctx, gimg, accumulate, need_input_grad)


def gen_backward(ops, G: GenNet, ctx, gimg, accumulate: bool, need_input_grad: bool = False, gzl=None, gzl_parts=None):
    """Parameter gradients of G for d(loss)/d(img) = gimg (NCHW fp32); with need_input_grad also d(loss)/d(noise)
    (N, E) -- off the reference's path (its noise never requires grad), one extra GEMM.  gzl / gzl_parts: the cotangent
    of the Tanh's input and its per-workgroup channel sums when the caller's kernel produced them (disc_backward
    input_post); gimg is not read then."""
    R = len(G.blocks)
    if gzl is None:
        gzl = ops.tanh_bwd(gimg, ctx.img)
    with ops.side(gzl):
        ops.skinny_wgrad(ctx.a[R], gzl, G.last.dw, accumulate)
    if gzl_parts is not None:
        ops.parts_chan_sum(gzl_parts, G.last.dbias, accumulate)
    else:
        ops.nchw_chan_sum(gzl, G.last.dbias, accumulate)
    ga = ops.first_down(gzl, G.last, None, 1.0)
    for l in range(R, 0, -1):
        cw, bn = G.blocks[l - 1]
        gz, _, _ = ops.bn_act_bwd(ctx.z[l], ga, ctx.mean[l], ctx.invstd[l], bn.gamma, bn.beta,
                                  G.slope, bn.dgamma, bn.dbeta, accumulate, keep_ga=False)
        with ops.side(gz):
            ops.conv_wgrad(ctx.a[l - 1], gz, cw, accumulate)
        # consumed by the BatchNorm backward of the layer below (next op): generator block l - 1, or G.0's BatchNorm
        nxt = (ctx.z[l - 1], ctx.mean[l - 1], ctx.invstd[l - 1]) + \
            ((G.blocks[l - 2][1].gamma, G.blocks[l - 2][1].beta) if l > 1 else (G.bn0.gamma, G.bn0.beta)) + (G.slope, 1)
        ga = ops.conv_down(gz, cw, defer=1, bn_bwd=nxt)
    gz0, _, _ = ops.bn_act_bwd(ctx.z[0], ga, ctx.mean[0], ctx.invstd[0], G.bn0.gamma, G.bn0.beta,
                               G.slope, G.bn0.dgamma, G.bn0.dbeta, accumulate, keep_ga=False)
    if not ops.g0_wgrad_deferred(ctx.noise, gz0, G.g0, accumulate):
        ops.g0_wgrad(ctx.noise, gz0, G.g0.dw, accumulate)
    gin = ops.g0_bwd_data(gz0, G.g0) if need_input_grad else None
    ops.join()
    return gin


# --------------------------------------------------------------------------------------------
# The three train_ops bodies (gradient part; the optimizer step is applied by the caller so that
# a data-parallel all-reduce can sit between the two)
# --------------------------------------------------------------------------------------------
# Each body is split into a PREFIX that reads only one of the two networks and the REST.  A data-parallel run
# keeps the previous train_op's gradient all-reduce and optimizer step of the OTHER network in flight while the
# prefix executes (losses._Runner.run_dp); a single process simply runs rest(prefix()).
def gen_loss_prefix(ops, G, noise):
    """G(z): reads the generator only."""
    return _gen_fwd(ops, G, noise)


def gen_loss_rest(ops, G, D: DiscNet, pre, grad_scale: float = 1.0):
    img, gctx = pre
    n = img.shape[0]
    out, dctx = disc_forward(ops, D, img)
    loss = ops.mean_diff(out, None, -1.0)
    # DCGAN generator: the cotangent of its Tanh (gimg * (1 - img^2)) and the channel sums of that (the last layer's bias
    # gradient) come out of the kernel that writes D's input gradient
    fuse = {"tanh_img": gctx.img} if isinstance(G, GenNet) else None
    gimg = disc_backward(ops, D, dctx, -grad_scale / n, wgrad=False, accumulate=False, need_input_grad=True, input_post=fuse)
    if dctx.gx_parts is not None:
        gen_backward(ops, G, gctx, None, accumulate=False, gzl=gimg, gzl_parts=dctx.gx_parts)
    else:
        _gen_bwd(ops, G, gctx, gimg, accumulate=False)
    return loss


def gen_loss_grads(ops, G: GenNet, D: DiscNet, noise, grad_scale: float = 1.0):
    """src/wgan_loss.py:113-126: loss = mean(-D(G(z))); fills G's gradients.  D's weight
    gradients, which the reference computes and discards, are not computed."""
    return gen_loss_rest(ops, G, D, gen_loss_prefix(ops, G, noise), grad_scale)


def disc_loss_prefix(ops, D: DiscNet, real):
    """D(real): reads the discriminator only."""
    return disc_forward(ops, D, real)


def disc_loss_prefix_bwd(ops, D: DiscNet, real, grad_scale: float = 1.0):
    """Data-parallel prefix of the D-loss step: D(real) forward AND its whole backward (the loss is a difference of two
    means and BatchNorm statistics are per call, so the real half's parameter gradients -mean'(D(real)) do not depend on
    the generator at all): everything here reads the discriminator only, so the generator's gradient all-reduce of the
    previous train_op (the largest collective of an iteration) stays in flight under a forward AND a backward pass.
    Parameter gradients are WRITTEN; disc_loss_rest_acc adds the fake half's."""
    out_r, ctx_r = disc_forward(ops, D, real)
    n = out_r.shape[0]
    disc_backward(ops, D, ctx_r, -grad_scale / n, wgrad=True, accumulate=False, need_input_grad=False)
    return out_r


def disc_loss_rest_acc(ops, G, D: DiscNet, out_r, noise, grad_scale: float = 1.0):
    """The rest of that step: G(z), D(fake), the loss, and the fake half's backward ACCUMULATED onto the real half's."""
    n = out_r.shape[0]
    img, _ = _gen_fwd(ops, G, noise, keep=False)
    out_f, ctx_f = disc_forward(ops, D, img)
    loss = ops.mean_diff(out_f, out_r, 1.0)
    disc_backward(ops, D, ctx_f, grad_scale / n, wgrad=True, accumulate=True, need_input_grad=False)
    return loss


def disc_loss_prefix_dgrad(ops, D: DiscNet, real, grad_scale: float = 1.0):
    """Data-parallel prefix, third form (RNAGAN_DP_PREFIX_BWD=2): D(real) forward and its DATA-gradient chain (BatchNorm
    backward, transposed convs, the small parameter gradients); the conv weight gradients wait for the fake half so that each
    layer's is one two-segment launch with one split-K reduction (disc_loss_rest_pairw) -- the prefix is shorter by those
    launches (less cover for the generator's all-reduce), the train_op loses five weight-gradient launches and reductions."""
    out_r, ctx_r = disc_forward(ops, D, real)
    n = out_r.shape[0]
    disc_backward(ops, D, ctx_r, -grad_scale / n, wgrad="defer", accumulate=False, need_input_grad=False)
    return out_r, ctx_r


def disc_loss_rest_pairw(ops, G, D: DiscNet, pre, noise, grad_scale: float = 1.0):
    out_r, ctx_r = pre
    n = out_r.shape[0]
    img, _ = _gen_fwd(ops, G, noise, keep=False)
    out_f, ctx_f = disc_forward(ops, D, img)
    loss = ops.mean_diff(out_f, out_r, 1.0)
    disc_backward(ops, D, ctx_f, grad_scale / n, wgrad=True, accumulate=True, need_input_grad=False, partner=ctx_r)
    return loss


def disc_loss_rest(ops, G, D: DiscNet, pre, noise, grad_scale: float = 1.0):
    out_r, ctx_r = pre
    n = out_r.shape[0]
    img, _ = _gen_fwd(ops, G, noise, keep=False)
    out_f, ctx_f = disc_forward(ops, D, img)
    loss = ops.mean_diff(out_f, out_r, 1.0)
    disc_backward_pair(ops, D, ctx_r, -grad_scale / n, ctx_f, grad_scale / n)
    return loss


def _half_partials(st, m_half: int, m_total: int, h: int):
    """Rows of the conv-epilogue statistics [rows][2][C] that belong to batch half h, or None when a partial row straddles
    the halves (then the BatchNorm pass reduces its half itself)."""
    if st is None:
        return None
    rows = st.shape[0]
    if rows % 2 or m_total % rows or m_half % (m_total // rows):
        return None
    return st[h * (rows // 2):(h + 1) * (rows // 2)]


def disc_loss_grads_batched(ops, G, D: DiscNet, real, noise, grad_scale: float = 1.0, next_noise=None):
    """The D-loss step with D(real) and D(fake) as ONE double batch through the conv layers (forward, data gradient and
    weight gradient: one launch each per layer instead of two, at a size where the 256 x 256-tile kernels need no split-K),
    BatchNorm per half exactly as two separate forward calls would do it (bn_forward2 / bn_act_bwd2: one set of launches over
    two batch groups; own batch statistics, running statistics updated by the real half first -- the reference's call order D(real), G(z), D(fake) of src/wgan_loss.py:241-253 -- own
    backward reductions, parameter gradients summed).  Same result as disc_loss_grads up to the kernels' tile shapes.
    next_noise: also produce G(next_noise) -- the fake batch of the penalty step that follows (same generator weights) -- in
    the same generator pass (gen_forward_pair); returns (loss, fake_next) then."""
    n = real.shape[0]
    R = len(D.blocks)
    fake_next = None
    if next_noise is not None and isinstance(G, GenNet):
        # the penalty step that follows needs a fake batch from the same generator weights: both forwards as one double batch
        img2 = gen_forward_pair(ops, G, torch.cat([noise, next_noise]))
        img, fake_next = img2[:n], img2[n:]
    else:
        img, _ = _gen_fwd(ops, G, noise, keep=False)
        if next_noise is not None:
            fake_next, _ = _gen_fwd(ops, G, next_noise, keep=False)
    xs = (real, img)
    _refresh(ops, D.blocks)
    H, W = real.shape[2], real.shape[3]
    C0 = D.conv0.w.shape[0]
    a = torch.empty((2 * n, H // 2, W // 2, C0), dtype=ops.act_dtype, device=real.device)
    # packed LeakyReLU sign bits of layer 0's output: only the 64-channel layer 0 under a 64 -> 128 layer 1 has that form
    bits = ops.sign_bits_for(2 * n, H // 2, W // 2, C0, D.blocks[0][0].O) if R > 0 else None
    for h in range(2):
        ops.first_down(xs[h], D.conv0, D.conv0.bias, D.slope,
                       out=(a[h * n:(h + 1) * n], None if bits is None else bits[h * n:(h + 1) * n]))
    if bits is not None:
        a._rg_sign_bits = bits
    acts, zs, means, invstds = [a], [None], [None], [None]
    for cw, bn in D.blocks:
        z, st = ops.conv_down(a, cw, want_stats=True, defer=2)
        m_half = z.numel() // z.shape[-1] // 2
        if _half_partials(st, m_half, 2 * m_half, 0) is None:
            st = None                                       # a partial row straddles the halves: BatchNorm reduces itself
        a, mean, invstd = ops.bn_forward2(z, bn.gamma, bn.beta, D.slope, bn.eps, bn.momentum, bn.running_mean,
                                          bn.running_var, bn.nbt, partials=st)
        zs.append(z); means.append(mean); invstds.append(invstd); acts.append(a)
    hh, out = ops.head_fwd(a, D.head, D.last_slope)
    loss = ops.mean_diff(out[n:], out[:n], 1.0)
    # backward: coefficient -1/n on the real half, +1/n on the fake half
    gh = torch.cat([ops.head_grad(hh[:n], -grad_scale / n, D.last_slope), ops.head_grad(hh[n:], grad_scale / n, D.last_slope)])
    with ops.side(gh):
        ops.head_wgrad(gh, acts[R], D.head.dw, False)
    ga = ops.head_bwd_data(gh, D.head)
    for l in range(R, 0, -1):
        cw, bn = D.blocks[l - 1]
        gz = ops.bn_act_bwd2(zs[l], ga, means[l], invstds[l], bn.gamma, bn.beta, D.slope, bn.dgamma, bn.dbeta, False)
        with ops.side(gz):
            ops.conv_wgrad(gz, acts[l - 1], cw, False)
        ga = ops.conv_up(gz, cw, defer=2, bn_bwd=_bnb(zs, means, invstds, D.blocks, l - 1, D.slope, 2)) if l > 1 \
            else ops.conv_up(gz, cw, acts[0], D.slope)
    gz0 = ga if R > 0 else ops.lrelu_bwd(ga, acts[0], D.slope)
    with ops.side(gz0):
        d0 = ops.skinny_wgrad(gz0[:n], xs[0], D.conv0.dw, False, dbias=D.conv0.dbias, dbias_accumulate=False)
        d1 = ops.skinny_wgrad(gz0[n:], xs[1], D.conv0.dw, True, dbias=D.conv0.dbias if d0 else None, dbias_accumulate=True)
    if not d0:
        ops.col_sum(gz0, D.conv0.dbias, False)
    elif not d1:
        ops.col_sum(gz0[n:], D.conv0.dbias, True)
    ops.join()
    return loss if next_noise is None else (loss, fake_next)


def disc_loss_grads(ops, G: GenNet, D: DiscNet, real, noise, grad_scale: float = 1.0):
    """src/wgan_loss.py:241-260: loss = mean(D(G(z).detach()) - D(real)); fills D's gradients.
    Forward order D(real), G(z), D(fake) as in the reference (BN running statistics)."""
    return disc_loss_rest(ops, G, D, disc_loss_prefix(ops, D, real), noise, grad_scale)


def gp_loss_prefix(ops, G, real, noise, eps):
    """fake = G(z); xhat = eps*real + (1-eps)*fake: reads the generator only."""
    img, _ = _gen_fwd(ops, G, noise, keep=False)
    return ops.interp(real, img, eps)


def gp_loss_prefix_fake(ops, real, fake, eps):
    """xhat from a fake batch that is already there (produced with the D-loss step's, see disc_loss_grads_batched)."""
    return ops.interp(real, fake, eps)


def gp_loss_rest(ops, D: DiscNet, xhat, lambd: float, grad_scale: float = 1.0):
    return disc_gradient_penalty(ops, D, xhat, lambd * grad_scale)


def gp_loss_grads(ops, G: GenNet, D: DiscNet, real, noise, eps: float, lambd: float, grad_scale: float = 1.0):
    """src/wgan_loss.py:371-387: fake = G(z); xhat = eps*real + (1-eps)*fake; D gradients of
    lambd*GP.  The generator gradients the reference produces here are never used and are skipped."""
    return gp_loss_rest(ops, D, gp_loss_prefix(ops, G, real, noise, eps), lambd, grad_scale)


# --------------------------------------------------------------------------------------------
# Building the engine view from nn.Modules that follow the torchgan key structure
# (model.{i}.0 = conv, model.{i}.1 = BatchNorm2d; disc.0 = head conv)
# --------------------------------------------------------------------------------------------
def _grad_of(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _slope_of(act, default):
    ns = getattr(act, "negative_slope", None)
    return float(ns) if ns is not None else default


def _bnp(bn):
    return BNP(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.num_batches_tracked,
               _grad_of(bn.weight), _grad_of(bn.bias), eps=bn.eps, momentum=bn.momentum)


def is_tap_major(t) -> bool:
    """4-D tensor of logical shape [O][I][4][4] whose storage order is [O][4][4][I]."""
    return t.dim() == 4 and not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous()


def _middle_convs(mod):
    """The stride-2 4x4 conv layers between the image-side layer and G.0 / the head (torchgan DCGAN recipe);
    none in the resize-convolution generator."""
    blocks = list(mod.model.children())
    sel = blocks[1:] if hasattr(mod, "disc") else blocks[1:-1]
    return [blk[0] for blk in sel if isinstance(blk[0], (torch.nn.Conv2d, torch.nn.ConvTranspose2d))
            and tuple(blk[0].kernel_size) == (4, 4)]


def tap_major_(mod):
    """Re-home the weights of the middle conv layers into tap-major storage w[O][4][4][I] (what the HIP conv
    kernels take, include/rnagan_hip.h).  The nn.Parameters keep their PyTorch shape [O][I][4][4] as strided
    views, so state_dict(), load_state_dict() and checkpoints are unchanged; only the memory order differs.
    Idempotent."""
    for conv in _middle_convs(mod):
        w = conv.weight
        if is_tap_major(w.data):
            continue
        g = w.grad
        w.data = w.data.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        if g is not None:
            w.grad = g.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return mod


def build_disc_net(mod) -> DiscNet:
    blocks = list(mod.model.children())
    c0 = blocks[0][0]
    conv0 = ConvW(c0.weight.data, c0.bias.data, _grad_of(c0.weight), _grad_of(c0.bias))
    slope = _slope_of(blocks[0][-1], 0.2)
    bl = []
    for blk in blocks[1:]:
        conv, bn = blk[0], blk[1]
        if conv.bias is not None or not isinstance(bn, torch.nn.BatchNorm2d):
            raise NotImplementedError("HIP path supports the batchnorm=True recipe only")
        bl.append((ConvW.from_param(conv.weight, _grad_of(conv.weight)), _bnp(bn)))
    hc = mod.disc[0]
    if hc.bias is not None:
        raise NotImplementedError("HIP path supports the batchnorm=True recipe only")
    head = ConvW(hc.weight.data, None, _grad_of(hc.weight))
    return DiscNet(conv0, bl, head, slope, _slope_of(mod.disc[-1], 0.2))


def build_gen_net(mod) -> GenNet:
    blocks = list(mod.model.children())
    c0, b0 = blocks[0][0], blocks[0][1]
    if c0.bias is not None or not isinstance(b0, torch.nn.BatchNorm2d):
        raise NotImplementedError("HIP path supports the batchnorm=True recipe only")
    g0 = ConvW(c0.weight.data, None, _grad_of(c0.weight))
    slope = _slope_of(blocks[0][-1], 0.2)
    bl = []
    for blk in blocks[1:-1]:
        conv, bn = blk[0], blk[1]
        bl.append((ConvW.from_param(conv.weight, _grad_of(conv.weight)), _bnp(bn)))
    lc = blocks[-1][0]
    if not isinstance(blocks[-1][-1], torch.nn.Tanh):
        raise NotImplementedError("HIP path supports last_nonlinearity=Tanh only")
    last = ConvW(lc.weight.data, lc.bias.data, _grad_of(lc.weight), _grad_of(lc.bias))
    return GenNet(g0, _bnp(b0), bl, last, slope)


def build_upgen_net(mod) -> UpGenNet:
    """Engine view of a DCGANUpGenerator-structured module: model.0 = [ConvT, BN, nl], model.{1..R} = [Upsample,
    ReflectionPad2d, Conv2d(3x3, bias), BN, nl], model.{R+1} = [Upsample, ReflectionPad2d, Conv2d(3x3, bias)]."""
    blocks = list(mod.model.children())
    c0, b0 = blocks[0][0], blocks[0][1]
    if c0.bias is not None or not isinstance(b0, torch.nn.BatchNorm2d):
        raise NotImplementedError("HIP path supports the batchnorm=True recipe only")
    g0 = ConvW(c0.weight.data, None, _grad_of(c0.weight))
    slope = _slope_of(blocks[0][-1], 0.2)

    def conv3(conv):
        if tuple(conv.kernel_size) != (3, 3) or conv.bias is None:
            raise NotImplementedError("resize-convolution blocks are Conv2d(3x3, bias=True)")
        return ConvW(conv.weight.data, conv.bias.data, _grad_of(conv.weight), _grad_of(conv.bias))
    bl = [(conv3(blk[2]), _bnp(blk[3])) for blk in blocks[1:-1]]
    return UpGenNet(g0, _bnp(b0), bl, conv3(blocks[-1][2]), slope)


def upgen_forward_eval(ops, G: UpGenNet, noise):
    def bn_eval(z, bn):
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        return ops.bn_act(z, bn.running_mean, invstd, bn.gamma, bn.beta, G.slope)
    a = bn_eval(ops.g0_fwd(noise, G.g0), G.bn0)
    for cw, bn in G.blocks:
        a = bn_eval(ops.upconv3(a, cw, cw.bias), bn)
    return ops.upconv3(a, G.last, G.last.bias, out_nchw=True)


def disc_features_eval(ops, D: DiscNet, x_nchw):
    """Discriminator trunk with BatchNorm in EVAL mode (running statistics): the activation in front of the head,
    NHWC (N, 4, 4, C).  Feature extractor of the Frechet-distance proxy (rna_gan_amd.fid)."""
    a = ops.first_down(x_nchw, D.conv0, D.conv0.bias, D.slope)
    for cw, bn in D.blocks:
        z = ops.conv_down(a, cw)
        invstd = torch.rsqrt(bn.running_var + bn.eps)     # C-length vector: host-side plumbing
        a = ops.bn_act(z, bn.running_mean, invstd, bn.gamma, bn.beta, D.slope)
    return a


def gen_forward_eval_fp8(ops, G: GenNet, noise):
    """Generator-only inference with fp8 (e4m3) weights and activations (BASELINE configs[4]): eval-mode BatchNorm folded,
    one kernel per block.  The chain starts in fp8 at the first layer and stays there while a layer's shape has an
    fp8 kernel (channels multiples of 128, enough rows); the remaining layers (for the reference generator: the
    128 -> 64 block and the image layer) run the fused bf16 kernels.  Returns (images NCHW fp32, number of fp8 layers)."""
    def folded(bn):
        scale = bn.gamma * torch.rsqrt(bn.running_var + bn.eps)
        return scale, bn.beta - bn.running_mean * scale
    N, E = noise.shape
    C0 = G.g0.w.shape[1]
    modes = [ops.fp8_supported(N, E, 16 * C0, 1)]
    hw = 4
    for cw, _ in G.blocks:
        modes.append(modes[-1] and ops.fp8_supported(N * hw * hw, 4 * cw.O, cw.I, 4))
        hw *= 2
    if not modes[0]:
        return gen_forward_eval(ops, G, noise), 0
    a = ops.g0_fwd_fp8(ops.cast_fp8(noise), G.g0, *folded(G.bn0), G.slope, out_fp8=modes[1] if len(modes) > 1 else False)
    for l, (cw, bn) in enumerate(G.blocks):
        if modes[l + 1]:
            nxt = modes[l + 2] if l + 2 < len(modes) else False
            a = ops.conv_up_fp8(a, cw, *folded(bn), G.slope, out_fp8=nxt)
        else:
            y = ops.conv_up_affine(a, cw, *folded(bn), G.slope)
            if y is None:
                invstd = torch.rsqrt(bn.running_var + bn.eps)
                y = ops.bn_act(ops.conv_up(a, cw), bn.running_mean, invstd, bn.gamma, bn.beta, G.slope)
            a = y
    return ops.last_up(a, G.last, G.last.bias, True), sum(modes)


def gen_forward_eval(ops, G: GenNet, noise, fused_epilogue=True):
    """Generator forward with BatchNorm in EVAL mode (running statistics), used for the per-epoch
    sample grid (torchgan Logger: generator.eval(); generator(test_noise)) and for generator-only inference.  On the
    bf16 MFMA path every Conv + BatchNorm + LeakyReLU block is ONE kernel (the folded affine and the activation sit in
    the conv epilogue: rg_conv_up_affine / rg_g0_fwd_affine); fused_epilogue=False keeps the conv -> bn_act pairs."""
    def bn_eval(z, bn):
        invstd = torch.rsqrt(bn.running_var + bn.eps)     # C-length vector: host-side plumbing
        return ops.bn_act(z, bn.running_mean, invstd, bn.gamma, bn.beta, G.slope)

    def folded(bn):                                       # eval-mode BatchNorm as y * scale + shift (C-length vectors)
        scale = bn.gamma * torch.rsqrt(bn.running_var + bn.eps)
        return scale, bn.beta - bn.running_mean * scale
    # (below ~128 samples the deep layers want split-K, which the fused epilogue excludes: 92 k vs 100 k imgs/s at 64)
    fused = getattr(ops, "conv_up_affine", None) is not None and fused_epilogue and noise.shape[0] >= 128
    a = ops.g0_fwd_affine(noise, G.g0, *folded(G.bn0), G.slope) if fused else None
    if a is None:
        a = bn_eval(ops.g0_fwd(noise, G.g0), G.bn0)
    for cw, bn in G.blocks:
        y = ops.conv_up_affine(a, cw, *folded(bn), G.slope) if fused else None     # one kernel per layer
        a = y if y is not None else bn_eval(ops.conv_up(a, cw), bn)
    return ops.last_up(a, G.last, G.last.bias, True)
