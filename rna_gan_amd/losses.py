"""Wasserstein losses with the reference's ``train_ops`` plugin interface.

Two families, selected by ``--loss_type`` in the reference CLI (src/histopathology_gan.py:265-278):
  * ``wganvae``: WassersteinGeneratorLossVAE / WassersteinDiscriminatorLossVAE /
    WassersteinGradientPenaltyVAE -- reference source src/wgan_loss.py:47-129,131-263,266-389.
    Same constructor arguments, same ``train_ops`` argument names (the Trainer resolves them by
    name), same RNG calls (uniform noise drawn on the CPU generator, ``torch.rand(1)`` for eps), same
    return value (python float; the penalty is returned unweighted).
  * ``wgan``: the stock torchgan losses (third-party, recalled in SURVEY Appendix A): randn noise
    on the device, tensor ``real_inputs``, optional weight clamp.

What differs from the reference, deliberately: the gradient computation is the explicit HIP
sequencing in rna_gan_amd.engine instead of autograd; work the reference computes and throws away
(discriminator weight grads in the G step, generator grads in the GP step, betaVAE grads) is not
computed; in a data-parallel run the flat gradient buffer is all-reduced before the optimizer step.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import dist as D_
from . import engine as E
from .betavae import betaVAE


def reduce(x, reduction=None):
    if reduction == "mean":
        return torch.mean(x)
    if reduction == "sum":
        return torch.sum(x)
    return x


def wasserstein_generator_loss(fgz, reduction="mean"):
    return reduce(-1.0 * fgz, reduction)


def wasserstein_discriminator_loss(fx, fgz, reduction="mean"):
    return reduce(fgz - fx, reduction)


# the *_vae functional forms ignore their reduction argument (src/wgan_loss.py:24-29)
def wasserstein_generator_loss_vae(fgz, reduction="mean"):
    return reduce(-1.0 * fgz, "mean")


def wasserstein_discriminator_loss_vae(fx, fgz, reduction="mean"):
    return reduce(fgz - fx, "mean")


class _GradientPenaltyFn(torch.autograd.Function):
    """(||d sum(d_interpolate) / d interpolate||_2 - 1)^2 as a differentiable scalar, for discriminator outputs produced
    by the HIP discriminator under autograd (models._DiscForwardFn).  torch cannot differentiate through that node a
    second time, so this Function takes the node's saved forward context and runs the engine's explicit second-order
    pass (engine.disc_gp_first / disc_gp_second: tangent forward + joint reverse) in its own backward: parameter
    gradients are ADDED into the flat gradient buffer, the gradient with respect to ``interpolate`` is returned."""

    @staticmethod
    def forward(ctx, interpolate, d_interpolate, *params):
        node = d_interpolate.grad_fn
        module, dctx = getattr(node, "module", None), getattr(node, "dctx", None)
        if module is None or dctx is None:
            raise RuntimeError("wasserstein_gradient_penalty: d_interpolate must be the direct output of a rna_gan_amd "
                               "discriminator called on `interpolate` with gradients enabled")
        if dctx.x.data_ptr() != interpolate.data_ptr() and not torch.equal(dctx.x, interpolate.detach().float()):
            raise RuntimeError("wasserstein_gradient_penalty: d_interpolate was not computed from this `interpolate`")
        ops, net = module.runtime()
        loss, st = E.disc_gp_first(ops, net, dctx, 1.0)
        ctx.module, ctx.dctx, ctx.st = module, dctx, st
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        ops, net = ctx.module.runtime()
        g, v = ctx.st
        scale = float(gloss)                      # host sync: this is the public functional form, not the hot path
        v = v if scale == 1.0 else v * scale      # v is linear in the upstream gradient (v = dL/dg)
        gx = E.disc_gp_second(ops, net, ctx.dctx, (g, v), accumulate=True, need_input_grad=ctx.needs_input_grad[0])
        return (gx, None) + (None,) * (len(ctx.module._rt_flat.params))


def wasserstein_gradient_penalty(interpolate, d_interpolate, reduction="mean"):
    """torchgan.losses.functional.wasserstein_gradient_penalty / src/wgan_loss.py:32-44: the squared distance from 1 of
    the 2-norm (over the WHOLE batch) of d sum(d_interpolate) / d interpolate.  ``d_interpolate`` must come from a
    rna_gan_amd discriminator applied to ``interpolate`` (requires_grad=True) with gradients enabled; the result is
    a scalar that can be back-propagated (see _GradientPenaltyFn)."""
    fn = d_interpolate.grad_fn
    module = getattr(fn, "module", None)
    if module is None:
        raise RuntimeError("wasserstein_gradient_penalty: d_interpolate carries no rna_gan_amd discriminator graph "
                           "(call the discriminator on an `interpolate` that requires grad, gradients enabled)")
    gp = _GradientPenaltyFn.apply(interpolate, d_interpolate, *module._rt_flat.params)
    return reduce(gp, reduction)


def wasserstein_gradient_penalty_vae(interpolate, d_interpolate, reduction="mean"):
    """src/wgan_loss.py:32-44 (the reduction argument is ignored there: always the mean of the scalar)."""
    return wasserstein_gradient_penalty(interpolate, d_interpolate, "mean")


class GeneratorLoss(nn.Module):
    def __init__(self, reduction="mean", override_train_ops=None):
        super().__init__()
        self.reduction = reduction
        self.override_train_ops = override_train_ops
        self.arg_map = {}

    def set_arg_map(self, value):
        self.arg_map.update(value)


class DiscriminatorLoss(nn.Module):
    def __init__(self, reduction="mean", override_train_ops=None):
        super().__init__()
        self.reduction = reduction
        self.override_train_ops = override_train_ops
        self.arg_map = {}

    def set_arg_map(self, value):
        self.arg_map.update(value)


# ------------------------------------------------------------------------------------------------
# shared step bodies
# ------------------------------------------------------------------------------------------------
def _pinned(*shape):
    return torch.empty(*shape, dtype=torch.float32, pin_memory=torch.cuda.is_available())


def _check_labels(generator, discriminator, labels):
    if generator.label_type != "none" or discriminator.label_type != "none":
        raise NotImplementedError("label-conditioned GANs are not on the RNA-GAN DCGAN path")


def _nets(generator, discriminator):
    og, gn = generator.runtime()
    od, dn = discriminator.runtime()
    if og.act_dtype != od.act_dtype:
        raise RuntimeError("generator and discriminator must use the same precision")
    return og, gn, dn


def _is_half(ops):
    """The backend runs the 16-bit-storage MFMA kernels (bf16 build, or the fp16 build of the same sources)."""
    return ops.act_dtype in (torch.bfloat16, torch.float16)


def _wire_kind(ops):
    """compress argument of the data-parallel all-reduce: the gradients travel in the build's 16-bit type, or as fp32."""
    if not _is_half(ops) or (ops.half == "f16" and not D_.F16_WIRE):
        return False
    return ops.half


def _finish(module, optimizer):
    """all-reduce (data parallel) -> optimizer step -> invalidate packed weights."""
    _reduce(module)
    _apply(module, optimizer)


def _reduce(module):
    ops, _ = module.runtime()
    D_.allreduce_sum_(module.flat.grad, compress=_wire_kind(ops))


def _apply(module, optimizer):
    ops, _ = module.runtime()
    if ops.loss_scale != 1.0 and getattr(optimizer, "_module", None) is not module:
        # fp16: the gradients in module.flat.grad carry the loss scale; only rna_gan_amd.optim.Adam bound to the module unscales
        raise RuntimeError("fp16 precision needs rna_gan_amd.optim.Adam(...).bind(module): a foreign optimizer would step on "
                           "loss-scaled gradients")
    optimizer.step()
    if not hasattr(optimizer, "note_replayed"):      # rna_gan_amd.optim.Adam reports the change itself
        module.weights_changed()
    return module.flat.data[:1]          # a tensor result, so that this half can be a graph of its own


# Every train_op body is  rest(prefix(...))  where the PREFIX reads only ONE of the two networks (engine.*_prefix):
#   G step : prefix = conditioned noise + G(z)            (reads G)   rest = D(G(z)), backward through D and G
#   D step : prefix = (clip) + D(real) [+ noise]          (reads D)   rest = G(z), D(fake), backward through D
#   GP step: prefix = noise + G(z) + interpolation        (reads G)   rest = penalty forward / double backward in D
# In a data-parallel run the previous train_op's all-reduce and optimizer step of the OTHER network stay in flight
# while the prefix runs (run_dp below).
def _g_prefix(generator, discriminator, noise):
    ops, gn, _ = _nets(generator, discriminator)
    return E.gen_loss_prefix(ops, gn, noise.contiguous().float())


def _g_rest(generator, discriminator, pre):
    ops, gn, dn = _nets(generator, discriminator)
    return E.gen_loss_rest(ops, gn, dn, pre, grad_scale=D_.grad_scale() * ops.loss_scale)


# data-parallel D-loss step: D(real)'s backward belongs to the prefix too (it reads the discriminator only), so the generator's
# gradient all-reduce -- started by the G-loss step just before -- is covered by a forward and a backward pass
# (RNAGAN_DP_PREFIX_BWD=0: forward only, as in round 2)
# RNAGAN_DP_PREFIX_BWD=2 (default since round 4): forward + data-gradient chain in the prefix, the conv weight gradients of
# both halves as two-segment launches in the rest -- engine.disc_loss_prefix_dgrad.  One rank, same box, interleaved: 12.07 ms
# against 12.21 ms for mode 1 (single-process path 11.24-11.39); the prefix shrinks from ~1.3 to ~0.9 ms, still longer than the
# generator's 90 MB collective at any plausible bus bandwidth.  1: the whole backward of the real half in the prefix (round 3).
# PROVISIONAL default: the only evidence is a one-rank timing and a world-2 run over gloo on a shared GPU; mode 2's shorter
# prefix gives the generator's collective less cover, and tools/dp_first_run.sh A/Bs mode 1 on the first real multi-GPU node.
def _env_int(name, dflt, allowed):
    try:
        v = int(os.environ.get(name, "") or dflt)
    except ValueError:
        v = dflt
    return v if v in allowed else dflt


DP_PREFIX_MODE = _env_int("RNAGAN_DP_PREFIX_BWD", 2, (0, 1, 2))
DP_PREFIX_BWD = DP_PREFIX_MODE != 0
# Data-parallel ROUTE of a train_op.  "prefix" (default): graph(prefix) / graph(rest) / all-reduce started and left in flight
# under the next train_op's prefix, which reads the other network (_Runner.run_dp) -- it HIDES the collectives, and gives up what a
# single process gains from running a train_op as one body: D(real) + D(fake) as one double batch, the penalty step's fake batch
# out of the D-loss step's generator pass, graph boundaries (+0.6 ms of 10.4 at one rank, DESIGN).  "whole": the single
# process's body as ONE graph (gradients onto the wire), then the all-reduce, waited for at once, then the optimizer step --
# nothing is hidden and none of that is lost.  Which of the two is faster depends on the collectives' real duration: the first
# run on a multi-GPU node A/Bs them (tools/dp_first_run.sh).
DP_ROUTE = os.environ.get("RNAGAN_DP_ROUTE", "prefix")
if DP_ROUTE not in ("prefix", "whole"):
    DP_ROUTE = "prefix"


def _d_prefix(generator, discriminator, real, noise, clip):
    ops, _, dn = _nets(generator, discriminator)
    if clip is not None:
        ops.clamp_(discriminator.flat.data, clip[0], clip[1])      # every D parameter (wgan_loss.py:213-215)
        discriminator.weights_changed()
    if DP_PREFIX_MODE == 2 and D_.active():
        return ("dgrad", E.disc_loss_prefix_dgrad(ops, dn, real.contiguous().float(), grad_scale=D_.grad_scale() * ops.loss_scale)), \
            noise.contiguous().float()
    if DP_PREFIX_BWD and D_.active():
        return ("bwd", E.disc_loss_prefix_bwd(ops, dn, real.contiguous().float(), grad_scale=D_.grad_scale() * ops.loss_scale)), \
            noise.contiguous().float()
    return E.disc_loss_prefix(ops, dn, real.contiguous().float()), noise.contiguous().float()


def _d_batched(generator, discriminator, real, noise, clip, next_noise=None):
    """single process: D(real) and D(fake) as one double batch through the conv layers (engine.disc_loss_grads_batched);
    next_noise: the noise of the penalty step that follows -- its fake batch comes out of the same generator pass"""
    ops, gn, dn = _nets(generator, discriminator)
    if clip is not None:
        ops.clamp_(discriminator.flat.data, clip[0], clip[1])
        discriminator.weights_changed()
    return E.disc_loss_grads_batched(ops, gn, dn, real.contiguous().float(), noise.contiguous().float(),
                                     grad_scale=D_.grad_scale() * ops.loss_scale,
                                     next_noise=None if next_noise is None else next_noise.contiguous().float())


class _FakeCache:
    """G(noise) of the penalty step, produced ahead by the D-loss step of the same iteration (both need a fake batch from the
    same generator weights, src/wgan_loss.py:247 and :371; one generator pass over the double batch instead of two).  The
    entry is used only by a penalty step that is handed the very noise tensor it was produced for, with the generator
    unchanged in between."""

    def __init__(self):
        self.key, self.src, self.img = None, None, None

    @staticmethod
    def make_key(generator, noise_tensors):
        _, gn = generator.runtime()
        return (id(generator), gn.g0.version) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in noise_tensors)

    def put(self, generator, noise_tensors, img):
        self.key, self.src, self.img = self.make_key(generator, noise_tensors), list(noise_tensors), img

    def take(self, generator, noise_tensors):
        if self.key is None or self.key != self.make_key(generator, noise_tensors):
            return None
        img = self.img
        self.key, self.src, self.img = None, None, None
        return img


_FAKE = _FakeCache()
LOOKAHEAD = os.environ.get("RNAGAN_FAKE_LOOKAHEAD", "1") != "0"


_NEXT_NOISE = [None]            # Trainer flow: the penalty step's noise, drawn by the D-loss plugin right after its own
TRAINER_LOOKAHEAD = [False]     # set by Trainer when a gradient-penalty plugin directly follows the D-loss plugin


def _lookahead_ok():
    # (data parallel, route "whole": the generator's pending update is applied before the D-loss step starts, so its weights are
    # the same in the D-loss and the penalty step, as in a single process)
    return LOOKAHEAD and D_BATCHED and (not D_.active() or DP_ROUTE == "whole")


def _d_step_lookahead(runner, key, generator, discriminator, optimizer, clip, inputs, noise_fn, next_tensors):
    """D-loss step whose generator pass also produces the fake batch of the penalty step that follows (kept in _FAKE)."""
    body = _d_body(generator, discriminator, clip, noise_fn, lookahead=True)
    loss, fake_next = _dispatch(runner, key + ("lookahead",), body, inputs, generator, discriminator, discriminator, optimizer)
    _FAKE.put(generator, next_tensors, fake_next)
    return loss


def _gp_fake_body(generator, discriminator, lambd):
    """penalty step on a fake batch that is already there: inputs (real, fake, eps)"""
    def prefix(real, fake, eps):
        ops, _, _ = _nets(generator, discriminator)
        return E.gp_loss_prefix_fake(ops, real.contiguous().float(), fake, eps if torch.is_tensor(eps) else float(eps))
    return _Body(prefix, lambda xhat: _gp_rest(generator, discriminator, xhat, lambd), generator)


def _d_rest(generator, discriminator, pre):
    ops, gn, dn = _nets(generator, discriminator)
    fwd_real, noise = pre
    if isinstance(fwd_real, tuple) and fwd_real[0] == "bwd":       # the real half's backward ran in the prefix
        return E.disc_loss_rest_acc(ops, gn, dn, fwd_real[1], noise, grad_scale=D_.grad_scale() * ops.loss_scale)
    if isinstance(fwd_real, tuple) and fwd_real[0] == "dgrad":     # ... its data-gradient chain did; weight gradients pair up here
        return E.disc_loss_rest_pairw(ops, gn, dn, fwd_real[1], noise, grad_scale=D_.grad_scale() * ops.loss_scale)
    return E.disc_loss_rest(ops, gn, dn, fwd_real, noise, grad_scale=D_.grad_scale() * ops.loss_scale)


def _gp_prefix(generator, discriminator, real, noise, eps):
    ops, gn, _ = _nets(generator, discriminator)
    return E.gp_loss_prefix(ops, gn, real.contiguous().float(), noise.contiguous().float(),
                            eps if torch.is_tensor(eps) else float(eps))


def _gp_rest(generator, discriminator, xhat, lambd):
    ops, _, dn = _nets(generator, discriminator)
    return E.gp_loss_rest(ops, dn, xhat, float(lambd), grad_scale=D_.gp_grad_scale())


def _g_step(generator, discriminator, optimizer_generator, noise):
    """Eager G train_op body (gradients + all-reduce + optimizer step)."""
    loss = _g_body(generator, discriminator).grads(noise)
    _finish(generator, optimizer_generator)
    return loss


def _d_step(generator, discriminator, optimizer_discriminator, real, noise, clip):
    loss = _d_body(generator, discriminator, clip).grads(real, noise)
    _finish(discriminator, optimizer_discriminator)
    return loss


def _gp_step(generator, discriminator, optimizer_discriminator, real, noise, eps, lambd):
    """eps: python float or 1-element device tensor."""
    loss = _gp_body(generator, discriminator, lambd).grads(real, noise, eps)
    _finish(discriminator, optimizer_discriminator)
    return loss


class _Body:
    """One train_op: prefix(*inputs) -> pre ; rest(pre) -> loss ; which network the prefix reads.  whole (optional): the
    single-process form of the same train_op when it is not simply rest(prefix(...))."""

    def __init__(self, prefix, rest, prefix_reads, whole=None):
        self.prefix, self.rest, self.prefix_reads, self.whole = prefix, rest, prefix_reads, whole

    def grads(self, *inputs):
        if self.whole is not None and (not D_.active() or DP_ROUTE == "whole"):
            return self.whole(*inputs)
        return self.rest(self.prefix(*inputs))


def _g_body(generator, discriminator, noise_fn=None):
    nf = noise_fn or (lambda nz: nz)
    return _Body(lambda *a: _g_prefix(generator, discriminator, nf(*a)),
                 lambda pre: _g_rest(generator, discriminator, pre), generator)


def _d_body(generator, discriminator, clip, noise_fn=None, lookahead=False):
    nf = noise_fn or (lambda nz: nz)
    if lookahead:
        # inputs (real, *conditioning, noise, next_noise): the penalty step's fake batch comes out of this step's generator pass
        return _Body(None, None, discriminator,
                     lambda real, *a: _d_batched(generator, discriminator, real, nf(*a[:-1]), clip, nf(*a[:-2], a[-1])))
    whole = (lambda real, *a: _d_batched(generator, discriminator, real, nf(*a), clip)) if D_BATCHED else None
    return _Body(lambda real, *a: _d_prefix(generator, discriminator, real, nf(*a), clip),
                 lambda pre: _d_rest(generator, discriminator, pre), discriminator, whole)


def _gp_body(generator, discriminator, lambd, noise_fn=None):
    nf = noise_fn or (lambda nz: nz)
    return _Body(lambda real, *a: _gp_prefix(generator, discriminator, real, nf(*a[:-1]), a[-1]),
                 lambda xhat: _gp_rest(generator, discriminator, xhat, lambd), generator)


def _dispatch(runner, key, body, inputs, generator, discriminator, stepped, optimizer):
    """single process: one graph per train_op (gradients + optimizer step); data parallel: see _Runner.run_dp"""
    mods = [generator, discriminator]
    if D_.active():
        return runner.run_dp(key, body, inputs, mods, stepped, optimizer)

    def full(*a):
        g0 = _fusable_g0(stepped, optimizer)
        if g0 is not None:
            g0.fuse_step = True          # this gradient pass is followed at once by optimizer.step(): see ConvW.fuse_step
        deferred = _slab_layers(stepped, optimizer)
        for cw in deferred:
            cw.defer_slabs = True        # ... and the split-K weight gradients of its 4 x 4 layers may stay unreduced slabs
        sk = _skinny_slab_layer(stepped, optimizer)
        ops_s = generator.runtime()[0] if sk is not None else None      # the HipOps the engine runs BOTH networks on (_nets)
        # nothing deferred may be left over from an earlier pass that raised before its optimizer step consumed it (a stale
        # pending_wgrad would also keep that pass's operand activations alive)
        _drop_deferred(deferred, sk, g0)
        if sk is not None:
            ops_s._skinny_defer = {sk.dw.data_ptr(): sk}      # ... and the image-side layer's per-workgroup partials too
        try:
            loss = body.grads(*a)
        except BaseException:
            _drop_deferred(deferred, sk, g0)     # the step below does not run: the next train_op starts clean
            raise
        finally:
            if g0 is not None:
                g0.fuse_step = False
            for cw in deferred:
                cw.defer_slabs = False
            if sk is not None:
                ops_s._skinny_defer = None
        _finish(stepped, optimizer)
        return loss
    return runner.run(key, full, inputs, mods, [optimizer])


def _drop_deferred(deferred, sk, g0):
    """Forget what a gradient pass left for an optimizer step that will not consume it (see _dispatch.full)."""
    for cw in list(deferred) + [c for c in (sk, g0) if c is not None]:
        cw.pending_slabs = None
        cw.pending_bias = None
        cw.pending_wgrad = None


# split-K weight-gradient slabs summed inside the optimizer step instead of by a reduction launch per layer (single process,
# bf16 kernels, rna_gan_amd.optim.Adam bound to the stepped module): ConvW.defer_slabs / ops_hip._wgrad_slabs / rg_adam_step_slabs
SLAB_ADAM = os.environ.get("RNAGAN_SLAB_ADAM", "1") != "0"


def _slab_layers(stepped, optimizer):
    if (not SLAB_ADAM or D_.active() or not hasattr(optimizer, "note_replayed") or
            getattr(optimizer, "_module", None) is not stepped):
        return []
    ops, net = stepped.runtime()
    if not _is_half(ops) or not isinstance(net, (E.GenNet, E.DiscNet)) or ops.stat_reduce is not None:
        return []
    return [b[0] for b in net.blocks]


SKINNY_SLAB_ADAM = os.environ.get("RNAGAN_SKINNY_SLAB_ADAM", "1") != "0"


def _skinny_slab_layer(stepped, optimizer):
    """The stepped network's image-side 4 x 4 layer (D's first conv / G's last transposed conv: 64 <-> 3 channels, PyTorch-layout
    weight) when its weight gradient's per-workgroup partials may stay unreduced for the optimizer step (conditions of
    _slab_layers; the tensor must sit 16-byte aligned in the flat buffer)."""
    if not SKINNY_SLAB_ADAM or not _slab_layers(stepped, optimizer):
        return None
    _, net = stepped.runtime()
    cw = net.conv0 if isinstance(net, E.DiscNet) else net.last
    flat = stepped.flat
    off = (cw.w.data_ptr() - flat.data.data_ptr()) // 4
    if cw.dw is None or not cw.w.is_contiguous() or off % 4 or cw.w.numel() % 4 or cw.w.shape[0] != 64:
        return None
    if cw.bias is not None and ((cw.bias.data_ptr() - flat.data.data_ptr()) // 4) % 4:
        return None                     # (its bias-gradient partials become a segment of the same step: 16-byte aligned too)
    return cw


G0_ADAM = os.environ.get("RNAGAN_G0_ADAM", "1") != "0"


def _fusable_g0(stepped, optimizer):
    """The stepped module's layer-0 weight handle if its gradient may be formed inside the fused optimizer step: the
    generator (DCGAN recipe), stepped by rna_gan_amd.optim.Adam bound to it, bf16 kernels, single process."""
    if not G0_ADAM or D_.active() or not hasattr(optimizer, "note_replayed") or getattr(optimizer, "_module", None) is not stepped:
        return None
    ops, net = stepped.runtime()
    g0 = getattr(net, "g0", None)
    if g0 is None or not isinstance(net, E.GenNet) or not _is_half(ops):
        return None
    return g0


# D-loss step of a single process: D(real) and D(fake) as one double batch (RNAGAN_D_BATCHED=0: two forward / backward chains)
D_BATCHED = os.environ.get("RNAGAN_D_BATCHED", "1") != "0"

# data-parallel runs: the gradient all-reduce + optimizer step of the last train_op, not yet applied
_PENDING = [None]
OVERLAP = os.environ.get("RNAGAN_DP_OVERLAP", "1") != "0"
FUSED_WIDEN = os.environ.get("RNAGAN_DP_FUSED_WIDEN", "1") != "0"


class _Pending:
    def __init__(self, module, optimizer, handle, factors=None):
        self.module, self.optimizer, self.handle, self.factors = module, optimizer, handle, factors


# data parallel, bf16 wire: the 4 x 4 layers' weight gradients go onto the wire without an fp32 gradient in between -- a split-K
# launch leaves its (bf16) slabs for rg_grad_to_wire, a launch without split-K writes its bf16 tile into the wire slice
# (ConvW.wire_slot / ops_hip._wgrad_slabs); needs ONE weight-gradient launch per layer and pass (the paired prefix form)
DP_WIRE_DIRECT = os.environ.get("RNAGAN_DP_WIRE_DIRECT", "1") != "0"


def _dp_wire_layers(stepped, optimizer):
    """[(ConvW, wire slice)] of the stepped module's layers whose weight gradients may bypass the fp32 gradient buffer."""
    if not (DP_WIRE_DIRECT and SLAB_ADAM and D_.active() and FUSED_WIDEN and (DP_PREFIX_MODE == 2 or DP_ROUTE == "whole")) \
            or D_.sync_stats():
        return []
    if not hasattr(optimizer, "grad_wire") or getattr(optimizer, "_module", None) is not stepped:
        return []
    ops, net = stepped.runtime()
    if not _wire_kind(ops) or not isinstance(net, (E.GenNet, E.DiscNet)) or ops.stat_reduce is not None:
        return []
    flat = stepped.flat
    wire = D_.wire_for(flat.grad, ops.half)
    if wire is None:
        return []
    out = []
    for cw, _ in net.blocks:
        off = (cw.w.data_ptr() - flat.data.data_ptr()) // 4
        n = cw.w.numel()
        if cw.layout == "OHWI" and off % 8 == 0 and n % 4 == 0 and off >= 0 and off + n <= flat.data.numel():
            out.append((cw, wire[off:off + n]))
    return out


def _dp_wire_table(stepped, layers, head):
    """Segment table over flat[head:] from what the pass left on the layers (pending_slabs), or None when nothing was deferred."""
    flat = stepped.flat
    rows = []
    for cw, _ in layers:
        ps, cw.pending_slabs = cw.pending_slabs, None
        if ps is not None:
            rows.append(((cw.w.data_ptr() - flat.data.data_ptr()) // 4, cw.w.numel(), 0 if ps[0] is None else ps[0].data_ptr(),
                         ps[1], ps[2]))
    if not rows:
        return None
    rows.sort(key=lambda t: t[0])
    total, pos, table = flat.data.numel(), head, []
    for off, n, ptr, ns, sdt in rows:
        if off < pos:
            raise RuntimeError("rna_gan_amd: a deferred weight gradient overlaps the part of the flat buffer that does not travel "
                               "on the wire")
        if off > pos:
            table.append((pos - head, off - pos, 0, 0, 0))
        table.append((off - head, n, ptr, ns, sdt))
        pos = off + n
    if total > pos:
        table.append((pos - head, total - pos, 0, 0, 0))
    return table


def _dp_factor_g0(stepped, optimizer, batch):
    """Data parallel: (g0 handle, gathered-factor buffers) when the stepped module's layer-0 weight gradient can travel as
    FACTORS (dist.G0_FACTORS): DCGAN generator stepped by rna_gan_amd.optim.Adam bound to it, bf16 kernels, rank-local
    statistics, and the fused kernel takes K = world x batch."""
    if not (D_.G0_FACTORS and G0_ADAM and D_.active()) or D_.sync_stats():
        return None
    if not hasattr(optimizer, "note_replayed") or getattr(optimizer, "_module", None) is not stepped:
        return None
    ops, net = stepped.runtime()
    g0 = getattr(net, "g0", None)
    if g0 is None or not isinstance(net, E.GenNet) or not _is_half(ops):
        return None
    En, C = g0.w.shape[0], g0.w.shape[1]
    if (g0.w.data_ptr() - stepped.flat.data.data_ptr()) != 0:
        return None
    if not ops.lib.rg_g0_wgrad_adam_supported(D_.world_size() * batch, En, C, ops.dt):
        return None
    return g0, D_.factor_buffers(id(stepped), batch, En, C, ops.h16, stepped.flat.data.device), ops.dt


def flush():
    """Apply the optimizer step a data-parallel train_op may have left in flight (all-reduce started, update not
    applied yet).  Called before anything reads parameters outside the train_ops: state_dict(), forward(), save."""
    pend, _PENDING[0] = _PENDING[0], None
    if pend is None:
        return
    # rna_gan_amd.optim.Adam steps straight from the all-reduced bf16 wire buffer (no widening pass; .grad then keeps
    # the rank-local gradient); other optimizers get the averaged gradient back in .grad first
    wire = D_.wire_of(pend.handle) if FUSED_WIDEN and hasattr(pend.optimizer, "grad_wire") else None
    # (with gathered G.0 factors the handle's head is excluded from the wire; the tail is widened like any other gradient
    # when the optimizer does not step from the wire -- RNAGAN_DP_FUSED_WIDEN=0 or a foreign optimizer)
    D_.allreduce_finish(pend.handle, widen=wire is None)
    fac = pend.factors
    if fac is not None:
        g0, (z_all, gy_all, _, _), dt, works = fac
        for w in works:
            if w is not None:
                w.wait()
        g0.pending_wgrad = (z_all, gy_all, dt)     # every rank's samples: the fused step forms sum_r z_r^T gz0_r itself
    if hasattr(pend.optimizer, "grad_wire"):
        pend.optimizer.grad_wire = wire
    try:
        _APPLY_RUNNER.run(("apply", id(pend.module), wire is not None, fac is not None),
                          lambda: _apply(pend.module, pend.optimizer), [], [], [pend.optimizer], [pend.module])
    finally:
        if fac is not None:
            fac[0].pending_wgrad = None
    if hasattr(pend.optimizer, "grad_wire"):
        pend.optimizer.grad_wire = None


class _Runner:
    """Executes a step body eagerly for the first calls, then from a captured HIP graph
    (rna_gan_amd.graphed.StepGraph).  One graph per (body, modules, optimizer, input shapes)."""

    MAX_GRAPHS = 24

    def __init__(self):
        self._graphs = {}
        self._pre = {}

    def __getstate__(self):          # loss objects are pickled into checkpoints; graphs are not state
        return {}

    def __setstate__(self, state):
        self._graphs = {}
        self._pre = {}

    def run_dp(self, key, body, inputs, modules, stepped, optimizer):
        """Data-parallel form.  Collectives are never captured, so a train_op is graph(prefix) -> graph(rest) ->
        eager all-reduce START; the all-reduce is only waited for -- and the optimizer step applied (a graph of its
        own) -- by the NEXT train_op after it has enqueued its prefix, which reads the other network: the RCCL
        transfer over xGMI overlaps with that compute.  flush() applies a trailing update."""
        if DP_ROUTE == "whole":
            return self._run_dp_whole(key, body, inputs, modules, stepped, optimizer)
        pend = _PENDING[0]
        if pend is not None and (not OVERLAP or pend.module is body.prefix_reads):
            flush()                                   # the prefix needs the updated weights: no overlap possible
        sg_pre = self._step_graph(key + ("pre",), body.prefix, inputs, [body.prefix_reads], [], [])
        pre = sg_pre(*inputs) if sg_pre is not None else body.prefix(*inputs)
        flush()
        # the rest reads the prefix's outputs through a holder: they are the prefix graph's static outputs once it is
        # captured, so the rest graph is tied to that prefix graph and may only be captured after it
        holder = self._pre.setdefault((key, id(sg_pre)), {})
        holder["pre"] = pre
        # generator-loss step: layer 0's weight gradient travels as gathered factors (dist.G0_FACTORS): the backward copies
        # z / gz0 into this rank's slices of the gathered buffers instead of forming the 67 M-element product
        fac = _dp_factor_g0(stepped, optimizer, inputs[0].shape[0]) if stepped is body.prefix_reads and inputs else None

        wired = _dp_wire_layers(stepped, optimizer)
        head = fac[0].w.numel() if fac is not None else 0
        # the wire segment table describes the launches of ONE rest graph: it lives with the StepGraph whose function wrote it
        # (sg_rest.wire_cell below), not in a holder that every variant under this prefix (packs_stale, lr, buf_gen) overwrites
        cell = {}

        def rest_body():
            for cw, slot in wired:
                cw.defer_slabs, cw.wire_slot, cw.pending_slabs = True, slot, None
            try:
                out = body.rest(holder["pre"])
                # what the weight-gradient launches of this pass left: kept with the graph this call captures (a replay runs
                # no host code, its launches are these)
                cell["wire_table"] = _dp_wire_table(stepped, wired, head)
                return out
            finally:
                for cw, _ in wired:
                    cw.defer_slabs, cw.wire_slot, cw.pending_slabs = False, None, None

        def rest():
            if fac is None:
                return rest_body()
            g0, bufs = fac[0], fac[1]
            g0.fuse_step, g0.factor_stage = True, (bufs[2], bufs[3])
            try:
                out = rest_body()
                if g0.pending_wgrad != "staged":
                    raise RuntimeError("data-parallel G step: layer 0's weight gradient was not left as factors")
                return out
            finally:
                g0.fuse_step, g0.factor_stage, g0.pending_wgrad = False, None, None
        sg_rest = self._step_graph(key + ("rest", id(sg_pre), fac is not None, len(wired)), rest, [], modules, [], [])
        if sg_rest is not None:
            if getattr(sg_rest, "wire_cell", None) is None:
                sg_rest.wire_cell = cell              # created by this call: its function is the closure over this cell
            cell = sg_rest.wire_cell                  # (an existing graph runs / replays the closure of the call that created it)
            loss = sg_rest(allow_capture=sg_pre is not None and sg_pre.graph is not None)
        else:
            loss = rest()
        ops, _ = stepped.runtime()
        handle = D_.allreduce_start(stepped.flat.grad, compress=_wire_kind(ops), head=head,
                                    table=cell.get("wire_table") if wired else None)
        if fac is not None:
            z_all, gy_all, z_mine, gy_mine = fac[1]
            fac = fac + ([D_.allgather_start(z_all, z_mine), D_.allgather_start(gy_all, gy_mine)],)
        _PENDING[0] = _Pending(stepped, optimizer, handle, fac)
        if not OVERLAP:
            flush()
        return loss

    def _run_dp_whole(self, key, body, inputs, modules, stepped, optimizer):
        """Route "whole" (DP_ROUTE): graph(the single process's gradient body) -> all-reduce -> wait -> graph(optimizer step)."""
        flush()
        _, net = stepped.runtime()
        fac = _dp_factor_g0(stepped, optimizer, inputs[0].shape[0]) if isinstance(net, E.GenNet) and inputs else None
        wired = _dp_wire_layers(stepped, optimizer)
        head = fac[0].w.numel() if fac is not None else 0
        cell = {}

        def grads(*a):
            for cw, slot in wired:
                cw.defer_slabs, cw.wire_slot, cw.pending_slabs = True, slot, None
            if fac is not None:
                fac[0].fuse_step, fac[0].factor_stage = True, (fac[1][2], fac[1][3])
            try:
                out = body.grads(*a)
                cell["wire_table"] = _dp_wire_table(stepped, wired, head)
                if fac is not None and fac[0].pending_wgrad != "staged":
                    raise RuntimeError("data-parallel G step: layer 0's weight gradient was not left as factors")
                return out
            finally:
                for cw, _ in wired:
                    cw.defer_slabs, cw.wire_slot, cw.pending_slabs = False, None, None
                if fac is not None:
                    fac[0].fuse_step, fac[0].factor_stage, fac[0].pending_wgrad = False, None, None
        sg = self._step_graph(key + ("whole", fac is not None, len(wired)), grads, inputs, modules, [], [])
        if sg is not None:
            if getattr(sg, "wire_cell", None) is None:
                sg.wire_cell = cell
            cell = sg.wire_cell
            loss = sg(*inputs)
        else:
            loss = grads(*inputs)
        ops, _ = stepped.runtime()
        handle = D_.allreduce_start(stepped.flat.grad, compress=_wire_kind(ops), head=head,
                                    table=cell.get("wire_table") if wired else None)
        if fac is not None:
            z_all, gy_all, z_mine, gy_mine = fac[1]
            fac = fac + ([D_.allgather_start(z_all, z_mine), D_.allgather_start(gy_all, gy_mine)],)
        _PENDING[0] = _Pending(stepped, optimizer, handle, fac)
        flush()                                   # nothing to overlap with: wait and apply now
        return loss

    def run(self, key, fn, inputs, modules, optimizers, stepped=None):
        sg = self._step_graph(key, fn, inputs, modules, optimizers, stepped)
        return sg(*inputs) if sg is not None else fn(*inputs)

    def _step_graph(self, key, fn, inputs, modules, optimizers, stepped=None):
        """The StepGraph for this body / launch-sequence variant, or None when it has to run eagerly."""
        from . import graphed
        if not graphed.ENABLED or D_.sync_stats():       # synchronised statistics put collectives inside the step
            return None
        if stepped is None:
            stepped = [o._module for o in optimizers if getattr(o, "_module", None) is not None]
        hip_opts = [o for o in optimizers if hasattr(o, "note_replayed")]
        if len(hip_opts) != len(optimizers):
            return None                     # a foreign optimizer keeps host-side state: no capture
        for o in hip_opts:
            o._ensure()                     # moment / step buffers exist (and their generation is final) before keying
        # A captured graph freezes (a) the launch sequence, which depends on which packed weights are stale, (b) the raw
        # addresses of the flat parameter / gradient / shadow buffers and of Adam's moment / step buffers, (c) the
        # optimizer hyper-parameters passed as kernel arguments.  All three are part of the key: a re-homed module
        # (.to(), load into new storage), re-allocated optimizer state or a scheduler's new lr gets a new graph
        # instead of replaying one that updates freed buffers or steps with the old lr.
        key = key + tuple(tuple(t.shape) for t in inputs) + tuple(id(m) for m in modules) + \
            tuple(id(o) for o in optimizers) + tuple(int(m.packs_stale()) for m in modules) + \
            tuple(m.flat.gen for m in modules) + \
            tuple((o.buf_gen, float(o.param_groups[0]["lr"]), tuple(float(b) for b in o.param_groups[0]["betas"]),
                   float(o.param_groups[0]["eps"]), float(o.param_groups[0].get("weight_decay", 0.0))) for o in hip_opts)
        sg = self._graphs.get(key)
        if sg is None:
            while len(self._graphs) >= self.MAX_GRAPHS:          # e.g. one lr value per epoch under a scheduler
                self._graphs.pop(next(iter(self._graphs)))
            sg = self._graphs[key] = graphed.StepGraph(fn, inputs, modules, hip_opts, stepped)
        return sg


_APPLY_RUNNER = _Runner()
D_.set_flush_hook(flush)


# ------------------------------------------------------------------------------------------------
# stock losses (--loss_type wgan)
# ------------------------------------------------------------------------------------------------
class WassersteinGeneratorLoss(GeneratorLoss):
    def __init__(self, reduction="mean", override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self._runner = _Runner()

    def forward(self, fgz):
        return wasserstein_generator_loss(fgz, self.reduction)

    def step(self, generator, discriminator, optimizer_generator, noise):
        """The train_op body on explicit inputs; returns the loss as a 1-element device tensor."""
        return _dispatch(self._runner, ("g",), _g_body(generator, discriminator), [noise],
                         generator, discriminator, generator, optimizer_generator)

    def train_ops_async(self, generator, discriminator, optimizer_generator, device, batch_size, labels=None):
        _check_labels(generator, discriminator, labels)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        return self.step(generator, discriminator, optimizer_generator, noise)

    def train_ops(self, generator, discriminator, optimizer_generator, device, batch_size, labels=None):
        return self.train_ops_async(generator, discriminator, optimizer_generator, device, batch_size, labels).item()
    train_ops._rg_async = "train_ops_async"


class WassersteinDiscriminatorLoss(DiscriminatorLoss):
    def __init__(self, reduction="mean", clip=None, override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self.clip = clip if isinstance(clip, (tuple, list)) and len(clip) > 1 else None
        self._runner = _Runner()

    def forward(self, fx, fgz):
        return wasserstein_discriminator_loss(fx, fgz, self.reduction)

    def step(self, generator, discriminator, optimizer_discriminator, real, noise, next_noise=None):
        """next_noise: the noise tensor the gradient-penalty step of this iteration will be called with; its fake batch is
        then produced by this step's generator pass (same generator weights) and picked up by that step."""
        clip = self.clip
        if next_noise is not None and _lookahead_ok():
            return _d_step_lookahead(self._runner, ("d", clip), generator, discriminator, optimizer_discriminator, clip,
                                     [real, noise, next_noise], None, (next_noise,))
        return _dispatch(self._runner, ("d", clip), _d_body(generator, discriminator, clip),
                         [real, noise], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops_async(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        batch_size = real_inputs.size(0)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        nxt = None
        if TRAINER_LOOKAHEAD[0] and _lookahead_ok():       # the penalty plugin's randn, drawn now (same generator order)
            nxt = _NEXT_NOISE[0] = torch.randn(batch_size, generator.encoding_dims, device=device)
        return self.step(generator, discriminator, optimizer_discriminator, real_inputs.to(device), noise, nxt)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        return self.train_ops_async(generator, discriminator, optimizer_discriminator, real_inputs, device, labels).item()
    train_ops._rg_async = "train_ops_async"


class WassersteinGradientPenalty(DiscriminatorLoss):
    def __init__(self, reduction="mean", lambd=10.0, override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self.lambd = lambd
        self._runner = _Runner()

    def forward(self, interpolate, d_interpolate):
        """torchgan WassersteinGradientPenalty.forward: the unweighted penalty as a differentiable scalar."""
        return wasserstein_gradient_penalty(interpolate, d_interpolate, self.reduction)

    def step(self, generator, discriminator, optimizer_discriminator, real, noise, eps):
        """eps: 1-element float32 device tensor (read inside the graph)."""
        lambd = self.lambd
        fake = _FAKE.take(generator, (noise,)) if (not D_.active() or DP_ROUTE == "whole") else None
        if fake is not None:                  # G(noise) came out of the D-loss step's generator pass
            return _dispatch(self._runner, ("gpf", lambd), _gp_fake_body(generator, discriminator, lambd),
                             [real, fake, eps], generator, discriminator, discriminator, optimizer_discriminator)
        return _dispatch(self._runner, ("gp", lambd), _gp_body(generator, discriminator, lambd),
                         [real, noise, eps], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops_async(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        batch_size = real_inputs.size(0)
        noise, _NEXT_NOISE[0] = _NEXT_NOISE[0], None       # drawn ahead by the D-loss plugin (Trainer flow), else now
        if noise is None or noise.shape != (batch_size, generator.encoding_dims):
            noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        eps = _pinned(1).uniform_(0.0, 1.0).to(device, non_blocking=True)   # CPU generator, as the reference
        return self.step(generator, discriminator, optimizer_discriminator, real_inputs.to(device), noise, eps)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        return self.train_ops_async(generator, discriminator, optimizer_discriminator, real_inputs, device, labels).item()
    train_ops._rg_async = "train_ops_async"


# ------------------------------------------------------------------------------------------------
# betaVAE-conditioned losses (--loss_type wganvae), src/wgan_loss.py
# ------------------------------------------------------------------------------------------------
class _LatentCache:
    """z_mean = betaVAE.encode(rna)[0] of the CURRENT batch, shared by the three loss plugins.

    The reference builds three frozen betaVAE copies from one checkpoint (src/wgan_loss.py:67-69, :159-161, :289-291)
    and encodes the same RNA rows three times per iteration (:96-97, :223-224, :353-354); the result is the same
    tensor each time.  Here the first train_op of a batch encodes (its own small HIP graph), the other two reuse the
    latent when (a) they are handed the same RNA tensor (the cache keeps a reference to it, so identity by address +
    version counter is sound) and (b) their encoder holds the same weights BY PROVENANCE (betaVAE.signature(): the token of
    the checkpoint file / state_dict the weights were loaded from, valid while no tensor of the module has been written
    since -- not a fingerprint of the values).  ``new_batch()`` (called by
    Trainer.train_iter and by bench.py at the start of every iteration) drops the entry, so a latent is never carried
    from one iteration to the next even when the caller reuses one input tensor."""

    def __init__(self):
        self.key = None
        self.src = None
        self.z = None
        self.hits = 0
        self.misses = 0

    def clear(self):
        self.key, self.src, self.z = None, None, None


_LATENT = _LatentCache()
LATENT_CACHE = os.environ.get("RNAGAN_LATENT_CACHE", "1") != "0"


def _drop_lookahead():
    _NEXT_NOISE[0] = None
    _FAKE.key, _FAKE.src, _FAKE.img = None, None, None


def new_batch():
    """Start of a new batch: forget the cached conditioning latent (and anything prepared ahead for a step that never ran)."""
    _LATENT.clear()
    _drop_lookahead()


class _VAEMixin:
    def _init_vae(self, checkpoint, rna_features, beta):
        self.betavae = betaVAE(rna_features, 2048, [6000, 4000, 2048], [4000, 6000], beta=beta)
        if checkpoint is not None:
            self.betavae.load_checkpoint_file(checkpoint)      # weights token = the file's identity (_LatentCache)
        self.betavae.eval()
        self._runner = _Runner()
        self._enc_runner = _Runner()

    def _inputs(self, generator, real_inputs, device):
        """src/wgan_loss.py:94-101: rna (to the device, once per batch); u ~ U(-0.3, 0.3) drawn on the CPU generator."""
        batch_size = real_inputs["image"].size(0)
        if next(self.betavae.parameters()).device != torch.device(device):
            self.betavae = self.betavae.to(device)
        # same generator consumption as torch.FloatTensor(bs, E).uniform_(-0.3, 0.3); drawn into pinned
        # memory so that the copy to the device is asynchronous
        u = _pinned(batch_size, generator.encoding_dims).uniform_(-0.3, 0.3).to(device, non_blocking=True)
        return real_inputs["rna_data"], u

    def _latent(self, rna):
        """z_mean = betavae.encode(rna)[0] (src/wgan_loss.py:96-97) for this batch: from the per-batch cache, or
        encoded now (rna: host or device tensor)."""
        dev = next(self.betavae.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the betaVAE of a loss plugin must live on the GPU (call train_ops / move it with .to())")
        key = (rna.data_ptr(), rna._version, tuple(rna.shape), rna.dtype, rna.device, self.betavae.signature())
        if LATENT_CACHE and _LATENT.key == key:
            _LATENT.hits += 1
            return _LATENT.z
        _LATENT.misses += 1
        rna_dev = rna.to(dev, non_blocking=True).float().contiguous()
        z = self._enc_runner.run(("enc",), lambda r: self.betavae.encode(r, mean_only=True)[0], [rna_dev], [], [])
        _LATENT.key, _LATENT.src, _LATENT.z = key, rna, z
        return z

    def _noise(self, generator, z, u):
        """src/wgan_loss.py:100-106: noise = standardise_columns(u + z_mean)."""
        ops, _ = generator.runtime()
        return ops.latent_prep(u, z)


class WassersteinGeneratorLossVAE(GeneratorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, beta=0.005):
        # the reference passes (checkpoint, rna_features) positionally to the base class, so the path
        # lands in .reduction and rna_features in .override_train_ops (src/wgan_loss.py:63-66); kept.
        super().__init__(checkpoint, rna_features)
        self._init_vae(checkpoint, rna_features, beta)

    def forward(self, fgz):
        return wasserstein_generator_loss_vae(fgz, self.reduction)

    def step(self, generator, discriminator, optimizer_generator, rna, u):
        return _dispatch(self._runner, ("g",),
                         _g_body(generator, discriminator, lambda z, uu: self._noise(generator, z, uu)),
                         [self._latent(rna), u], generator, discriminator, generator, optimizer_generator)

    def train_ops_async(self, generator, discriminator, optimizer_generator, device, batch_size, real_inputs,
                        labels=None):
        _check_labels(generator, discriminator, labels)
        rna, u = self._inputs(generator, real_inputs, device)
        return self.step(generator, discriminator, optimizer_generator, rna, u)

    def train_ops(self, generator, discriminator, optimizer_generator, device, batch_size, real_inputs,
                  labels=None):
        return self.train_ops_async(generator, discriminator, optimizer_generator, device, batch_size, real_inputs,
                                    labels).item()
    train_ops._rg_async = "train_ops_async"


class WassersteinDiscriminatorLossVAE(DiscriminatorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, beta=0.005, reduction="mean", clip=None, override_train_ops=None):
        super().__init__(checkpoint, rna_features)
        self.clip = clip if isinstance(clip, (tuple, list)) and len(clip) > 1 else None
        self._init_vae(checkpoint, rna_features, beta)

    def forward(self, fx, fgz):
        return wasserstein_discriminator_loss_vae(fx, fgz, self.reduction)

    def step(self, generator, discriminator, optimizer_discriminator, real, rna, u, next_u=None):
        """next_u: the uniform draw the gradient-penalty step of this iteration will be called with (same RNA batch): its fake
        batch is then produced by this step's generator pass and picked up by that step."""
        clip = self.clip
        nf = lambda z, uu: self._noise(generator, z, uu)
        if next_u is not None and _lookahead_ok():
            return _d_step_lookahead(self._runner, ("d", clip), generator, discriminator, optimizer_discriminator, clip,
                                     [real, self._latent(rna), u, next_u], nf, (next_u,))
        return _dispatch(self._runner, ("d", clip), _d_body(generator, discriminator, clip, nf),
                         [real, self._latent(rna), u], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops_async(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        rna, u = self._inputs(generator, real_inputs, device)
        nxt = None
        if TRAINER_LOOKAHEAD[0] and _lookahead_ok():       # the penalty plugin's draw, made now (same generator order)
            nxt = _NEXT_NOISE[0] = self._inputs(generator, real_inputs, device)[1]
        real = real_inputs["image"].to(device)
        return self.step(generator, discriminator, optimizer_discriminator, real, rna, u, nxt)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        return self.train_ops_async(generator, discriminator, optimizer_discriminator, real_inputs, device, labels).item()
    train_ops._rg_async = "train_ops_async"


class WassersteinGradientPenaltyVAE(DiscriminatorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, reduction="mean", lambd=10.0, override_train_ops=None, beta=0.005):
        super().__init__(checkpoint, rna_features)
        self.lambd = lambd
        self.override_train_ops = override_train_ops
        self._init_vae(checkpoint, rna_features, beta)

    def forward(self, interpolate, d_interpolate):
        """src/wgan_loss.py:293-312 (``self.reduction`` holds the checkpoint path there and is ignored by the
        functional form, :44)."""
        return wasserstein_gradient_penalty_vae(interpolate, d_interpolate, self.reduction)

    def step(self, generator, discriminator, optimizer_discriminator, real, rna, u, eps):
        lambd = self.lambd
        fake = _FAKE.take(generator, (u,)) if (not D_.active() or DP_ROUTE == "whole") else None
        if fake is not None:                  # G(noise(u)) came out of the D-loss step's generator pass
            return _dispatch(self._runner, ("gpf", lambd), _gp_fake_body(generator, discriminator, lambd),
                             [real, fake, eps], generator, discriminator, discriminator, optimizer_discriminator)
        return _dispatch(self._runner, ("gp", lambd),
                         _gp_body(generator, discriminator, lambd, lambda z, uu: self._noise(generator, z, uu)),
                         [real, self._latent(rna), u, eps], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops_async(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        ahead, _NEXT_NOISE[0] = _NEXT_NOISE[0], None       # u drawn ahead by the D-loss plugin (Trainer flow), else now
        if ahead is not None and ahead.shape == (real_inputs["image"].size(0), generator.encoding_dims):
            rna, u = real_inputs["rna_data"], ahead
        else:
            rna, u = self._inputs(generator, real_inputs, device)
        real = real_inputs["image"].to(device)
        eps = _pinned(1).uniform_(0.0, 1.0).to(device, non_blocking=True)   # torch.rand(1) in the reference (:376)
        return self.step(generator, discriminator, optimizer_discriminator, real, rna, u, eps)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        return self.train_ops_async(generator, discriminator, optimizer_discriminator, real_inputs, device, labels).item()
    train_ops._rg_async = "train_ops_async"
