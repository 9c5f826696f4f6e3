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


def _finish(module, optimizer):
    """all-reduce (data parallel) -> optimizer step -> invalidate packed weights."""
    _reduce(module)
    _apply(module, optimizer)


def _reduce(module):
    ops, _ = module.runtime()
    D_.allreduce_sum_(module.flat.grad, compress=(ops.act_dtype == torch.bfloat16))


def _apply(module, optimizer):
    optimizer.step()
    if not hasattr(optimizer, "note_replayed"):      # rna_gan_amd.optim.Adam reports the change itself
        module.weights_changed()
    return module.flat.data[:1]          # a tensor result, so that this half can be a graph of its own


# gradient halves (no collective, no optimizer): capturable on their own in a data-parallel run
def _g_grads(generator, discriminator, noise):
    ops, gn, dn = _nets(generator, discriminator)
    return E.gen_loss_grads(ops, gn, dn, noise.contiguous().float(), grad_scale=D_.grad_scale())


def _d_grads(generator, discriminator, real, noise, clip):
    ops, gn, dn = _nets(generator, discriminator)
    if clip is not None:
        ops.clamp_(discriminator.flat.data, clip[0], clip[1])      # every D parameter (wgan_loss.py:213-215)
        discriminator.weights_changed()
    return E.disc_loss_grads(ops, gn, dn, real.contiguous().float(), noise.contiguous().float(),
                             grad_scale=D_.grad_scale())


def _gp_grads(generator, discriminator, real, noise, eps, lambd):
    ops, gn, dn = _nets(generator, discriminator)
    return E.gp_loss_grads(ops, gn, dn, real.contiguous().float(), noise.contiguous().float(),
                           eps if torch.is_tensor(eps) else float(eps), float(lambd), grad_scale=D_.grad_scale())


def _g_step(generator, discriminator, optimizer_generator, noise):
    loss = _g_grads(generator, discriminator, noise)
    _finish(generator, optimizer_generator)
    return loss


def _d_step(generator, discriminator, optimizer_discriminator, real, noise, clip):
    loss = _d_grads(generator, discriminator, real, noise, clip)
    _finish(discriminator, optimizer_discriminator)
    return loss


def _gp_step(generator, discriminator, optimizer_discriminator, real, noise, eps, lambd):
    """eps: python float or 1-element device tensor."""
    loss = _gp_grads(generator, discriminator, real, noise, eps, lambd)
    _finish(discriminator, optimizer_discriminator)
    return loss


def _dispatch(runner, key, grads_fn, full_fn, inputs, generator, discriminator, stepped, optimizer):
    """single process: one graph per train_op; data parallel: gradients graph, eager all-reduce, step graph"""
    mods = [generator, discriminator]
    if D_.active():
        return runner.run_dp(key, grads_fn, inputs, mods, stepped, optimizer)
    return runner.run(key, full_fn, inputs, mods, [optimizer])


class _Runner:
    """Executes a step body eagerly for the first calls, then from a captured HIP graph
    (rna_gan_amd.graphed.StepGraph).  One graph per (body, modules, optimizer, input shapes)."""

    def __init__(self):
        self._graphs = {}

    def __getstate__(self):          # loss objects are pickled into checkpoints; graphs are not state
        return {}

    def __setstate__(self, state):
        self._graphs = {}

    def run_dp(self, key, grads_fn, inputs, modules, stepped, optimizer):
        """Data-parallel form: graph(gradients) -> eager RCCL all-reduce -> graph(optimizer step)."""
        loss = self.run(key + ("grads",), grads_fn, inputs, modules, [], [])
        _reduce(stepped)
        self.run(key + ("apply",), lambda: _apply(stepped, optimizer), [], [], [optimizer], [stepped])
        return loss

    def run(self, key, fn, inputs, modules, optimizers, stepped=None):
        from . import graphed
        if not graphed.ENABLED:
            return fn(*inputs)
        if stepped is None:
            stepped = [o._module for o in optimizers if getattr(o, "_module", None) is not None]
        # the launch sequence depends on which packed weights are stale: one graph per pattern
        key = key + tuple(tuple(t.shape) for t in inputs) + tuple(id(m) for m in modules) + \
            tuple(id(o) for o in optimizers) + tuple(int(m.packs_stale()) for m in modules)
        sg = self._graphs.get(key)
        if sg is None:
            hip_opts = [o for o in optimizers if hasattr(o, "note_replayed")]
            if len(hip_opts) != len(optimizers):
                return fn(*inputs)          # a foreign optimizer keeps host-side state: no capture
            sg = self._graphs[key] = graphed.StepGraph(fn, inputs, modules, hip_opts, stepped)
        return sg(*inputs)


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
        return _dispatch(self._runner, ("g",), lambda nz: _g_grads(generator, discriminator, nz),
                         lambda nz: _g_step(generator, discriminator, optimizer_generator, nz), [noise],
                         generator, discriminator, generator, optimizer_generator)

    def train_ops(self, generator, discriminator, optimizer_generator, device, batch_size, labels=None):
        _check_labels(generator, discriminator, labels)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        return self.step(generator, discriminator, optimizer_generator, noise).item()


class WassersteinDiscriminatorLoss(DiscriminatorLoss):
    def __init__(self, reduction="mean", clip=None, override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self.clip = clip if isinstance(clip, (tuple, list)) and len(clip) > 1 else None
        self._runner = _Runner()

    def forward(self, fx, fgz):
        return wasserstein_discriminator_loss(fx, fgz, self.reduction)

    def step(self, generator, discriminator, optimizer_discriminator, real, noise):
        clip = self.clip
        return _dispatch(self._runner, ("d", clip), lambda r, nz: _d_grads(generator, discriminator, r, nz, clip),
                         lambda r, nz: _d_step(generator, discriminator, optimizer_discriminator, r, nz, clip),
                         [real, noise], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        batch_size = real_inputs.size(0)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        return self.step(generator, discriminator, optimizer_discriminator, real_inputs.to(device), noise).item()


class WassersteinGradientPenalty(DiscriminatorLoss):
    def __init__(self, reduction="mean", lambd=10.0, override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self.lambd = lambd
        self._runner = _Runner()

    def step(self, generator, discriminator, optimizer_discriminator, real, noise, eps):
        """eps: 1-element float32 device tensor (read inside the graph)."""
        lambd = self.lambd
        return _dispatch(self._runner, ("gp", lambd),
                         lambda r, nz, e: _gp_grads(generator, discriminator, r, nz, e, lambd),
                         lambda r, nz, e: _gp_step(generator, discriminator, optimizer_discriminator, r, nz, e, lambd),
                         [real, noise, eps], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        batch_size = real_inputs.size(0)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        eps = _pinned(1).uniform_(0.0, 1.0).to(device, non_blocking=True)   # CPU generator, as the reference
        return self.step(generator, discriminator, optimizer_discriminator, real_inputs.to(device), noise,
                         eps).item()


# ------------------------------------------------------------------------------------------------
# betaVAE-conditioned losses (--loss_type wganvae), src/wgan_loss.py
# ------------------------------------------------------------------------------------------------
class _VAEMixin:
    def _init_vae(self, checkpoint, rna_features, beta):
        self.betavae = betaVAE(rna_features, 2048, [6000, 4000, 2048], [4000, 6000], beta=beta)
        if checkpoint is not None:
            self.betavae.load_state_dict(torch.load(checkpoint, map_location="cpu"))
        self.betavae.eval()
        self._runner = _Runner()

    def _inputs(self, generator, real_inputs, device):
        """src/wgan_loss.py:94-101: rna to the device; u ~ U(-0.3, 0.3) drawn on the CPU generator."""
        batch_size = real_inputs["image"].size(0)
        if next(self.betavae.parameters()).device != torch.device(device):
            self.betavae = self.betavae.to(device)
        rna = real_inputs["rna_data"].to(device, non_blocking=True).float()
        # same generator consumption as torch.FloatTensor(bs, E).uniform_(-0.3, 0.3); drawn into pinned
        # memory so that the copy to the device is asynchronous
        u = _pinned(batch_size, generator.encoding_dims).uniform_(-0.3, 0.3).to(device, non_blocking=True)
        return rna, u

    def _noise(self, generator, rna, u):
        """src/wgan_loss.py:96-106: z_mean = betavae.encode(rna)[0]; noise = standardise_columns(u + z_mean)."""
        z, _, _ = self.betavae.encode(rna)
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
                         lambda r, uu: _g_grads(generator, discriminator, self._noise(generator, r, uu)),
                         lambda r, uu: _g_step(generator, discriminator, optimizer_generator,
                                               self._noise(generator, r, uu)),
                         [rna, u], generator, discriminator, generator, optimizer_generator)

    def train_ops(self, generator, discriminator, optimizer_generator, device, batch_size, real_inputs,
                  labels=None):
        _check_labels(generator, discriminator, labels)
        rna, u = self._inputs(generator, real_inputs, device)
        return self.step(generator, discriminator, optimizer_generator, rna, u).item()


class WassersteinDiscriminatorLossVAE(DiscriminatorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, beta=0.005, reduction="mean", clip=None, override_train_ops=None):
        super().__init__(checkpoint, rna_features)
        self.clip = clip if isinstance(clip, (tuple, list)) and len(clip) > 1 else None
        self._init_vae(checkpoint, rna_features, beta)

    def forward(self, fx, fgz):
        return wasserstein_discriminator_loss_vae(fx, fgz, self.reduction)

    def step(self, generator, discriminator, optimizer_discriminator, real, rna, u):
        clip = self.clip
        return _dispatch(self._runner, ("d", clip),
                         lambda x, r, uu: _d_grads(generator, discriminator, x, self._noise(generator, r, uu), clip),
                         lambda x, r, uu: _d_step(generator, discriminator, optimizer_discriminator, x,
                                                  self._noise(generator, r, uu), clip),
                         [real, rna, u], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        rna, u = self._inputs(generator, real_inputs, device)
        real = real_inputs["image"].to(device)
        return self.step(generator, discriminator, optimizer_discriminator, real, rna, u).item()


class WassersteinGradientPenaltyVAE(DiscriminatorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, reduction="mean", lambd=10.0, override_train_ops=None, beta=0.005):
        super().__init__(checkpoint, rna_features)
        self.lambd = lambd
        self.override_train_ops = override_train_ops
        self._init_vae(checkpoint, rna_features, beta)

    def step(self, generator, discriminator, optimizer_discriminator, real, rna, u, eps):
        lambd = self.lambd
        return _dispatch(self._runner, ("gp", lambd),
                         lambda x, r, uu, e: _gp_grads(generator, discriminator, x, self._noise(generator, r, uu), e,
                                                       lambd),
                         lambda x, r, uu, e: _gp_step(generator, discriminator, optimizer_discriminator, x,
                                                      self._noise(generator, r, uu), e, lambd),
                         [real, rna, u, eps], generator, discriminator, discriminator, optimizer_discriminator)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        rna, u = self._inputs(generator, real_inputs, device)
        real = real_inputs["image"].to(device)
        eps = _pinned(1).uniform_(0.0, 1.0).to(device, non_blocking=True)   # torch.rand(1) in the reference (:376)
        return self.step(generator, discriminator, optimizer_discriminator, real, rna, u, eps).item()
