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
    D_.allreduce_sum_(module.flat.grad)
    optimizer.step()
    module.weights_changed()


def _g_step(generator, discriminator, optimizer_generator, noise):
    ops, gn, dn = _nets(generator, discriminator)
    loss = E.gen_loss_grads(ops, gn, dn, noise.contiguous().float(), grad_scale=D_.grad_scale())
    _finish(generator, optimizer_generator)
    return loss


def _d_step(generator, discriminator, optimizer_discriminator, real, noise, clip):
    ops, gn, dn = _nets(generator, discriminator)
    if clip is not None:
        ops.clamp_(discriminator.flat.data, clip[0], clip[1])      # every D parameter (wgan_loss.py:213-215)
        discriminator.weights_changed()
    loss = E.disc_loss_grads(ops, gn, dn, real.contiguous().float(), noise.contiguous().float(),
                             grad_scale=D_.grad_scale())
    _finish(discriminator, optimizer_discriminator)
    return loss


def _gp_step(generator, discriminator, optimizer_discriminator, real, noise, eps, lambd):
    ops, gn, dn = _nets(generator, discriminator)
    loss = E.gp_loss_grads(ops, gn, dn, real.contiguous().float(), noise.contiguous().float(), float(eps),
                           float(lambd), grad_scale=D_.grad_scale())
    _finish(discriminator, optimizer_discriminator)
    return loss


# ------------------------------------------------------------------------------------------------
# stock losses (--loss_type wgan)
# ------------------------------------------------------------------------------------------------
class WassersteinGeneratorLoss(GeneratorLoss):
    def forward(self, fgz):
        return wasserstein_generator_loss(fgz, self.reduction)

    def train_ops(self, generator, discriminator, optimizer_generator, device, batch_size, labels=None):
        _check_labels(generator, discriminator, labels)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        return _g_step(generator, discriminator, optimizer_generator, noise).item()


class WassersteinDiscriminatorLoss(DiscriminatorLoss):
    def __init__(self, reduction="mean", clip=None, override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self.clip = clip if isinstance(clip, (tuple, list)) and len(clip) > 1 else None

    def forward(self, fx, fgz):
        return wasserstein_discriminator_loss(fx, fgz, self.reduction)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        batch_size = real_inputs.size(0)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        return _d_step(generator, discriminator, optimizer_discriminator, real_inputs.to(device), noise,
                       self.clip).item()


class WassersteinGradientPenalty(DiscriminatorLoss):
    def __init__(self, reduction="mean", lambd=10.0, override_train_ops=None):
        super().__init__(reduction, override_train_ops)
        self.lambd = lambd

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        batch_size = real_inputs.size(0)
        noise = torch.randn(batch_size, generator.encoding_dims, device=device)
        eps = torch.rand(1).item()
        return _gp_step(generator, discriminator, optimizer_discriminator, real_inputs.to(device), noise, eps,
                        self.lambd).item()


# ------------------------------------------------------------------------------------------------
# betaVAE-conditioned losses (--loss_type wganvae), src/wgan_loss.py
# ------------------------------------------------------------------------------------------------
class _VAEMixin:
    def _init_vae(self, checkpoint, rna_features, beta):
        self.betavae = betaVAE(rna_features, 2048, [6000, 4000, 2048], [4000, 6000], beta=beta)
        if checkpoint is not None:
            self.betavae.load_state_dict(torch.load(checkpoint, map_location="cpu"))
        self.betavae.eval()

    def _conditioned_noise(self, generator, real_inputs, device):
        """src/wgan_loss.py:94-106: z_mean = betavae.encode(rna)[0]; u ~ U(-0.3, 0.3) drawn on the CPU
        generator; noise = standardise_columns(u + z_mean)."""
        batch_size = real_inputs["image"].size(0)
        gene_coding = real_inputs["rna_data"]
        if next(self.betavae.parameters()).device != torch.device(device):
            self.betavae = self.betavae.to(device)
        z, _, _ = self.betavae.encode(gene_coding.to(device))
        u = torch.FloatTensor(batch_size, generator.encoding_dims).uniform_(-0.3, 0.3).to(device)
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

    def train_ops(self, generator, discriminator, optimizer_generator, device, batch_size, real_inputs,
                  labels=None):
        _check_labels(generator, discriminator, labels)
        noise = self._conditioned_noise(generator, real_inputs, device)
        return _g_step(generator, discriminator, optimizer_generator, noise).item()


class WassersteinDiscriminatorLossVAE(DiscriminatorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, beta=0.005, reduction="mean", clip=None, override_train_ops=None):
        super().__init__(checkpoint, rna_features)
        self.clip = clip if isinstance(clip, (tuple, list)) and len(clip) > 1 else None
        self._init_vae(checkpoint, rna_features, beta)

    def forward(self, fx, fgz):
        return wasserstein_discriminator_loss_vae(fx, fgz, self.reduction)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        noise = self._conditioned_noise(generator, real_inputs, device)
        real = real_inputs["image"].to(device)
        return _d_step(generator, discriminator, optimizer_discriminator, real, noise, self.clip).item()


class WassersteinGradientPenaltyVAE(DiscriminatorLoss, _VAEMixin):
    def __init__(self, checkpoint, rna_features, reduction="mean", lambd=10.0, override_train_ops=None, beta=0.005):
        super().__init__(checkpoint, rna_features)
        self.lambd = lambd
        self.override_train_ops = override_train_ops
        self._init_vae(checkpoint, rna_features, beta)

    def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
        _check_labels(generator, discriminator, labels)
        noise = self._conditioned_noise(generator, real_inputs, device)
        real = real_inputs["image"].to(device)
        eps = torch.rand(1).item()
        return _gp_step(generator, discriminator, optimizer_discriminator, real, noise, eps, self.lambd).item()
