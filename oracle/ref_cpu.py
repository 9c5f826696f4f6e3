"""CPU oracle for the RNA-GAN WGAN-GP training path (TEST INFRASTRUCTURE ONLY).

This file is a plain-PyTorch (fp32, autograd) restatement of the reference's
hot path.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``rna_gan_amd/`` imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this code.

Parity status (see DESIGN.md "Oracle"):
  * PINNED by golden fixtures generated from the importable reference files
    (tests/golden/make_fixtures.py imports /root/reference/src/{betaVAE,
    wgan_loss,dcgan}.py): betaVAE.encode, conditioned-noise construction, the
    three functional losses incl. the gradient penalty, the three
    ``*LossVAE.train_ops`` control flows.
  * UNPINNED (third-party ``torchgan==0.1.0``, requirements.txt:155, not present
    in the container and not vendored by the reference): the exact layer recipe
    of ``DCGANDiscriminator`` and the stock ``Wasserstein*`` losses/Trainer.  The
    generator recipe is pinned indirectly: src/dcgan.py:27-44,52,57-75,82 is a
    lightly edited copy of torchgan's DCGANGenerator and still carries the
    original layers as comments.

Each function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------
# Models
# --------------------------------------------------------------------------------------
def _num_repeats(size: int) -> int:
    # src/dcgan.py:24-27: size must be a power of two >= 16; repeats = bit_length - 4
    if size < 16 or (size & (size - 1)) != 0:
        raise Exception("Target Image Size must be at least 16*16 and an exact power of 2")
    return size.bit_length() - 4


class OracleDCGANGenerator(nn.Module):
    """torchgan DCGANGenerator as used by src/histopathology_gan.py:176-185.

    Recipe mirrored by src/dcgan.py:27-44 (first block), :52 (commented stride-2
    ConvTranspose2d block), :57-75 (batchnorm=False branch), :82 (commented last
    block incl. last_nl), :85-99 (forward: view(-1, E, 1, 1)).
    """

    def __init__(self, encoding_dims=100, out_size=32, out_channels=3, step_channels=64,
                 batchnorm=True, nonlinearity=None, last_nonlinearity=None, label_type="none"):
        super().__init__()
        self.encoding_dims = encoding_dims
        self.label_type = label_type
        reps = _num_repeats(out_size)
        self.ch = out_channels
        self.n = step_channels
        use_bias = not batchnorm
        nl = nn.LeakyReLU(0.2) if nonlinearity is None else nonlinearity
        last_nl = nn.Tanh() if last_nonlinearity is None else last_nonlinearity
        d = int(self.n * (2 ** reps))
        blocks: List[nn.Module] = []
        first = [nn.ConvTranspose2d(encoding_dims, d, 4, 1, 0, bias=use_bias)]
        if batchnorm:
            first.append(nn.BatchNorm2d(d))
        first.append(nl)
        blocks.append(nn.Sequential(*first))
        for _ in range(reps):
            blk = [nn.ConvTranspose2d(d, d // 2, 4, 2, 1, bias=use_bias)]
            if batchnorm:
                blk.append(nn.BatchNorm2d(d // 2))
            blk.append(nl)
            blocks.append(nn.Sequential(*blk))
            d //= 2
        blocks.append(nn.Sequential(nn.ConvTranspose2d(d, self.ch, 4, 2, 1, bias=True), last_nl))
        self.model = nn.Sequential(*blocks)

    def forward(self, x, feature_matching=False):
        x = x.view(-1, x.size(1), 1, 1)
        return self.model(x)

    def sampler(self, sample_size, device):
        # torchgan base Generator.sampler, used at src/gan_utils.py:226
        return [torch.randn(sample_size, self.encoding_dims, device=device)]


class OracleDCGANUpGenerator(nn.Module):
    """DCGANUpGenerator (resize-convolution generator), restated from src/dcgan.py:8-99.

    First block as the DCGAN generator (src/dcgan.py:36-44: ConvTranspose2d(E, d, 4, 1, 0, bias=not batchnorm) +
    BN + nl); ``num_repeats`` blocks [Upsample(x2, bilinear) + ReflectionPad2d(1) + Conv2d(d, d/2, 3, 1, 0) + BN +
    nl] (src/dcgan.py:45-56; the Conv2d keeps its default bias=True); last block [Upsample + ReflectionPad2d(1) +
    Conv2d(d, out_channels, 3)] WITHOUT an activation (src/dcgan.py:76-84: ``last_nl`` is built at :32 but never
    used).  Only the batchnorm=True recipe uses the resize convolution (:57-75 falls back to ConvTranspose2d).
    state_dict keys model.0.{0,1}.*, model.{1..R}.{2,3}.*, model.{R+1}.2.* -- pinned by tests/golden F4/F6."""

    def __init__(self, encoding_dims=100, out_size=32, out_channels=3, step_channels=64,
                 batchnorm=True, nonlinearity=None, last_nonlinearity=None, label_type="none"):
        super().__init__()
        if not batchnorm:
            raise NotImplementedError("oracle covers the batchnorm=True recipe (the resize-convolution one)")
        self.encoding_dims = encoding_dims
        self.label_type = label_type
        reps = _num_repeats(out_size)
        self.ch = out_channels
        self.n = step_channels
        nl = nn.LeakyReLU(0.2) if nonlinearity is None else nonlinearity
        d = int(self.n * (2 ** reps))
        blocks: List[nn.Module] = [nn.Sequential(nn.ConvTranspose2d(encoding_dims, d, 4, 1, 0, bias=False),
                                                 nn.BatchNorm2d(d), nl)]
        for _ in range(reps):
            blocks.append(nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"), nn.ReflectionPad2d(1),
                                        nn.Conv2d(d, d // 2, kernel_size=3, stride=1, padding=0),
                                        nn.BatchNorm2d(d // 2), nl))
            d //= 2
        blocks.append(nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"), nn.ReflectionPad2d(1),
                                    nn.Conv2d(d, self.ch, kernel_size=3, stride=1, padding=0)))
        self.model = nn.Sequential(*blocks)

    def forward(self, x, feature_matching=False):
        x = x.view(-1, x.size(1), 1, 1)
        return self.model(x)

    def sampler(self, sample_size, device):
        return [torch.randn(sample_size, self.encoding_dims, device=device)]


class OracleDCGANDiscriminator(nn.Module):
    """torchgan DCGANDiscriminator, ctor kwargs per src/histopathology_gan.py:186-192.

    No source under /root/reference (parity unpinned): mirror image of the
    generator recipe: first block Conv(k4,s2,p1,bias=True)+nl,
    ``num_repeats`` x [Conv(d->2d,k4,s2,p1,bias=not batchnorm)+BN+nl], head
    ``disc`` = Conv(d->1,k4,s1,p0,bias=not batchnorm)+last_nl, output view(N).
    """

    def __init__(self, in_size=32, in_channels=3, step_channels=64, batchnorm=True,
                 nonlinearity=None, last_nonlinearity=None, label_type="none"):
        super().__init__()
        self.input_dims = in_channels
        self.label_type = label_type
        reps = _num_repeats(in_size)
        self.n = step_channels
        use_bias = not batchnorm
        nl = nn.LeakyReLU(0.2) if nonlinearity is None else nonlinearity
        last_nl = nn.LeakyReLU(0.2) if last_nonlinearity is None else last_nonlinearity
        d = self.n
        blocks: List[nn.Module] = [nn.Sequential(nn.Conv2d(in_channels, d, 4, 2, 1, bias=True), nl)]
        for _ in range(reps):
            blk = [nn.Conv2d(d, d * 2, 4, 2, 1, bias=use_bias)]
            if batchnorm:
                blk.append(nn.BatchNorm2d(d * 2))
            blk.append(nl)
            blocks.append(nn.Sequential(*blk))
            d *= 2
        self.model = nn.Sequential(*blocks)
        self.disc = nn.Sequential(nn.Conv2d(d, 1, 4, 1, 0, bias=use_bias), last_nl)

    def forward(self, x, feature_matching=False):
        x = self.model(x)
        if feature_matching:
            return x
        x = self.disc(x)
        return x.view(x.size(0))


class OracleBetaVAE(nn.Module):
    """betaVAE module layout of src/betaVAE.py:18-42,63-107 (same state_dict keys).

    ``encode`` is on the GAN hot path (src/wgan_loss.py:96-97,223-224,353-354); ``forward_with`` restates the full
    forward for the betaVAE training row, pinned by tests/golden/f7_vae_train.npz.
    """

    def __init__(self, in_channels, z_dim, encoder_dims, hidden_dims_decoder, beta=2):
        super().__init__()

        class _Enc(nn.Module):
            def __init__(self, cin, dims):
                super().__init__()
                mods: List[nn.Module] = [nn.Sequential(nn.Dropout())]
                for h in dims:
                    mods.append(nn.Sequential(nn.Linear(cin, h), nn.BatchNorm1d(h), nn.LeakyReLU()))
                    cin = h
                self.encoder = nn.Sequential(*mods)

            def forward(self, x):
                return self.encoder(x)

        self.encoder = _Enc(in_channels, encoder_dims)
        self.z_mu = nn.Linear(z_dim, z_dim)
        self.z_logvar = nn.Linear(z_dim, z_dim)
        self.beta = beta
        mods: List[nn.Module] = []
        cin = z_dim
        for h in hidden_dims_decoder:
            mods.append(nn.Sequential(nn.Linear(cin, h), nn.BatchNorm1d(h), nn.LeakyReLU()))
            cin = h
        mods.append(nn.Sequential(nn.Linear(cin, in_channels), nn.Tanh()))
        self.decoder = nn.Sequential(*mods)
        self.z_dim = z_dim

    def encode(self, x):
        # src/betaVAE.py:102-107
        h = self.encoder(x)
        return self.z_mu(h), self.z_logvar(h), h

    def forward_with(self, x, mask=None, eps=None, bf16_gemm=False):
        """src/betaVAE.py:108-114 with the two random draws made explicit: ``mask`` = the keep-mask nn.Dropout()
        (p = 0.5, :27) drew in train mode (None in eval mode: identity), ``eps`` = reparametrize's randn_like (:96-100).
        bf16_gemm: the twin of the bf16 product path -- every GEMM (forward, data and weight gradient) takes its two
        operands rounded to bf16 and accumulates in fp32; everything between the GEMMs stays fp32."""
        lin = (lambda h, l: _Bf16GemmLinear.apply(h, l.weight, l.bias)) if bf16_gemm else (lambda h, l: l(h))
        blocks = list(self.encoder.encoder.children())
        h = x
        if self.training:
            p = blocks[0][0].p
            h = x * mask.to(x.dtype) / (1.0 - p)
        for blk in blocks[1:]:
            h = blk[2](blk[1](lin(h, blk[0])))
        z_mean, z_log_var = lin(h, self.z_mu), lin(h, self.z_logvar)
        h = z_mean + eps * torch.exp(0.5 * z_log_var)
        dec = list(self.decoder.children())
        for blk in dec[:-1]:
            h = blk[2](blk[1](lin(h, blk[0])))
        return dec[-1][1](lin(h, dec[-1][0])), z_mean, z_log_var


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _Bf16GemmLinear(torch.autograd.Function):
    """y = bf16(x) . bf16(W)^T + b ;  dx = bf16(gy) . bf16(W) ;  dW = bf16(gy)^T . bf16(x) ;  db = sum gy  (fp32 sums)"""

    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = _r16(x), _r16(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t() + b

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = _r16(gy)
        return gr @ wr, gr.t() @ xr, gy.sum(0)


def oracle_vae_loss(x, x_recons, z_mean, z_logvar, beta, training=True):
    """betaVAEloss, src/betaVAE.py:145-163 (kld_weight is unused there)."""
    recons = torch.mean((x_recons - x) ** 2)
    kld = torch.mean(-0.5 * torch.sum(1 + z_logvar - z_mean ** 2 - z_logvar.exp(), dim=1), dim=0)
    total = recons + beta * kld if training else recons
    return {"total_loss": total, "reconstruction_loss": recons, "kl_loss": kld}


def oracle_vae_train_step(model, optimizer, x, mask, eps, bf16_gemm=False):
    """One 'train'-phase iteration of train_betaVAE (src/betaVAE.py:218-234): zero_grad, forward, loss, backward, step."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    out, z_mean, z_log_var = model.forward_with(x, mask, eps, bf16_gemm)
    losses = oracle_vae_loss(x, out, z_mean, z_log_var, model.beta, training=True)
    losses["total_loss"].backward()
    optimizer.step()
    return out.detach(), z_mean.detach(), z_log_var.detach(), {k: v.detach() for k, v in losses.items()}


# --------------------------------------------------------------------------------------
# Seeded, portable parameter / input generators (numpy PCG64: identical on every box)
# --------------------------------------------------------------------------------------
def _name_seed(seed: int, name: str) -> List[int]:
    import zlib
    return [int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode("utf-8"))]


def seeded_tensor(name: str, shape: Sequence[int], seed: int) -> torch.Tensor:
    """One tensor of the seeded, build-owned weight generator.

    Seeded per tensor NAME (numpy PCG64 seeded with [seed, crc32(name)]), so the value of a
    tensor does not depend on module traversal order and can be regenerated anywhere:
      * ``...running_var``           1 + |N(0, 0.1^2)|
      * ``...running_mean``          N(0, 0.1^2)
      * 1-D ``...weight`` (BN gamma) 1 + N(0, 0.1^2)
      * other 1-D (biases, BN beta)  N(0, 0.05^2)
      * >=2-D weights                N(0, 2/fan_in), fan_in = prod(shape[1:])  (kaiming scale;
        for ConvTranspose2d (I,O,kh,kw) this is O*kh*kw, as torch's kaiming_normal_ computes)
    """
    rng = np.random.default_rng(_name_seed(seed, name))
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if name.endswith("running_var"):
        v = 1.0 + np.abs(rng.normal(0.0, 0.1, size=shape))
    elif name.endswith("running_mean"):
        v = rng.normal(0.0, 0.1, size=shape)
    elif len(shape) == 1 and name.endswith("weight"):
        v = 1.0 + rng.normal(0.0, 0.1, size=shape)
    elif len(shape) <= 1:
        v = rng.normal(0.0, 0.05, size=shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        v = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_in))
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def seeded_fill_(module: nn.Module, seed: int) -> nn.Module:
    """Fill every parameter and buffer of ``module`` in place with ``seeded_tensor``."""
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            t.copy_(seeded_tensor(name, t.shape, seed))
    return module


def synthetic_images(n: int, size: int, seed: int, channels: int = 3) -> torch.Tensor:
    """uint8 uniform tiles -> /255 -> (x-0.5)/0.5, as src/histopathology_gan.py:106-109."""
    rng = np.random.default_rng(seed)
    u8 = rng.integers(0, 256, size=(n, channels, size, size), dtype=np.uint8)
    return (torch.from_numpy(u8).float() / 255.0 - 0.5) / 0.5


def synthetic_rna(n: int, features: int, seed: int, distinct: int = 16) -> torch.Tensor:
    """N(0,1) rows (StandardScaler output, src/histopathology_gan.py:148-151); tiles of a
    slide share one RNA row, so only ``distinct`` different rows appear in a batch."""
    rng = np.random.default_rng(seed)
    rows = rng.normal(0.0, 1.0, size=(min(distinct, n), features)).astype(np.float32)
    idx = np.arange(n) % rows.shape[0]
    return torch.from_numpy(rows[idx])


def synthetic_uniform(n: int, dims: int, seed: int, lo=-0.3, hi=0.3) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(lo, hi, size=(n, dims)).astype(np.float32))


def synthetic_normal(n: int, dims: int, seed: int) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.normal(0.0, 1.0, size=(n, dims)).astype(np.float32))


# --------------------------------------------------------------------------------------
# Functional pieces of src/wgan_loss.py
# --------------------------------------------------------------------------------------
def conditioned_noise(u: torch.Tensor, z_mean: torch.Tensor) -> torch.Tensor:
    """src/wgan_loss.py:100-106 (= :227-233, :357-363; src/gan_utils.py:211-216):
    noise = u + z; per-column standardisation with the *unbiased* std over the batch."""
    n = u + z_mean
    return (n - torch.mean(n, dim=0)) / torch.std(n, dim=0)


def generator_loss(dgz: torch.Tensor) -> torch.Tensor:
    """src/wgan_loss.py:24-25 (reduction argument is ignored there: always mean)."""
    return torch.mean(-1.0 * dgz)


def discriminator_loss(dx: torch.Tensor, dgz: torch.Tensor) -> torch.Tensor:
    """src/wgan_loss.py:28-29."""
    return torch.mean(dgz - dx)


def gradient_penalty(interpolate: torch.Tensor, d_interpolate: torch.Tensor) -> torch.Tensor:
    """src/wgan_loss.py:32-44: ONE 2-norm over the whole batch gradient tensor."""
    g = torch.autograd.grad(outputs=d_interpolate, inputs=interpolate,
                            grad_outputs=torch.ones_like(d_interpolate),
                            create_graph=True, retain_graph=True, only_inputs=True)[0]
    return (g.norm(2) - 1) ** 2


def encode_latent(betavae: OracleBetaVAE, rna: torch.Tensor) -> torch.Tensor:
    """z_mean of src/betaVAE.py:102-107 in eval mode (src/wgan_loss.py:69)."""
    betavae.eval()
    return betavae.encode(rna)[0]


# --------------------------------------------------------------------------------------
# The three train_ops with injected randomness (src/wgan_loss.py:82-129,181-263,314-389)
# --------------------------------------------------------------------------------------
def gen_step(G, D, opt_g, noise: torch.Tensor) -> float:
    """WassersteinGeneratorLoss(VAE).train_ops, src/wgan_loss.py:107-129.
    ``noise`` is the final generator input (already conditioned/standardised)."""
    opt_g.zero_grad()
    fake = G(noise)
    dgz = D(fake)
    loss = generator_loss(dgz)
    loss.backward()
    opt_g.step()
    return loss.item()


def disc_step(G, D, opt_d, real: torch.Tensor, noise: torch.Tensor,
              clip: Optional[Tuple[float, float]] = None) -> float:
    """WassersteinDiscriminatorLoss(VAE).train_ops, src/wgan_loss.py:213-263.
    Order matters for BN running stats: clamp, D(real), G(noise), D(fake.detach())."""
    if clip is not None:
        for p in D.parameters():
            p.data.clamp_(clip[0], clip[1])
    opt_d.zero_grad()
    dx = D(real)
    fake = G(noise)
    dgz = D(fake.detach())
    loss = discriminator_loss(dx, dgz)
    loss.backward()
    opt_d.step()
    return loss.item()


def gp_step(G, D, opt_d, real: torch.Tensor, noise: torch.Tensor, eps: float,
            lambd: float = 10.0) -> float:
    """WassersteinGradientPenalty(VAE).train_ops, src/wgan_loss.py:369-389.
    ``fake`` is NOT detached in the reference (generator grads are produced and never
    used); the returned value is the UNWEIGHTED penalty."""
    opt_d.zero_grad()
    fake = G(noise)
    interpolate = eps * real + (1 - eps) * fake
    d_int = D(interpolate)
    loss = gradient_penalty(interpolate, d_int)
    (lambd * loss).backward()
    opt_d.step()
    return loss.item()


def make_adam(params, lr):
    """src/histopathology_gan.py:252,257: Adam(lr, betas=(0.5, 0.999)), torch defaults otherwise."""
    return torch.optim.Adam(params, lr=lr, betas=(0.5, 0.999))


def train_iteration(G, D, opt_g, opt_d, real, noises: Sequence[torch.Tensor], eps: float,
                    clip=None, lambd: float = 10.0) -> Dict[str, float]:
    """One hot-loop iteration = the three train_ops in torchgan Trainer order
    (losses list order, src/histopathology_gan.py:267-278; ncritic=1)."""
    out = {}
    out["g"] = gen_step(G, D, opt_g, noises[0])
    out["d"] = disc_step(G, D, opt_d, real, noises[1], clip=clip)
    out["gp"] = gp_step(G, D, opt_d, real, noises[2], eps, lambd=lambd)
    return out


def generate_images(generator, gene_exp=None, sample_size=64, betavae=None):
    """Restatement of generate_images (src/gan_utils.py:197-244) on explicit modules: conditioned noise (:211-216),
    generator on chunks of 10 rows in its current mode (:217-221 / :226-231), un-normalise + NHWC (:236-243)."""
    if gene_exp is not None:
        noise = torch.FloatTensor(sample_size, generator.encoding_dims).uniform_(-0.3, 0.3)
        z, _, _ = betavae.encode(gene_exp)
        noise = noise + z.detach()
        noise = (noise - torch.mean(noise, dim=0)) / torch.std(noise, dim=0)
    else:
        noise = generator.sampler(sample_size, torch.device("cpu"))[0]
    with torch.no_grad():
        images = torch.cat([generator(chunk) for chunk in torch.split(noise, 10)], dim=0)
    images = images.view((-1, 3, images.shape[-2], images.shape[-1]))
    images = (images - (-1.0)) / 2.0                      # Normalize((-mean/std), (1/std)), mean = std = 0.5
    return images.permute(0, 2, 3, 1).contiguous().numpy()


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """Restatement of calculate_frechet_distance (src/fid.py:112-163; statistics as calculate_activation_statistics
    :106-109: mean and np.cov(rowvar=False)):  |mu1-mu2|^2 + Tr(C1) + Tr(C2) - 2 Tr((C1 C2)^(1/2))."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))
