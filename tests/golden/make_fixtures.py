#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_fixtures.py

What is executed from the reference (paths relative to /root/reference/src):
  * betaVAE.py   -> betaVAE(...).encode           (imports as-is)
  * wgan_loss.py -> wasserstein_*_loss_vae, wasserstein_gradient_penalty_vae and the three
                    *LossVAE.train_ops           (needs the 4 torchgan base classes -> shim)
  * dcgan.py     -> DCGANUpGenerator              (same shim)
The torchgan shim (tests/golden/_torchgan_shim) holds constructors only; no arithmetic.

Inputs and weights come from oracle.ref_cpu's seeded generators (numpy PCG64 keyed by tensor
name), so tests regenerate them instead of storing them.  Only expected OUTPUTS are stored
(plus the RNG draws the reference made internally: the uniform noise and eps).  Large tensors
are stored as digests (sum, sum of squares, first/last 64 values).
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "_torchgan_shim"))
sys.path.insert(0, REF)

from oracle import ref_cpu as R  # noqa: E402

import betaVAE as ref_betavae  # noqa: E402  (reference)
import dcgan as ref_dcgan  # noqa: E402      (reference)
import wgan_loss as ref_wgan  # noqa: E402   (reference)

DIGEST_OVER = 20000


def pack(t):
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if a.size <= DIGEST_OVER:
        return a
    f = a.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), (f * f).sum()], f[:64], f[-64:]]).astype(np.float64)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: pack(v) for k, v in arrays.items()})
    print("wrote", path, {k: np.asarray(pack(v)).shape for k, v in arrays.items()})


def fill_by_name(module, seed):
    return R.seeded_fill_(module, seed)


# ---------------------------------------------------------------------------------------
def f1_betavae():
    torch.manual_seed(0)
    m = ref_betavae.betaVAE(64, 16, [48, 32, 16], [32, 48], beta=0.005)
    fill_by_name(m, 11)
    m.eval()
    rna = R.synthetic_rna(6, 64, seed=12, distinct=4)
    with torch.no_grad():
        z_mean, z_logvar, h = m.encode(rna)
    save("f1_betavae_small.npz", z_mean=z_mean, z_logvar=z_logvar, x_encoded=h)

    # full size (the ctor args of src/wgan_loss.py:67)
    m = ref_betavae.betaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000], beta=0.005)
    fill_by_name(m, 13)
    m.eval()
    rna = R.synthetic_rna(4, 19198, seed=14, distinct=4)
    with torch.no_grad():
        z_mean, _, h = m.encode(rna)
    np.savez_compressed(os.path.join(HERE, "f1_betavae_full.npz"),
                        z_mean_first=z_mean[:, :64].numpy(), z_mean_last=z_mean[:, -64:].numpy(),
                        z_mean_sum=z_mean.double().sum(1).numpy(),
                        z_mean_sumsq=(z_mean.double() ** 2).sum(1).numpy())
    print("wrote f1_betavae_full.npz")
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def tiny_D(seed=21):
    D = R.OracleDCGANDiscriminator(16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                   last_nonlinearity=nn.LeakyReLU(0.2))
    return fill_by_name(D, seed)


def tiny_G(seed=31, enc=2048):
    G = R.OracleDCGANGenerator(enc, 16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                               last_nonlinearity=nn.Tanh())
    return fill_by_name(G, seed)


def f3_losses():
    D = tiny_D()
    D.train()
    real = R.synthetic_images(4, 16, seed=22)
    fake = torch.tanh(R.synthetic_normal(4, 3 * 16 * 16, seed=23).view(4, 3, 16, 16))
    eps = 0.37
    interp = (eps * real + (1 - eps) * fake).requires_grad_(True)
    d_int = D(interp)
    gp = ref_wgan.wasserstein_gradient_penalty_vae(interp, d_int)
    (10.0 * gp).backward()
    grads = {("grad." + k): p.grad for k, p in D.named_parameters()}
    fx = R.synthetic_normal(1, 8, seed=24).view(8)
    fgz = R.synthetic_normal(1, 8, seed=25).view(8)
    save("f3_losses_tinyD.npz", gp=gp.detach(), d_int=d_int.detach(),
         gen_loss=ref_wgan.wasserstein_generator_loss_vae(fgz),
         disc_loss=ref_wgan.wasserstein_discriminator_loss_vae(fx, fgz),
         running_mean=D.state_dict()["model.1.1.running_mean"],
         running_var=D.state_dict()["model.1.1.running_var"], **grads)


def f4_upgen():
    out = {}
    manifest = {}
    for size in (16, 32):
        G = ref_dcgan.DCGANUpGenerator(16, size, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                       last_nonlinearity=nn.Tanh())
        fill_by_name(G, 41)
        G.train()
        z = R.synthetic_normal(3, 16, seed=42)
        y = G(z)
        cot = R.synthetic_normal(3, 3 * size * size, seed=43).view(3, 3, size, size)
        (y * cot).sum().backward()
        out[f"y{size}"] = y.detach()
        for k, p in G.named_parameters():
            out[f"grad{size}.{k}"] = p.grad
        for k, b in G.named_buffers():
            out[f"buf{size}.{k}"] = b
    save("f4_upgen_tiny.npz", **out)
    G = ref_dcgan.DCGANUpGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2),
                                   last_nonlinearity=nn.Tanh())
    manifest = {k: list(v.shape) for k, v in G.state_dict().items()}
    return manifest


def f5_trainops():
    """The reference's own three *LossVAE.train_ops, 2 iterations, tiny G/D, rna_features=64."""
    RNA_F = 64
    bv = ref_betavae.betaVAE(RNA_F, 2048, [6000, 4000, 2048], [4000, 6000], beta=0.005)
    fill_by_name(bv, 51)
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "bv.pt")
        torch.save(bv.state_dict(), ck)
        del bv
        lg = ref_wgan.WassersteinGeneratorLossVAE(checkpoint=ck, rna_features=RNA_F)
        ld = ref_wgan.WassersteinDiscriminatorLossVAE(checkpoint=ck, rna_features=RNA_F)
        lp = ref_wgan.WassersteinGradientPenaltyVAE(checkpoint=ck, rna_features=RNA_F)
    # ctor quirk noted in SURVEY 8 a9: the path lands in .reduction, rna_features in .override_train_ops
    quirks = dict(reduction_is_path=isinstance(lg.reduction, str) and lg.reduction.endswith("bv.pt"),
                  override_train_ops=int(lg.override_train_ops), lambd=float(lp.lambd),
                  clip_none=ld.clip is None)
    G, D = tiny_G(), tiny_D()
    G.train(); D.train()
    opt_g = R.make_adam(G.parameters(), 1e-4)
    opt_d = R.make_adam(D.parameters(), 4e-4)
    dev = torch.device("cpu")
    bs = 6
    out = {}
    call = 0
    for it in range(2):
        batch = {"image": R.synthetic_images(bs, 16, seed=60 + it),
                 "rna_data": R.synthetic_rna(bs, RNA_F, seed=70 + it, distinct=3)}
        for tag, fn in (("g", lambda: lg.train_ops(G, D, opt_g, dev, bs, batch)),
                        ("d", lambda: ld.train_ops(G, D, opt_d, batch, dev)),
                        ("gp", lambda: lp.train_ops(G, D, opt_d, batch, dev))):
            seed = 1000 + call
            torch.manual_seed(seed)
            loss = fn()
            # replay the RNG draws the reference just made (src/wgan_loss.py:100,227,357,376)
            torch.manual_seed(seed)
            u = torch.FloatTensor(bs, 2048).uniform_(-0.3, 0.3)
            out[f"u.{it}.{tag}"] = u.numpy().copy()          # stored in full: it is an INPUT
            if tag == "gp":
                out[f"eps.{it}"] = np.float64(torch.rand(1).item())
            out[f"loss.{it}.{tag}"] = np.float64(loss)
            call += 1
    for k, v in G.state_dict().items():
        out["G." + k] = v
    for k, v in D.state_dict().items():
        out["D." + k] = v
    for nm, opt, mod in (("optG", opt_g, G), ("optD", opt_d, D)):
        names = [k for k, _ in mod.named_parameters()]
        for i, k in enumerate(names):
            st = opt.state_dict()["state"][i]
            out[f"{nm}.{k}.exp_avg"] = st["exp_avg"]
            out[f"{nm}.{k}.exp_avg_sq"] = st["exp_avg_sq"]
            out[f"{nm}.{k}.step"] = np.float64(float(st["step"]))
    # generator grads left behind by the GP step (fake not detached, src/wgan_loss.py:371)
    out["G.grad_nonzero_after_gp"] = np.float64(float(sum(float(p.grad.abs().sum()) for p in G.parameters()) > 0))
    path = os.path.join(HERE, "f5_trainops_vae.npz")
    packed = {}
    for k, v in out.items():
        packed[k] = v if k.startswith("u.") else pack(v)
    np.savez_compressed(path, **packed)
    print("wrote", path, len(packed), "arrays")
    return quirks


def main():
    torch.set_num_threads(8)
    manifests = {}
    manifests["betaVAE(19198,2048,[6000,4000,2048],[4000,6000])"] = f1_betavae()
    f3_losses()
    manifests["DCGANUpGenerator(2048,256,3,64)"] = f4_upgen()
    quirks = f5_trainops()
    G = R.OracleDCGANGenerator(2048, 256, 3, 64)
    D = R.OracleDCGANDiscriminator(256, 3, 64)
    manifests["UNPINNED torchgan DCGANGenerator(2048,256,3,64) [recipe: src/dcgan.py comments]"] = \
        {k: list(v.shape) for k, v in G.state_dict().items()}
    manifests["UNPINNED torchgan DCGANDiscriminator(256,3,64) [recalled recipe]"] = \
        {k: list(v.shape) for k, v in D.state_dict().items()}
    with open(os.path.join(HERE, "f6_manifests.json"), "w") as f:
        json.dump({"manifests": manifests, "loss_ctor_quirks": quirks,
                   "torch": torch.__version__}, f, indent=1)
    print("wrote f6_manifests.json")


if __name__ == "__main__":
    main()
