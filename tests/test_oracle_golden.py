"""Pin the CPU oracle (oracle/ref_cpu.py) to the golden vectors produced by the reference's
own code (tests/golden/make_fixtures.py imported /root/reference/src/{betaVAE,wgan_loss,dcgan}.py).
CPU only; nothing here reads /root/reference."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import ref_cpu as R

DIGEST_OVER = 20000


def digest(t):
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if a.size <= DIGEST_OVER:
        return a
    f = a.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), (f * f).sum()], f[:64], f[-64:]])


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                               rtol=rtol, atol=atol)


def test_f1_betavae_small(golden_dir):
    fx = np.load(os.path.join(golden_dir, "f1_betavae_small.npz"))
    m = R.seeded_fill_(R.OracleBetaVAE(64, 16, [48, 32, 16], [32, 48]), 11)
    rna = R.synthetic_rna(6, 64, seed=12, distinct=4)
    with torch.no_grad():
        m.eval()
        zm, zl, h = m.encode(rna)
        close(R.encode_latent(m, rna), fx["z_mean"])
    close(zm, fx["z_mean"]); close(zl, fx["z_logvar"]); close(h, fx["x_encoded"])


def test_f1_betavae_full(golden_dir):
    fx = np.load(os.path.join(golden_dir, "f1_betavae_full.npz"))
    m = R.OracleBetaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000])
    R.seeded_fill_(m, 13)
    rna = R.synthetic_rna(4, 19198, seed=14, distinct=4)
    with torch.no_grad():
        z = R.encode_latent(m, rna)
    close(z[:, :64], fx["z_mean_first"], rtol=1e-4, atol=1e-5)
    close(z[:, -64:], fx["z_mean_last"], rtol=1e-4, atol=1e-5)
    close(z.double().sum(1), fx["z_mean_sum"], rtol=1e-4, atol=1e-3)
    close((z.double() ** 2).sum(1), fx["z_mean_sumsq"], rtol=1e-4)


def _tiny_D(seed=21):
    D = R.OracleDCGANDiscriminator(16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                   last_nonlinearity=nn.LeakyReLU(0.2))
    return R.seeded_fill_(D, seed)


def _tiny_G(seed=31, enc=2048):
    G = R.OracleDCGANGenerator(enc, 16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                               last_nonlinearity=nn.Tanh())
    return R.seeded_fill_(G, seed)


def test_f3_losses(golden_dir):
    fx = np.load(os.path.join(golden_dir, "f3_losses_tinyD.npz"))
    D = _tiny_D(); D.train()
    real = R.synthetic_images(4, 16, seed=22)
    fake = torch.tanh(R.synthetic_normal(4, 3 * 16 * 16, seed=23).view(4, 3, 16, 16))
    eps = 0.37
    interp = (eps * real + (1 - eps) * fake).requires_grad_(True)
    d_int = D(interp)
    gp = R.gradient_penalty(interp, d_int)
    (10.0 * gp).backward()
    close(gp.detach(), fx["gp"]); close(d_int.detach(), fx["d_int"])
    for k, p in D.named_parameters():
        close(p.grad, fx["grad." + k], rtol=1e-4, atol=1e-6)
    fxv = R.synthetic_normal(1, 8, seed=24).view(8)
    fgz = R.synthetic_normal(1, 8, seed=25).view(8)
    close(R.generator_loss(fgz), fx["gen_loss"])
    close(R.discriminator_loss(fxv, fgz), fx["disc_loss"])
    close(D.state_dict()["model.1.1.running_mean"], fx["running_mean"])
    close(D.state_dict()["model.1.1.running_var"], fx["running_var"])


@pytest.mark.parametrize("size", [16, 32])
def test_f4_up_generator(golden_dir, size):
    """The reference's DCGANUpGenerator (src/dcgan.py, imported when the fixture was made): output, every parameter
    gradient and the BN buffers of one forward/backward, against the oracle's restatement."""
    fx = np.load(os.path.join(golden_dir, "f4_upgen_tiny.npz"))
    G = R.OracleDCGANUpGenerator(16, size, 3, 4, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    R.seeded_fill_(G, 41)
    G.train()
    z = R.synthetic_normal(3, 16, seed=42)
    y = G(z)
    cot = R.synthetic_normal(3, 3 * size * size, seed=43).view(3, 3, size, size)
    (y * cot).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx[f"y{size}"], rtol=1e-5, atol=1e-6)
    for k, p in G.named_parameters():
        ref = fx[f"grad{size}.{k}"]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    for k, b in G.named_buffers():
        np.testing.assert_allclose(b.numpy(), fx[f"buf{size}.{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


def test_f5_trainops(golden_dir):
    """The oracle's three steps (injected noise/eps) reproduce the reference's *LossVAE.train_ops."""
    fx = np.load(os.path.join(golden_dir, "f5_trainops_vae.npz"))
    RNA_F, bs = 64, 6
    bv = R.seeded_fill_(R.OracleBetaVAE(RNA_F, 2048, [6000, 4000, 2048], [4000, 6000]), 51)
    G, D = _tiny_G(), _tiny_D()
    G.train(); D.train()
    opt_g = R.make_adam(G.parameters(), 1e-4)
    opt_d = R.make_adam(D.parameters(), 4e-4)
    for it in range(2):
        real = R.synthetic_images(bs, 16, seed=60 + it)
        rna = R.synthetic_rna(bs, RNA_F, seed=70 + it, distinct=3)
        with torch.no_grad():
            z = R.encode_latent(bv, rna)
        noises = [R.conditioned_noise(torch.from_numpy(fx[f"u.{it}.{t}"]), z) for t in ("g", "d", "gp")]
        out = R.train_iteration(G, D, opt_g, opt_d, real, noises, float(fx[f"eps.{it}"]))
        for t in ("g", "d", "gp"):
            close(out[t], fx[f"loss.{it}.{t}"], rtol=2e-4, atol=1e-6)
    for k, v in G.state_dict().items():
        close(digest(v), fx["G." + k], rtol=2e-4, atol=2e-6)
    for k, v in D.state_dict().items():
        close(digest(v), fx["D." + k], rtol=2e-4, atol=2e-6)
    for nm, opt, mod in (("optG", opt_g, G), ("optD", opt_d, D)):
        for i, (k, _) in enumerate(mod.named_parameters()):
            st = opt.state_dict()["state"][i]
            close(digest(st["exp_avg"]), fx[f"{nm}.{k}.exp_avg"], rtol=1e-3, atol=1e-7)
            close(digest(st["exp_avg_sq"]), fx[f"{nm}.{k}.exp_avg_sq"], rtol=1e-3, atol=1e-10)
            assert float(st["step"]) == float(fx[f"{nm}.{k}.step"])
    assert float(fx["G.grad_nonzero_after_gp"]) == 1.0


def test_f6_manifests(golden_dir):
    with open(os.path.join(golden_dir, "f6_manifests.json")) as f:
        mf = json.load(f)
    ref_bv = mf["manifests"]["betaVAE(19198,2048,[6000,4000,2048],[4000,6000])"]
    with torch.device("meta"):
        bv = R.OracleBetaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000])
    mine = {k: list(v.shape) for k, v in bv.state_dict().items()}
    assert mine == ref_bv
    assert sum(int(np.prod(s)) for k, s in ref_bv.items() if "num_batches" not in k and "running" not in k) == 303238046
    q = mf["loss_ctor_quirks"]
    assert q["reduction_is_path"] and q["override_train_ops"] == 64 and q["lambd"] == 10.0 and q["clip_none"]
