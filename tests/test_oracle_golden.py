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


VAE_CFG = dict(features=70, z=16, enc=[48, 32, 16], dec=[32, 48], beta=2.0, batch=12, lr=3e-3, weight_decay=1e-4)


def vae_fixture_model():
    c = VAE_CFG
    m = R.seeded_fill_(R.OracleBetaVAE(c["features"], c["z"], c["enc"], c["dec"], beta=c["beta"]), 21)
    x = R.synthetic_rna(c["batch"], c["features"], seed=22, distinct=c["batch"])
    return m, x


# Linear biases in front of a BatchNorm have a mathematically ZERO gradient (the batch mean is subtracted again); what
# autograd returns is rounding noise (~1e-9) and Adam normalises it to a full-size step of +-lr in a direction that
# depends on summation order.  They are compared through Adam's step bound instead of value by value.
VAE_DEAD_BIASES = ("encoder.encoder.1.0.bias", "encoder.encoder.2.0.bias", "encoder.encoder.3.0.bias",
                   "decoder.0.0.bias", "decoder.1.0.bias")


def check_vae_state_after_step(sd, g, rtol, atol):
    for k in g.files:
        if not k.startswith("after."):
            continue
        name = k[6:]
        got = sd[name].detach().cpu().numpy()
        if name in VAE_DEAD_BIASES:
            assert np.abs(got - g[k]).max() <= 2.02 * VAE_CFG["lr"], name
        else:
            np.testing.assert_allclose(got, g[k], rtol=rtol, atol=atol, err_msg=k)


def test_f7_vae_training_step(golden_dir):
    """betaVAE training row: the oracle's forward / betaVAEloss / backward / Adam(weight_decay) step and
    the eval-mode forward against the reference's own run (tests/golden/make_vae_train_fixture.py)."""
    g = np.load(os.path.join(golden_dir, "f7_vae_train.npz"))
    c = VAE_CFG
    m, x = vae_fixture_model()
    opt = torch.optim.Adam(m.parameters(), weight_decay=c["weight_decay"], lr=c["lr"])
    out, mu, lv, losses = R.oracle_vae_train_step(m, opt, x, torch.from_numpy(g["train.mask"]), torch.from_numpy(g["train.eps"]))
    for name, t in (("out", out), ("z_mean", mu), ("z_log_var", lv)):
        np.testing.assert_allclose(t.numpy(), g["train." + name], rtol=1e-5, atol=1e-6, err_msg=name)
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(g["train." + k]), rtol=1e-5, err_msg=k)
    check_vae_state_after_step(m.state_dict(), g, rtol=1e-5, atol=1e-6)
    # eval forward from the reference's own post-step state (in eval mode the dead biases are no longer cancelled)
    m.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("after.")})
    m.eval()
    with torch.no_grad():
        out, mu, lv = m.forward_with(x, None, torch.from_numpy(g["eval.eps"]))
        losses = R.oracle_vae_loss(x, out, mu, lv, m.beta, training=False)
    np.testing.assert_allclose(out.numpy(), g["eval.out"], rtol=1e-5, atol=1e-6)
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(g["eval." + k]), rtol=1e-5, err_msg=k)


def test_f9_generate_images_pins_oracle(golden_dir):
    """f9_generate_images.npz = what the REFERENCE's gan_utils.generate_images (src/gan_utils.py:197-244, imported by
    tests/golden/make_fid_genimg_fixtures.py) returned around the oracle generator and the reference betaVAE class:
    the oracle restatement (generator + OracleBetaVAE) must reproduce it with the same torch seed."""
    fx = np.load(os.path.join(golden_dir, "f9_generate_images.npz"))
    E_, F = 16, 40
    G = R.seeded_fill_(R.OracleDCGANGenerator(E_, 256, 3, 1, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 81)
    G.train()
    bv = R.seeded_fill_(R.OracleBetaVAE(F, E_, [32, 24, E_], [24, 32], beta=0.0005), 82)
    bv.eval()
    rna = R.synthetic_rna(1, F, seed=83, distinct=1)
    torch.manual_seed(5)
    cond = R.generate_images(G, gene_exp=rna, sample_size=13, betavae=bv)
    assert cond.shape == (13, 256, 256, 3)
    np.testing.assert_allclose(cond[:, ::16, ::16, :], fx["cond.sub"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cond.astype(np.float64).sum(), float(fx["cond.sum"]), rtol=1e-7)
    np.testing.assert_allclose((cond.astype(np.float64) ** 2).sum(), float(fx["cond.sumsq"]), rtol=1e-7)
    for k, v in G.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), fx["bn_after_cond." + k], rtol=1e-6, atol=1e-7, err_msg=k)
    torch.manual_seed(6)
    unc = R.generate_images(G, sample_size=20)
    np.testing.assert_allclose(unc[:, ::16, ::16, :], fx["unc.sub"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(unc.astype(np.float64).sum(), float(fx["unc.sum"]), rtol=1e-7)
