"""Trainer end to end on the GPU: torchgan-style model dict, loss plugins resolved by argument name,
checkpoint dict keys, load_model round trip, sample grid."""
import os

import pytest
import torch
import torch.nn as nn
from torch.optim import Adam
from torch.utils.data import DataLoader, TensorDataset

pytestmark = pytest.mark.gpu

import rna_gan_amd as P
from oracle import ref_cpu as R


def network(in_size=32, enc=64):
    return {
        "generator": {"name": P.DCGANGenerator,
                      "args": {"encoding_dims": enc, "out_channels": 3, "step_channels": 64, "out_size": in_size,
                               "nonlinearity": nn.LeakyReLU(0.2), "last_nonlinearity": nn.Tanh()},
                      "optimizer": {"name": Adam, "args": {"lr": 0.0001, "betas": (0.5, 0.999)}}},
        "discriminator": {"name": P.DCGANDiscriminator,
                          "args": {"in_size": in_size, "in_channels": 3, "step_channels": 64,
                                   "nonlinearity": nn.LeakyReLU(0.2), "last_nonlinearity": nn.LeakyReLU(0.2)},
                          "optimizer": {"name": Adam, "args": {"lr": 0.0004, "betas": (0.5, 0.999)}}},
    }


def test_trainer_checkpoint_roundtrip(tmp_path):
    torch.manual_seed(0)
    imgs = R.synthetic_images(32, 32, seed=5)
    loader = DataLoader(TensorDataset(imgs, torch.zeros(32)), batch_size=8)
    losses = [P.WassersteinGeneratorLoss(), P.WassersteinDiscriminatorLoss(clip=(-0.01, 0.01)),
              P.WassersteinGradientPenalty()]
    ck = str(tmp_path / "gan")
    tr = P.Trainer(network(), losses, checkpoints=ck, sample_size=16, epochs=2, devices=[0],
                   recon=str(tmp_path / "img"), nrow=4)
    assert isinstance(tr.optimizer_generator, P.Adam) and isinstance(tr.optimizer_generator, Adam)
    tr(loader)
    assert tr.batch_size == 8
    assert os.path.exists(ck + "0.model") and os.path.exists(ck + "1.model")
    assert os.path.exists(str(tmp_path / "img" / "epoch1_generator.png"))
    assert tr.loss_information["generator_iters"] == 8 and tr.loss_information["discriminator_iters"] == 16
    assert len(tr.loss_logs["WassersteinGradientPenalty"]) == 8
    sd = torch.load(ck + "1.model", map_location="cpu", weights_only=False)
    for key in ("epoch", "loss_information", "loss_objects", "metric_objects", "loss_logs", "metric_logs",
                "generator", "discriminator", "optimizer_generator", "optimizer_discriminator"):
        assert key in sd, key
    assert sd["epoch"] == 2
    assert float(sd["optimizer_discriminator"]["state"][0]["step"]) == 16.0
    # weights were clamped + moved, all finite
    for p in list(tr.generator.parameters()) + list(tr.discriminator.parameters()):
        assert torch.isfinite(p).all()
    assert float(tr.discriminator.flat.data.abs().max()) <= 0.02      # clamp(0.01) + two Adam steps
    # round trip into a fresh trainer (and into the ORACLE modules: same keys/shapes)
    tr2 = P.Trainer(network(), [P.WassersteinGeneratorLoss(), P.WassersteinDiscriminatorLoss(),
                                P.WassersteinGradientPenalty()], checkpoints=str(tmp_path / "gan2"),
                    sample_size=16, epochs=2, recon=str(tmp_path / "img2"))
    tr2.load_model(load_path=ck + "1.model")
    assert tr2.start_epoch == 2
    for a, b in zip(tr.generator.state_dict().values(), tr2.generator.state_dict().values()):
        assert torch.equal(a.cpu(), b.cpu())
    Go = R.OracleDCGANGenerator(64, 32, 3, 64)
    Do = R.OracleDCGANDiscriminator(32, 3, 64)
    Go.load_state_dict(sd["generator"]); Do.load_state_dict(sd["discriminator"])
    opt = torch.optim.Adam(Do.parameters(), lr=4e-4, betas=(0.5, 0.999))
    opt.load_state_dict(sd["optimizer_discriminator"])          # reference-side optimizer accepts our state
    # eval-mode sample from the product generator == oracle generator in eval mode (bf16 tolerance)
    z = torch.randn(4, 64)
    tr.generator.eval(); Go.eval()
    with torch.no_grad():
        a = tr.generator(z.cuda()).cpu(); b = Go(z)
    tr.generator.train()
    assert float((a - b).abs().max()) <= 5e-2


@pytest.mark.gpu
def test_generate_images_matches_oracle():
    """Conditioned generator-only inference (src/gan_utils.py:197-244): chunks of 10 in train mode, betaVAE-conditioned
    noise, un-normalised NHWC export -- product (fp32 kernels) against the oracle restatement with the same CPU draws."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from oracle import ref_cpu as R
    import rna_gan_amd as P
    from rna_gan_amd.gan_utils import generate_images
    E_, S = 32, 32
    Go = R.OracleDCGANGenerator(E_, S, 3, 4)
    R.seeded_fill_(Go, 61)
    Go.train()
    bvo = R.OracleBetaVAE(48, E_, [40, 36, E_], [36, 40], beta=0.005) if hasattr(R, "OracleBetaVAE") else None
    Gp = P.DCGANGenerator(E_, S, 3, 4)
    Gp.load_state_dict(Go.state_dict())
    Gp = Gp.cuda().train().set_precision("fp32")
    tr = SimpleNamespace(generator=Gp, device=torch.device("cuda:0"))
    if bvo is not None:
        R.seeded_fill_(bvo, 62)
        bvo.eval()
        bvp = P.betaVAE(48, E_, [40, 36, E_], [36, 40], beta=0.005)
        bvp.load_state_dict(bvo.state_dict())
        bvp.eval()
        rna = R.synthetic_rna(1, 48, seed=63, distinct=1)
        torch.manual_seed(5)
        ref = R.generate_images(Go, gene_exp=rna, sample_size=23, betavae=bvo)
        torch.manual_seed(5)
        out = generate_images(tr, gene_exp=rna, sample_size=23, betavae=bvp)
        assert out.shape == ref.shape == (23, S, S, 3) and out.dtype == np.float32
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-3)
        assert 0.0 <= out.min() and out.max() <= 1.0
    # unconditioned branch: randn on the device in the product, so compare the plumbing with identical noise
    z = torch.randn(20, E_)
    with torch.no_grad():
        ref = torch.cat([Go(c) for c in torch.split(z, 10)], 0)
    ref = ((ref + 1) / 2).permute(0, 2, 3, 1).numpy()
    ops, _ = Gp.runtime()
    imgs = torch.cat([Gp(c.cuda()) for c in torch.split(z, 10)], 0)
    np.testing.assert_allclose(ops.export_images_nhwc(imgs.contiguous()).cpu().numpy(), ref, rtol=0, atol=2e-3)


@pytest.mark.gpu
def test_fid_proxy_and_feature_matching():
    """feature_matching=True returns the trunk activation in front of the head (torchgan contract); the Frechet-distance
    proxy built on it is ~0 between a set and itself and clearly positive between images and noise."""
    import numpy as np
    import torch
    from oracle import ref_cpu as R
    import rna_gan_amd as P
    from rna_gan_amd import fid
    Do = R.OracleDCGANDiscriminator(32, 3, 16)
    R.seeded_fill_(Do, 71)
    D = P.DCGANDiscriminator(32, 3, 16)
    D.load_state_dict(Do.state_dict())
    D = D.cuda().set_precision("fp32")
    x = R.synthetic_images(24, 32, seed=72)
    # eval-mode trunk features against the oracle module's trunk (running statistics)
    Do.eval()
    with torch.no_grad():
        want = Do.model(x)
    D.eval()
    got = D(x.cuda(), feature_matching=True).cpu()
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-3, atol=2e-4)
    same = fid.fid_proxy(D, x, x.clone(), batch_size=8)
    other = fid.fid_proxy(D, x, torch.randn(24, 3, 32, 32).clamp(-1, 1), batch_size=8)
    assert abs(same) < 1e-6 and other > 1e-3, (same, other)
    assert not D.training      # mode restored


@pytest.mark.gpu
def test_graphs_survive_noop_move_and_optimizer_reload():
    """Captured step graphs hold raw buffer addresses.  (a) generator.to(device) on a module that already lives there
    (generate_images does it on every call) keeps the flat buffers -- same generation, graphs still valid; (b)
    Adam.load_state_dict after training has started copies the loaded moments INTO the existing buffers, so the
    replayed graph steps from them: the run equals an eager (RNAGAN_GRAPHS=0 semantics) run doing the same."""
    import copy
    from rna_gan_amd import graphed, losses as PL
    enc, size, n = 32, 32, 8

    def make():
        G = P.DCGANGenerator(enc, size, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
        D = P.DCGANDiscriminator(size, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
        R.seeded_fill_(G, 1); R.seeded_fill_(D, 2)
        G.set_precision("fp32"); D.set_precision("fp32")
        G, D = G.cuda().train(), D.cuda().train()
        og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
        od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
        return G, D, og, od

    real = R.synthetic_images(n, size, seed=3).cuda()
    noises = [R.synthetic_normal(n, enc, seed=10 + k).cuda() for k in range(12)]
    eps = torch.tensor([0.4], device="cuda")

    def run(use_graphs):
        was = graphed.ENABLED
        graphed.ENABLED = use_graphs
        try:
            G, D, og, od = make()
            lg, ld, lp = P.WassersteinGeneratorLoss(), P.WassersteinDiscriminatorLoss(), P.WassersteinGradientPenalty()
            saved = None
            for it in range(6):
                lg.step(G, D, og, noises[2 * it])
                ld.step(G, D, od, real, noises[2 * it + 1])
                lp.step(G, D, od, real, noises[2 * it + 1], eps)
                if it == 1:
                    saved = (copy.deepcopy(og.state_dict()), copy.deepcopy(od.state_dict()))
                if it == 3:                      # graphs are captured by now (third call)
                    gen_before = (G.flat.gen, D.flat.gen)
                    G.to(torch.device("cuda:0")); D.cuda()
                    assert (G.flat.gen, D.flat.gen) == gen_before, "no-op move must not re-home the parameters"
                    og.load_state_dict(saved[0]); od.load_state_dict(saved[1])
                    assert og.state_dict()["state"][0]["step"] == saved[0]["state"][0]["step"]
            torch.cuda.synchronize()
            return [p.detach().cpu().clone() for p in list(G.parameters()) + list(D.parameters())], og, od
        finally:
            graphed.ENABLED = was

    pe, _, _ = run(False)
    pg, og, od = run(True)
    for a, b in zip(pe, pg):
        assert torch.allclose(a, b, rtol=0, atol=1e-6), float((a - b).abs().max())
    # steps counted from the reloaded value: 2 iterations before the save, reload at it == 3, 2 more iterations
    assert float(og.state_dict()["state"][0]["step"]) == 2 + 2
    assert float(od.state_dict()["state"][0]["step"]) == 4 + 4


@pytest.mark.gpu
def test_latent_is_encoded_once_per_batch():
    """The three betaVAE-conditioned loss plugins share one encode per batch (rna_gan_amd.losses._LatentCache): equal
    results with the cache on and off, 1 miss + 2 hits per iteration, and a different RNA tensor or a plugin whose
    encoder weights differ always re-encodes.  Encoders are identified by the PROVENANCE of their weights (one state_dict /
    one checkpoint file loaded into all three, untouched since), never by a fingerprint of the values: three encoders filled
    independently with equal values do not share."""
    from rna_gan_amd import losses as PL
    enc, size, n, F = 2048, 32, 8, 96

    def run(cache, shared=True):
        PL.LATENT_CACHE = cache
        PL.new_batch()
        G = P.DCGANGenerator(enc, size, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
        D = P.DCGANDiscriminator(size, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
        R.seeded_fill_(G, 1); R.seeded_fill_(D, 2)
        G, D = G.cuda().train(), D.cuda().train()
        og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
        od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
        ls = [P.WassersteinGeneratorLossVAE(None, F), P.WassersteinDiscriminatorLossVAE(None, F),
              P.WassersteinGradientPenaltyVAE(None, F)]
        R.seeded_fill_(ls[0].betavae, 5)
        sd = ls[0].betavae.state_dict()
        for l in ls:
            if l is not ls[0]:
                if shared:
                    l.betavae.load_state_dict(sd)          # same weights by provenance
                else:
                    R.seeded_fill_(l.betavae, 5)           # equal values, independent fills: never shared
            l.betavae = l.betavae.cuda().eval()
        real = R.synthetic_images(n, size, seed=3).cuda()
        out = []
        for it in range(3):
            PL.new_batch()
            rna = R.synthetic_rna(n, F, seed=20 + it, distinct=4).cuda()
            u = [R.synthetic_uniform(n, enc, seed=30 + 3 * it + j).cuda() for j in range(3)]
            out += [ls[0].step(G, D, og, rna, u[0]).item(), ls[1].step(G, D, od, real, rna, u[1]).item(),
                    ls[2].step(G, D, od, real, rna, u[2], torch.tensor([0.3], device="cuda")).item()]
        return out, ls, (G, D, og, od, real)

    try:
        h0, m0 = PL._LATENT.hits, PL._LATENT.misses
        a, ls, (G, D, og, od, real) = run(True)
        assert PL._LATENT.misses - m0 == 3 and PL._LATENT.hits - h0 == 6
        b, _, _ = run(False)
        assert a == b
        h2, m2 = PL._LATENT.hits, PL._LATENT.misses
        c, _, _ = run(True, shared=False)
        assert a == c and PL._LATENT.misses - m2 == 9 and PL._LATENT.hits == h2
        PL.LATENT_CACHE = True
        PL.new_batch()
        rna = R.synthetic_rna(n, F, seed=50, distinct=4).cuda()
        u = R.synthetic_uniform(n, enc, seed=51).cuda()
        m1 = PL._LATENT.misses
        ls[0].step(G, D, og, rna, u)
        ls[1].step(G, D, od, real, rna.clone(), u)             # another tensor with equal contents: re-encoded
        assert PL._LATENT.misses - m1 == 2
        with torch.no_grad():
            ls[2].betavae.z_mu.bias.add_(1.0)
        ls[2].betavae.weights_changed()
        ls[2].step(G, D, od, real, rna.clone(), u, torch.tensor([0.3], device="cuda"))   # different encoder weights
        assert PL._LATENT.misses - m1 == 3
    finally:
        PL.LATENT_CACHE = True
        PL.new_batch()


@pytest.mark.gpu
def test_generate_images_reference_fixture(golden_dir):
    """Product generate_images (HIP kernels, fp32 path) against f9_generate_images.npz = the output of the REFERENCE's
    gan_utils.generate_images on the same seeded generator / betaVAE weights and the same torch seed."""
    import numpy as np
    from types import SimpleNamespace
    from rna_gan_amd.gan_utils import generate_images
    fx = np.load(os.path.join(golden_dir, "f9_generate_images.npz"))
    E_, F = 16, 40
    Go = R.seeded_fill_(R.OracleDCGANGenerator(E_, 256, 3, 1, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 81)
    Gp = P.DCGANGenerator(E_, 256, 3, 1, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    Gp.load_state_dict(Go.state_dict())
    Gp = Gp.cuda().train().set_precision("fp32")
    bvo = R.seeded_fill_(R.OracleBetaVAE(F, E_, [32, 24, E_], [24, 32], beta=0.0005), 82)
    bvp = P.betaVAE(F, E_, [32, 24, E_], [24, 32], beta=0.0005)
    bvp.load_state_dict(bvo.state_dict())
    bvp.set_precision("fp32").eval()
    tr = SimpleNamespace(generator=Gp, device=torch.device("cuda:0"))
    rna = R.synthetic_rna(1, F, seed=83, distinct=1)
    torch.manual_seed(5)
    cond = generate_images(tr, gene_exp=rna, sample_size=13, betavae=bvp)
    assert cond.shape == (13, 256, 256, 3) and cond.dtype == np.float32
    np.testing.assert_allclose(cond[:, ::16, ::16, :], fx["cond.sub"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(cond.astype(np.float64).sum(), float(fx["cond.sum"]), rtol=2e-4)
    for k, v in Gp.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), fx["bn_after_cond." + k], rtol=2e-3, atol=2e-5, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_betavae_encode_reference_fixtures(golden_dir, precision, tol):
    """betaVAE.encode (eval) on the GPU against what the REFERENCE's betaVAE.encode produced: the reduced model in full
    (f1_betavae_small.npz) and the full-size 19198 -> 2048 encoder through first / last 64 columns and row sums
    (f1_betavae_full.npz)."""
    import numpy as np
    fx = np.load(os.path.join(golden_dir, "f1_betavae_small.npz"))
    m = P.betaVAE(64, 16, [48, 32, 16], [32, 48], beta=0.005)
    R.seeded_fill_(m, 11)
    m = m.set_precision(precision).cuda().eval()
    zm, zl, h = m.encode(R.synthetic_rna(6, 64, seed=12, distinct=4).cuda())
    for got, key in ((zm, "z_mean"), (zl, "z_logvar"), (h[:, :16], "x_encoded")):
        want = fx[key]
        assert float(np.abs(got.cpu().numpy() - want).max()) <= tol * float(np.abs(want).max()), key
    fx = np.load(os.path.join(golden_dir, "f1_betavae_full.npz"))
    m = P.betaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000], beta=0.005)
    R.seeded_fill_(m, 13)
    m = m.set_precision(precision).cuda().eval()
    zm, _, _ = m.encode(R.synthetic_rna(4, 19198, seed=14, distinct=4).cuda(), mean_only=True)
    zc = zm.cpu().double().numpy()
    scale = float(np.abs(fx["z_mean_first"]).max())
    assert float(np.abs(zc[:, :64] - fx["z_mean_first"]).max()) <= tol * scale
    assert float(np.abs(zc[:, -64:] - fx["z_mean_last"]).max()) <= tol * scale
    np.testing.assert_allclose((zc ** 2).sum(1), fx["z_mean_sumsq"], rtol=4 * tol)


def test_penalty_fake_from_the_d_step_generator_pass():
    """D-loss step with next_noise: the penalty step's fake batch comes out of the D-loss step's generator pass (one double
    batch) and the penalty step picks it up; same training as three independent train_ops with the same noise (within the
    bf16 noise of different tile shapes), and a penalty step called with other noise ignores the cached batch."""
    import torch.nn as nn
    import rna_gan_amd as P
    from rna_gan_amd import losses as PL
    from oracle import ref_cpu as R
    in_size, step, enc, n = 64, 64, 128, 8
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    res = []
    for ahead in (False, True):
        G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
        D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
        G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
        G, D = G.cuda().train(), D.cuda().train()
        og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
        od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
        lg, ld, lp = PL.WassersteinGeneratorLoss(), PL.WassersteinDiscriminatorLoss(), PL.WassersteinGradientPenalty()
        out = []
        for it in range(5):                       # eager, eager, then captured graphs
            PL.new_batch()
            real = R.synthetic_images(n, in_size, seed=100 + it).cuda()
            nz = [R.synthetic_normal(n, enc, seed=200 + 3 * it + j).cuda() for j in range(3)]
            eps = torch.tensor([0.1 + 0.2 * it], device="cuda")
            out.append(lg.step(G, D, og, nz[0]).item())
            out.append(ld.step(G, D, od, real, nz[1], next_noise=nz[2] if ahead else None).item())
            if ahead:
                assert PL._FAKE.key is not None
            out.append(lp.step(G, D, od, real, nz[2], eps).item())
            assert PL._FAKE.key is None
        if ahead:                                 # a cached batch for OTHER noise is not used
            real = R.synthetic_images(n, in_size, seed=300).cuda()
            a, b, c = (R.synthetic_normal(n, enc, seed=400 + j).cuda() for j in range(3))
            ld.step(G, D, od, real, a, next_noise=b)
            key = PL._FAKE.key
            lp.step(G, D, od, real, c, torch.tensor([0.5], device="cuda"))
            assert PL._FAKE.key == key
            PL.new_batch()
            assert PL._FAKE.key is None
        res.append((out, G.flat.data.float().cpu().clone(), D.flat.data.float().cpu().clone()))
    (la, ga, da), (lb, gb, db) = res
    for i, (x, y) in enumerate(zip(la[:9], lb[:9])):          # first three iterations: before the trajectories drift apart
        assert abs(x - y) <= (0.35 if i % 3 == 2 else 5e-2) * (abs(x) + 1.0), (i, la, lb)
    assert float((ga - gb).norm() / ga.norm()) <= 2e-2 and float((da - db).norm() / da.norm()) <= 2e-2


def _mixed_tissue_config(tmp_path, n_per_tissue=3):
    """Two tissue tables with their own tile stores (32 x 32 tiles, 48 genes) and the CLI config that names them."""
    import json
    import numpy as np
    import pandas as pd
    from rna_gan_amd import data as PD
    rng = np.random.default_rng(11)
    genes = ["rna_G%d" % i for i in range(48)]
    csvs, roots = [], []
    for t, tissue in enumerate(("lung", "brain")):
        names = ["%s_%d.svs" % (tissue, i) for i in range(n_per_tissue)]
        df = pd.DataFrame(rng.gamma(2.0, 3.0 + 4.0 * t, size=(n_per_tissue, len(genes))), columns=genes)
        df.insert(0, "wsi_file_name", names)
        csv = str(tmp_path / (tissue + ".csv"))
        df.to_csv(csv, index=False)
        root = str(tmp_path / ("patches_" + tissue))
        for wsi in names:
            tiles = [rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8) for _ in range(8)]
            PD.write_tile_store(os.path.join(root, wsi, wsi.replace(".svs", "")), tiles, slide_id=wsi)
        csvs.append(csv); roots.append(root)
    cfg = {"path_csv": csvs, "patch_data_path": roots, "img_size": 32, "rna_features": len(genes),
           "save_dir": str(tmp_path / "ck"), "flag": "mixed"}
    cfg_path = str(tmp_path / "cfg.json")
    with open(cfg_path, "w") as f:
        json.dump(cfg, f)
    return cfg_path


def _cli_cmd(tmp_path, cfg_path):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return [sys.executable, os.path.join(repo, "histopathology_gan.py"), "--config", cfg_path, "--loss_type", "wganvae",
            "--num_epochs", "1", "--num_patches", "8", "--batch_size", "8", "--model_dir", str(tmp_path / "model"),
            "--image_dir", str(tmp_path / "img"), "--betavae_checkpoint", str(tmp_path / "none.pt")]


def _checkpoint_files(tmp_path):
    cks = [f for f in os.listdir(str(tmp_path)) if f.startswith("model")] + \
          ([os.path.join("model", f) for f in os.listdir(str(tmp_path / "model"))] if os.path.isdir(str(tmp_path / "model")) else [])
    return [os.path.join(str(tmp_path), c) for c in cks if os.path.isfile(os.path.join(str(tmp_path), c))]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_cli_trains_on_mixed_tissue_tables(tmp_path, precision):
    """The reference CLI's real-data path end to end (src/histopathology_gan.py:111-168, BASELINE configs[3]'s data side): two
    tissue tables with their own tile stores -> concatenated table -> log / standardised RNA -> per-slide tile sampling ->
    DataLoader -> Trainer with the three betaVAE-conditioned plugins, on the HIP kernels.  One short epoch at 32 x 32; checks
    the run completes, the checkpoint has the reference's keys and the logged losses are finite."""
    import subprocess
    import numpy as np
    cfg_path = _mixed_tissue_config(tmp_path)
    # (precision fp16: BASELINE configs[3] in its stated dtype -- fp16 storage, fp16 betaVAE encoder GEMMs, loss-scaled backward)
    r = subprocess.run(_cli_cmd(tmp_path, cfg_path) + ["--precision", precision], capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Training of the Model is Complete" in r.stdout
    vals = [float(l.split(":")[1]) for l in r.stdout.splitlines() if "Mean Loss" in l]
    assert len(vals) >= 2 and all(np.isfinite(v) for v in vals)
    files = _checkpoint_files(tmp_path)
    assert files, "no checkpoint written"
    ck = torch.load(files[0], map_location="cpu", weights_only=False)
    assert {"epoch", "generator", "discriminator", "optimizer_generator", "optimizer_discriminator"} <= set(ck)
    for net in ("generator", "discriminator"):
        for k, v in ck[net].items():
            assert not v.dtype.is_floating_point or (v.dtype == torch.float32 and bool(torch.isfinite(v).all())), (net, k)


def test_cli_stock_wgan_literal_command(tmp_path):
    """BASELINE configs[0]'s literal command (src/histopathology_gan.py:94,267-272): ``--gan_type dcgan --loss_type wgan`` -- the
    STOCK torchgan losses (randn noise on the device, tensor batches, weight clamp (-0.01, 0.01) before the D step) at the
    reference's hard-coded batch 8 on 256 x 256 synthetic tiles, reference model size, through the CLI.  Three iterations;
    checks completion, the checkpoint dictionary's keys, three logged values per plugin, finite weights, and that the
    discriminator's parameters sit inside the clamp interval up to the TWO Adam steps (lr 4e-4; D-loss and penalty train_ops)
    taken after the last clamp -- early Adam steps are up to ~3 lr per element (bias-corrected m / sqrt(v) with betas (0.5, 0.999)
    exceeds 1 when a gradient shrinks), an unclamped weight of this layer would be ~0.2."""
    import subprocess
    import sys
    import numpy as np
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(repo, "histopathology_gan.py"), "--config", os.path.join(repo, "configs", "gan_run_synthetic.json"),
           "--gan_type", "dcgan", "--loss_type", "wgan", "--synthetic", "--num_epochs", "1", "--steps_per_epoch", "3",
           "--model_dir", str(tmp_path / "model"), "--image_dir", str(tmp_path / "img")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Training of the Model is Complete" in r.stdout
    vals = [float(l.split(":")[1]) for l in r.stdout.splitlines() if "Mean Loss" in l]
    assert len(vals) == 2 and all(np.isfinite(v) for v in vals)
    files = _checkpoint_files(tmp_path)
    assert files, "no checkpoint written"
    ck = torch.load(files[0], map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "loss_information", "loss_objects", "metric_objects", "loss_logs", "metric_logs", "generator",
                       "discriminator", "optimizer_generator", "optimizer_discriminator"}
    assert ck["epoch"] == 1
    assert list(ck["loss_logs"]) == ["WassersteinGeneratorLoss", "WassersteinDiscriminatorLoss", "WassersteinGradientPenalty"]
    assert all(len(v) == 3 and all(np.isfinite(x) for x in v) for v in ck["loss_logs"].values())
    assert ck["loss_information"]["generator_iters"] == 3 and ck["loss_information"]["discriminator_iters"] == 6
    assert ck["generator"]["model.0.0.weight"].shape == (2048, 2048, 4, 4) and ck["discriminator"]["disc.0.weight"].shape == (1, 2048, 4, 4)
    for k, v in ck["discriminator"].items():
        if v.dtype.is_floating_point and "running" not in k:
            assert float(v.abs().max()) <= 0.01 + 2 * 3 * 4e-4, (k, float(v.abs().max()))
    for k, v in ck["generator"].items():
        assert not v.dtype.is_floating_point or bool(torch.isfinite(v).all()), k
    assert os.path.exists(str(tmp_path / "img" / "epoch1_generator.png"))


def test_cli_two_ranks_on_the_shared_gpu(tmp_path):
    """The same CLI run as TWO data-parallel ranks (the box has one GPU: both ranks use it and all-reduce over gloo,
    RNAGAN_DIST_BACKEND): equal-length rank shards of the slide table, rank 0's parameters broadcast, the betaVAE broadcast,
    the plugins' data-parallel route inside Trainer.train (prefetcher + pipelined train_ops on top), checkpoint and console
    summary from rank 0 only; both ranks finish."""
    import socket
    import subprocess
    import numpy as np
    cfg_path = _mixed_tissue_config(tmp_path, n_per_tissue=4)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RNAGAN_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(_cli_cmd(tmp_path, cfg_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      cwd=str(tmp_path)))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-1500:] + err[-2500:]
        outs.append(out)
    assert "Training of the Model is Complete" in outs[0] and "Training of the Model is Complete" not in outs[1]
    vals = [float(l.split(":")[1]) for l in outs[0].splitlines() if "Mean Loss" in l]
    assert len(vals) >= 2 and all(np.isfinite(v) for v in vals)
    files = _checkpoint_files(tmp_path)
    assert files, "no checkpoint written by rank 0"
    ck = torch.load(files[0], map_location="cpu", weights_only=False)
    assert {"epoch", "generator", "discriminator", "optimizer_generator", "optimizer_discriminator"} <= set(ck)
    for k, v in ck["generator"].items():
        assert not v.dtype.is_floating_point or bool(torch.isfinite(v).all()), k


def test_pipelined_train_iter_matches_the_synchronous_loop(tmp_path):
    """Trainer(pipeline=True) takes train_op k's .item() after launching train_op k + 1; the launches, the random draws and
    therefore every logged loss value are those of the synchronous loop (pipeline=False), bit for bit, through the eager
    iterations and the graph replays; a plugin whose train_ops is overridden is called synchronously as written."""
    import random
    logs = {}
    for pipeline in (False, True):
        torch.manual_seed(3); random.seed(3)
        imgs = R.synthetic_images(48, 32, seed=9)
        loader = DataLoader(TensorDataset(imgs, torch.zeros(48)), batch_size=8)
        losses = [P.WassersteinGeneratorLoss(), P.WassersteinDiscriminatorLoss(), P.WassersteinGradientPenalty()]
        tr = P.Trainer(network(), losses, checkpoints=str(tmp_path / ("p%d" % pipeline)), sample_size=4, epochs=1,
                       recon=None, pipeline=pipeline)
        assert tr.pipeline is pipeline
        tr(loader)
        assert tr.loss_information["generator_iters"] == 6 and tr.loss_information["discriminator_iters"] == 12
        logs[pipeline] = ({k: list(v) for k, v in tr.loss_logs.items()}, dict(tr.loss_information),
                          {k: v.detach().cpu().clone() for k, v in tr.generator.state_dict().items()})
    a, b = logs[False], logs[True]
    assert a[0] == b[0] and all(isinstance(x, float) for v in b[0].values() for x in v)
    # epoch totals: the same numbers summed in a different grouping (the last train_op of an iteration is added with the next)
    for k in a[1]:
        assert a[1][k] == pytest.approx(b[1][k], rel=1e-12, abs=1e-12), k
    assert all(torch.equal(a[2][k], b[2][k]) for k in a[2])

    calls = []

    class Custom(P.WassersteinDiscriminatorLoss):
        def train_ops(self, generator, discriminator, optimizer_discriminator, real_inputs, device, labels=None):
            calls.append("custom")
            return 1.25
    assert getattr(Custom.train_ops, "_rg_async", None) is None
    tr = P.Trainer.__new__(P.Trainer)
    tr.losses = {"Custom": Custom()}
    tr.loss_logs = {"Custom": []}
    tr.loss_information = {"generator_losses": 0.0, "discriminator_losses": 0.0, "generator_iters": 0, "discriminator_iters": 0}
    tr.ncritic, tr.pipeline = 1, True
    for k in ("generator", "discriminator", "optimizer_discriminator", "real_inputs", "device", "labels"):
        setattr(tr, k, None)
    tr._store_loss_maps()
    assert tr.train_iter() == (0.0, 1.25, 0, 1) and calls == ["custom"] and tr.loss_logs["Custom"] == [1.25]
