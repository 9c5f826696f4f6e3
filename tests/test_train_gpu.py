"""GPU parity of the PRODUCT path (rna_gan_amd modules / losses / optimizer through the C ABI):
  * against the golden fixture produced by the reference's own *LossVAE.train_ops
    (tests/golden/f5_trainops_vae.npz, made by importing /root/reference/src/wgan_loss.py);
  * against the CPU oracle for two full iterations at a mid-size config (fp32 and bf16 paths);
  * full-size (256x256, reference model) one iteration in bf16: finite, losses near the oracle.
"""
import copy
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
import rna_gan_amd as P
from rna_gan_amd import losses as PL

DIGEST_OVER = 20000


def digest(t):
    a = t.detach().float().cpu().numpy()
    if a.size <= DIGEST_OVER:
        return a
    f = a.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), (f * f).sum()], f[:64], f[-64:]])


def l2rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def product_pair(in_size, step, enc, precision, G_src, D_src):
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    G.load_state_dict(G_src.state_dict()); D.load_state_dict(D_src.state_dict())
    G.set_precision(precision); D.set_precision(precision)
    G, D = G.cuda().train(), D.cuda().train()
    og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
    od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
    return G, D, og, od


def test_reference_trainops_fixture(golden_dir):
    """Our *LossVAE.train_ops (HIP, fp32 path) vs what the REFERENCE's train_ops produced."""
    fx = np.load(os.path.join(golden_dir, "f5_trainops_vae.npz"))
    RNA_F, bs = 64, 6
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(2048, 16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 31)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(16, 3, 4, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 21)
    G, D, og, od = product_pair(16, 4, 2048, "fp32", G0, D0)
    bv = R.seeded_fill_(R.OracleBetaVAE(RNA_F, 2048, [6000, 4000, 2048], [4000, 6000]), 51)
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "bv.pt")
        torch.save(bv.state_dict(), ck)
        lg = PL.WassersteinGeneratorLossVAE(checkpoint=ck, rna_features=RNA_F)
        ld = PL.WassersteinDiscriminatorLossVAE(checkpoint=ck, rna_features=RNA_F)
        lp = PL.WassersteinGradientPenaltyVAE(checkpoint=ck, rna_features=RNA_F)
    for l in (lg, ld, lp):
        l.betavae.set_precision("fp32")
    assert isinstance(lg.reduction, str) and lg.override_train_ops == RNA_F and lp.lambd == 10.0 and ld.clip is None
    dev = torch.device("cuda:0")
    call = 0
    for it in range(2):
        batch = {"image": R.synthetic_images(bs, 16, seed=60 + it),
                 "rna_data": R.synthetic_rna(bs, RNA_F, seed=70 + it, distinct=3)}
        for tag, fn in (("g", lambda: lg.train_ops(G, D, og, dev, bs, batch)),
                        ("d", lambda: ld.train_ops(G, D, od, batch, dev)),
                        ("gp", lambda: lp.train_ops(G, D, od, batch, dev))):
            torch.manual_seed(1000 + call)      # the reference run was seeded the same way per call
            loss = fn()
            ref = float(fx[f"loss.{it}.{tag}"])
            assert abs(loss - ref) <= 2e-3 * (abs(ref) + 1e-2), (it, tag, loss, ref)
            call += 1
    for k, v in G.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(fx["G." + k]); continue
        assert l2rel(digest(v), fx["G." + k]) <= 2e-3, "G." + k
    for k, v in D.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(fx["D." + k]); continue
        assert l2rel(digest(v), fx["D." + k]) <= 2e-3, "D." + k
    for nm, opt, mod in (("optG", og, G), ("optD", od, D)):
        sd = opt.state_dict()["state"]
        for i, (k, _) in enumerate(mod.named_parameters()):
            assert float(sd[i]["step"]) == float(fx[f"{nm}.{k}.step"])
            assert l2rel(digest(sd[i]["exp_avg"]), fx[f"{nm}.{k}.exp_avg"]) <= 5e-3, f"{nm}.{k}.exp_avg"
            assert l2rel(digest(sd[i]["exp_avg_sq"]), fx[f"{nm}.{k}.exp_avg_sq"]) <= 1e-2, f"{nm}.{k}.exp_avg_sq"


@pytest.mark.parametrize("precision,tol_loss,tol_p", [("fp32", 2e-3, 2e-3), ("bf16", 4e-2, 2.5e-1)])
def test_two_iterations_vs_oracle(precision, tol_loss, tol_p):
    """Two full iterations (3 optimizer steps each) at in_size 32 / step 64 / enc 128, batch 16.
    Compared quantity for parameters: the UPDATE (p_after - p_before), L2-relative."""
    in_size, step, enc, n = 32, 64, 128, 16
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    G, D, og, od = product_pair(in_size, step, enc, precision, G0, D0)
    for it in range(2):
        real = R.synthetic_images(n, in_size, seed=100 + it)
        noises = [R.synthetic_normal(n, enc, seed=200 + 3 * it + j) for j in range(3)]
        eps = 0.25 + 0.5 * it
        ref = R.train_iteration(Go, Do, ogo, odo, real, noises, eps, clip=(-0.01, 0.01) if it == 1 else None)
        rd = real.cuda()
        lg = PL._g_step(G, D, og, noises[0].cuda()).item()
        ld = PL._d_step(G, D, od, rd, noises[1].cuda(), (-0.01, 0.01) if it == 1 else None).item()
        lp = PL._gp_step(G, D, od, rd, noises[2].cuda(), eps, 10.0).item()
        for got, want, nm in ((lg, ref["g"], "g"), (ld, ref["d"], "d"), (lp, ref["gp"], "gp")):
            # losses are means/differences of O(1) critic outputs: tolerance relative to that scale
            assert np.isfinite(got) and abs(got - want) <= tol_loss * (abs(want) + 0.5), (it, nm, got, want)
        if it == 0:
            for mod, ref_mod, src in ((G, Go, G0), (D, Do, D0)):
                for (k, p), (_, q), (_, s) in zip(mod.named_parameters(), ref_mod.named_parameters(),
                                                  src.named_parameters()):
                    du, dr = p.detach().cpu() - s.detach(), q.detach() - s.detach()
                    # Adam's first steps move every weight by ~lr*sign(g): compare update directions
                    cos = float((du * dr).sum() / (du.norm() * dr.norm() + 1e-30))
                    assert cos >= 1 - tol_p, (k, cos)
    for mod, ref_mod in ((G, Go), (D, Do)):
        for (k, b), (_, q) in zip(mod.named_buffers(), ref_mod.named_buffers()):
            if k.endswith("num_batches_tracked"):
                assert int(b) == int(q), k
            else:
                assert l2rel(b.cpu().numpy(), q.numpy()) <= (5e-4 if precision == "fp32" else 3e-2), k


def test_full_size_one_iteration_bf16():
    """Reference model size (enc 2048, step 64, 256x256), batch 8, bf16 MFMA path: runs, finite,
    losses close to the fp32 CPU oracle."""
    in_size, step, enc, n = 256, 64, 2048, 8
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 17)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 18)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    G, D, og, od = product_pair(in_size, step, enc, "bf16", G0, D0)
    real = R.synthetic_images(n, in_size, seed=300)
    noises = [R.synthetic_normal(n, enc, seed=400 + j) for j in range(3)]
    ref = R.train_iteration(Go, Do, ogo, odo, real, noises, 0.4)
    rd = real.cuda()
    lg = PL._g_step(G, D, og, noises[0].cuda()).item()
    ld = PL._d_step(G, D, od, rd, noises[1].cuda(), None).item()
    lp = PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.4, 10.0).item()
    print("full-size losses hip/ref:", lg, ref["g"], ld, ref["d"], lp, ref["gp"])
    for got, want in ((lg, ref["g"]), (ld, ref["d"]), (lp, ref["gp"])):
        assert np.isfinite(got) and abs(got - want) <= 6e-2 * (abs(want) + 0.1)
    for p in list(G.parameters()) + list(D.parameters()):
        assert torch.isfinite(p).all()
    img = G(noises[0].cuda())
    assert img.shape == (n, 3, 256, 256) and torch.isfinite(img).all() and float(img.abs().max()) <= 1.0


def test_full_size_one_iteration_fp32():
    """The reference's own configuration -- model size of src/histopathology_gan.py:178-192, its hard-coded batch 8 (:94), fp32
    arithmetic (src/betaVAE.py:184,223,230-236) -- on the fp32 mode, whose convolutions run on the f32 matrix cores
    (gemm_mfma32_kernel: 128 x 128 and 64 x 64 tiles, split-K weight gradients, every layer shape of the model): one iteration
    against the fp32 CPU oracle.  Losses within 3e-3 (|want| + 0.1) (measured 2e-5 / 7e-4 / 8e-4); direction of the Adam update of
    every parameter tensor (cosine >= 0.95 for tensors of >= 4096 elements, measured >= 0.974: the first Adam steps are lr *
    sign(g), so the cosine counts sign agreements -- two fp32 evaluations with different summation orders disagree on the
    elements whose gradient is within rounding noise of zero, and the discriminator takes two such steps)."""
    in_size, step, enc, n = 256, 64, 2048, 8
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 17)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 18)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    G, D, og, od = product_pair(in_size, step, enc, "fp32", G0, D0)
    real = R.synthetic_images(n, in_size, seed=300)
    noises = [R.synthetic_normal(n, enc, seed=400 + j) for j in range(3)]
    ref = R.train_iteration(Go, Do, ogo, odo, real, noises, 0.4)
    rd = real.cuda()
    lg = PL._g_step(G, D, og, noises[0].cuda()).item()
    ld = PL._d_step(G, D, od, rd, noises[1].cuda(), None).item()
    lp = PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.4, 10.0).item()
    print("full-size fp32 losses hip/ref:", lg, ref["g"], ld, ref["d"], lp, ref["gp"])
    for got, want in ((lg, ref["g"]), (ld, ref["d"]), (lp, ref["gp"])):
        assert np.isfinite(got) and abs(got - want) <= 3e-3 * (abs(want) + 0.1), (got, want)
    worst = (1.0, None)
    for tag, mod, mod0, modo in (("G", G, G0, Go), ("D", D, D0, Do)):
        sd, sdo, sd0 = mod.state_dict(), modo.state_dict(), mod0.state_dict()
        for name, _ in mod0.named_parameters():
            w0 = sd0[name].double()
            du_hip, du_ref = sd[name].cpu().double() - w0, sdo[name].double() - w0
            cos = float((du_hip * du_ref).sum() / (du_hip.norm() * du_ref.norm() + 1e-30))
            worst = min(worst, (cos, tag + "." + name))
            assert cos >= (0.95 if w0.numel() >= 4096 else 0.85), (tag, name, cos)
    print("worst update cosine (fp32 mode):", worst)
    for mod, ref_mod in ((G, Go), (D, Do)):
        for (k, b), (_, q) in zip(mod.named_buffers(), ref_mod.named_buffers()):
            if not k.endswith("num_batches_tracked"):
                # (the penalty step's forward runs behind the discriminator's first Adam step, whose sign-like update differs
                # in the elements counted above: 7e-4 measured on the first block's running mean, 2.6e-3 on the fourth's, whose
                # means are near zero)
                assert l2rel(b.cpu().numpy(), q.numpy()) <= 1e-2, k
            else:
                assert int(b) == int(q), k


def test_full_size_batch64_bf16():
    """BASELINE configs[1] shape exactly (batch 64 at the reference model size): this is where the MFMA launchers pick
    their large-batch variants (256x256 tiles, class-fastest block order, 512-block wgrad grids, row-staged image-side
    kernels).  One iteration against the fp32 CPU oracle: the three losses and the direction of the Adam update of
    every parameter tensor."""
    in_size, step, enc, n = 256, 64, 2048, 64
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 27)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 28)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    G, D, og, od = product_pair(in_size, step, enc, "bf16", G0, D0)
    real = R.synthetic_images(n, in_size, seed=310)
    noises = [R.synthetic_normal(n, enc, seed=410 + j) for j in range(3)]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = R.train_iteration(Go, Do, ogo, odo, real, noises, 0.35)
    rd = real.cuda()
    lg = PL._g_step(G, D, og, noises[0].cuda()).item()
    ld = PL._d_step(G, D, od, rd, noises[1].cuda(), None).item()
    lp = PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.35, 10.0).item()
    print("batch-64 losses hip/ref:", lg, ref["g"], ld, ref["d"], lp, ref["gp"])
    for got, want in ((lg, ref["g"]), (ld, ref["d"]), (lp, ref["gp"])):
        assert np.isfinite(got) and abs(got - want) <= 6e-2 * (abs(want) + 0.1)
    # EVERY parameter tensor: direction of the accumulated update (G: one Adam step, D: two).  The first Adam step is
    # lr * sign(g) element-wise, so the cosine is (sign agreements - disagreements) / numel: with the 10-15 % relative
    # gradient noise of bf16 storage (test_bf16_kernels_vs_bf16_rounding_twin) ~5-8 % of the signs flip (elements with
    # |g| inside the noise) -> cosine ~0.85-0.9; a wrong kernel on any layer drives its tensor's cosine towards 0.
    worst = (1.0, None)
    for tag, mod, mod0, modo in (("G", G, G0, Go), ("D", D, D0, Do)):
        sd, sdo, sd0 = mod.state_dict(), modo.state_dict(), mod0.state_dict()
        for name, _ in mod0.named_parameters():
            w0 = sd0[name].double()
            du_hip = sd[name].cpu().double() - w0
            du_ref = sdo[name].double() - w0
            cos = float((du_hip * du_ref).sum() / (du_hip.norm() * du_ref.norm() + 1e-30))
            print("%s.%-22s numel %9d  update cosine %.4f" % (tag, name, w0.numel(), cos))
            worst = min(worst, (cos, tag + "." + name))
            assert cos >= (0.8 if w0.numel() >= 4096 else 0.65), (tag, name, cos)
    print("worst update cosine:", worst)


def test_graph_replay_equals_eager():
    """A train_op replayed from a captured HIP graph produces bit-identical parameters to eager launches."""
    from rna_gan_amd import graphed
    in_size, step, enc, n = 32, 64, 128, 8
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    results = []
    for use_graphs in (True, False):
        graphed.ENABLED = use_graphs
        G, D, og, od = product_pair(in_size, step, enc, "bf16", G0, D0)
        lg, ld, lp = PL.WassersteinGeneratorLoss(), PL.WassersteinDiscriminatorLoss(clip=(-0.01, 0.01)), \
            PL.WassersteinGradientPenalty()
        losses = []
        for it in range(5):                      # calls 1-2 eager, 3 captures + replays, 4-5 replay
            real = R.synthetic_images(n, in_size, seed=100 + it).cuda()
            nz = [R.synthetic_normal(n, enc, seed=200 + 3 * it + j).cuda() for j in range(3)]
            eps = torch.tensor([0.1 + 0.2 * it], device="cuda")
            losses += [lg.step(G, D, og, nz[0]).item(), ld.step(G, D, od, real, nz[1]).item(),
                       lp.step(G, D, od, real, nz[2], eps).item()]
        results.append((losses, G.flat.data.clone(), D.flat.data.clone(), og.state_dict(), od.state_dict(),
                        {k: v.clone() for k, v in D.state_dict().items()}))
    graphed.ENABLED = True
    (la, ga, da, oga, oda, sda), (lb, gb, db, ogb, odb, sdb) = results
    assert la == lb
    assert torch.equal(ga, gb) and torch.equal(da, db)
    assert float(oga["state"][0]["step"]) == float(ogb["state"][0]["step"]) == 5.0
    assert float(oda["state"][0]["step"]) == float(odb["state"][0]["step"]) == 10.0
    for k in sda:
        assert torch.equal(sda[k], sdb[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("size", [16, 32])
def test_up_generator_reference_fixture(size):
    """The reference's DCGANUpGenerator (fixture F4, generated by importing src/dcgan.py): output, every parameter
    gradient and the BatchNorm buffers of one forward/backward through the product module on the HIP kernels."""
    import numpy as np
    import torch.nn as nn
    from rna_gan_amd import DCGANUpGenerator
    from rna_gan_amd import engine as E
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "f4_upgen_tiny.npz"))
    G = DCGANUpGenerator(16, size, 3, 4, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    R.seeded_fill_(G, 41)
    G = G.cuda().train().set_precision("fp32")
    z = R.synthetic_normal(3, 16, seed=42).cuda()
    cot = R.synthetic_normal(3, 3 * size * size, seed=43).view(3, 3, size, size).cuda().contiguous()
    ops, net = G.runtime()
    y, ctx = E.upgen_forward(ops, net, z)
    E.upgen_backward(ops, net, ctx, cot, accumulate=False)
    np.testing.assert_allclose(y.cpu().numpy(), fx[f"y{size}"], rtol=2e-4, atol=2e-5)
    for k, p in G.named_parameters():
        ref = fx[f"grad{size}.{k}"]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * scale, err_msg=k)
    for k, b in G.named_buffers():
        np.testing.assert_allclose(b.cpu().numpy(), fx[f"buf{size}.{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
    # the module's own forward (train mode) gives the same image
    G2 = DCGANUpGenerator(16, size, 3, 4)
    R.seeded_fill_(G2, 41)
    G2 = G2.cuda().train().set_precision("fp32")
    np.testing.assert_allclose(G2(z).detach().cpu().numpy(), fx[f"y{size}"], rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
def test_autograd_wrappers_custom_loss():
    """A loss written against plain torch autograd (not one of the train_ops plugins): out = D(G(z)), D(real);
    least-squares objectives; loss.backward() -- the autograd.Function wrappers must leave the same .grad in every
    parameter as PyTorch autograd does on the oracle modules (fp32 kernels, tight tolerance)."""
    in_size, step, enc, n = 32, 4, 16, 6
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step), 81)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step), 82)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    G, D, og, od = product_pair(in_size, step, enc, "fp32", G0, D0)
    z = R.synthetic_normal(n, enc, seed=83)
    real = R.synthetic_images(n, in_size, seed=84)
    # generator objective through both networks
    loss_o = ((Do(Go(z)) - 1.0) ** 2).mean()
    loss_o.backward()
    og.zero_grad(); od.zero_grad()
    loss_p = ((D(G(z.cuda())) - 1.0) ** 2).mean()
    loss_p.backward()
    assert abs(float(loss_p) - float(loss_o)) <= 1e-4 * (abs(float(loss_o)) + 1)
    for (k, po), pp in zip(Go.named_parameters(), G.parameters()):
        ref = po.grad.numpy()
        np.testing.assert_allclose(pp.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(ref).max())),
                                   err_msg="G." + k)
    for (k, po), pp in zip(Do.named_parameters(), D.parameters()):
        ref = po.grad.numpy()
        np.testing.assert_allclose(pp.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(ref).max())),
                                   err_msg="D." + k)
    # discriminator objective on real + detached fake, gradients accumulate over two backward calls like autograd's
    for m in (Go, Do):
        m.zero_grad()
    og.zero_grad(); od.zero_grad()
    (Do(real) ** 2).mean().backward()
    ((Do(Go(z).detach()) + 1.0) ** 2).mean().backward()
    (D(real.cuda()) ** 2).mean().backward()
    ((D(G(z.cuda()).detach()) + 1.0) ** 2).mean().backward()
    for (k, po), pp in zip(Do.named_parameters(), D.parameters()):
        ref = po.grad.numpy()
        np.testing.assert_allclose(pp.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(ref).max())),
                                   err_msg="D(acc)." + k)


def test_public_functional_forms_fp32():
    """The callable forms the reference's loss classes expose besides train_ops (src/wgan_loss.py:24-44, :293-312) and
    the module modes it never uses but an nn.Module has: penalty.forward(interpolate, d_interpolate) through torch
    autograd on the product discriminator (second-order pass inside _GradientPenaltyFn) -- value, parameter gradients
    and d/d interpolate against torch's double backward on the oracle; eval-mode discriminator; d/dz of the generator."""
    in_size, step, enc, n = 32, 16, 32, 6
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    G, D, og, od = product_pair(in_size, step, enc, "fp32", G0, D0)
    x = R.synthetic_images(n, in_size, seed=3)
    # --- oracle: 10 * penalty + mean(D(x)) back-propagated to parameters and x
    xo = x.clone().requires_grad_(True)
    do = Do(xo)
    gpo = R.gradient_penalty(xo, do)
    (10.0 * gpo + do.mean()).backward()
    # --- product, through the public forms
    for pen in (PL.WassersteinGradientPenalty(),):
        D.flat.grad.zero_()                     # (builds the runtime: parameters / .grad views live in flat buffers)
        xp = x.cuda().requires_grad_(True)
        dp = D(xp)
        gp = pen(xp, dp)                       # nn.Module.__call__ -> forward(interpolate, d_interpolate)
        assert gp.dim() == 0 and abs(float(gp.detach()) - float(gpo.detach())) <= 2e-4 * (abs(float(gpo.detach())) + 1e-3)
        (10.0 * gp + dp.mean()).backward()
        assert l2rel(xp.grad.cpu().numpy(), xo.grad.numpy()) <= 2e-3
        for (k, po), pp in zip(Do.named_parameters(), D.parameters()):
            assert l2rel(pp.grad.detach().cpu().numpy(), po.grad.numpy()) <= 5e-3, k
    xp2, xq = x.cuda().requires_grad_(True), x.clone().requires_grad_(True)
    got_v = float(PL.wasserstein_gradient_penalty_vae(xp2, D(xp2), "sum").detach())      # reduction is ignored (:44)
    assert abs(got_v - float(R.gradient_penalty(xq, Do(xq)).detach())) <= 1e-3
    # --- eval-mode discriminator
    D.eval(); Do.eval()
    with torch.no_grad():
        want = Do(x)
        got = D(x.cuda())
    assert got.shape == want.shape and l2rel(got.cpu().numpy(), want.numpy()) <= 1e-4
    D.train(); Do.train()
    # --- generator input gradient
    zo = R.synthetic_normal(n, enc, seed=4).requires_grad_(True)
    Go(zo).square().sum().backward()
    zp = zo.detach().clone().cuda().requires_grad_(True)
    G(zp).square().sum().backward()
    assert l2rel(zp.grad.cpu().numpy(), zo.grad.numpy()) <= 2e-3


def test_betavae_encode_train_mode():
    """betaVAE.encode with the module in train mode (Dropout keep-mask injected, batch-statistics BatchNorm1d with a
    running-statistics update) against the oracle module's train-mode encoder."""
    F, Z, n = 96, 32, 12
    bo = R.seeded_fill_(R.OracleBetaVAE(F, Z, [64, 48, Z], [48, 64]), 5)
    bp = P.betaVAE(F, Z, [64, 48, Z], [48, 64])
    bp.load_state_dict(bo.state_dict())
    bp.set_precision("fp32")
    bp = bp.cuda().train()
    bo.train()
    x = R.synthetic_rna(n, F, seed=6, distinct=n)
    mask = (torch.rand(n, F, generator=torch.Generator().manual_seed(1)) > 0.5)
    bp.fixed_mask = mask.to(torch.uint8)
    zm, zl, h = bp.encode(x.cuda())
    # oracle: the encoder with the same keep-mask (Dropout p = 0.5 -> scale 2), BatchNorm in train mode
    hh = x * mask.float() * 2.0
    for blk in list(bo.encoder.encoder.children())[1:]:
        hh = blk(hh)
    np.testing.assert_allclose(h[:, :Z].cpu().numpy(), hh.detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(zm.cpu().numpy(), bo.z_mu(hh).detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(zl.cpu().numpy(), bo.z_logvar(hh).detach().numpy(), rtol=2e-4, atol=2e-5)
    sp, so = bp.state_dict(), bo.state_dict()
    for k in so:
        if "running" in k:
            np.testing.assert_allclose(sp[k].cpu().numpy(), so[k].numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_generator_eval_fused_epilogue():
    """Generator-only inference, eval-mode BatchNorm: the fused path (folded BatchNorm affine + LeakyReLU in
    the conv / GEMM epilogue, one kernel per block) against the unfused conv -> bn_act pairs and against the oracle
    generator in eval mode; then the BASELINE configs[4] batch (4096 samples in chunks of 256 at the reference size)
    with device-resident output, spot-checked against the unfused path."""
    from rna_gan_amd import engine as E
    in_size, step, enc, n = 64, 64, 128, 128
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7).eval()
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    G.load_state_dict(G0.state_dict())
    G = G.set_precision("bf16").cuda().eval()
    z = R.synthetic_normal(n, enc, seed=5)
    with torch.no_grad():
        want = G0(z)
    ops, net = G.runtime()
    fused = E.gen_forward_eval(ops, net, z.cuda())
    plain = E.gen_forward_eval(ops, net, z.cuda(), fused_epilogue=False)
    assert fused.shape == want.shape == (n, 3, in_size, in_size)
    assert float((fused.cpu() - want).abs().max()) <= 6e-2 and float((plain.cpu() - want).abs().max()) <= 6e-2
    assert float((fused - plain).abs().max()) <= 4e-2
    assert float((fused.cpu() - want).abs().mean()) <= 1.2 * float((plain.cpu() - want).abs().mean()) + 1e-4
    assert torch.equal(G(z.cuda()), fused)                     # nn.Module call in eval mode = the fused path
    # configs[4] shape: 4096 samples, reference model size, chunks of 256, output stays on the device
    Gf = P.DCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    R.seeded_fill_(Gf, 3)
    Gf = Gf.set_precision("bf16").cuda().eval()
    ops, net = Gf.runtime()
    noise = torch.randn(4096, 2048, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    outs = []
    with torch.no_grad():
        for c in torch.split(noise, 256):
            img = E.gen_forward_eval(ops, net, c.contiguous())
            assert img.is_cuda and img.shape == (256, 3, 256, 256)
            outs.append(img[:2].clone())
    got = torch.cat(outs)
    assert torch.isfinite(got).all() and float(got.abs().max()) <= 1.0
    ref = torch.cat([E.gen_forward_eval(ops, net, c[:64].contiguous(), fused_epilogue=False)[:2]
                     for c in torch.split(noise, 256)[:3]])
    # (batch-independent in eval mode: the first samples of a chunk do not depend on the chunk size)
    assert float((got[:6] - ref).abs().max()) <= 5e-2


def test_generator_inference_fp8():
    """BASELINE configs[4] (generator-only synthesis, fp8): eval-mode generator with fp8 (e4m3) weights / activations on
    the layers that have an fp8 kernel -- against the oracle generator (fp32, eval mode) at a covered size, and at the
    reference size for 4096 samples (chunks of 512, device-resident output) against the bf16 path on sampled images."""
    from rna_gan_amd import engine as E
    in_size, step, enc, n = 64, 64, 512, 256
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7).eval()
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    G.load_state_dict(G0.state_dict())
    G = G.set_precision("bf16").cuda().eval()
    z = R.synthetic_normal(n, enc, seed=5)
    with torch.no_grad():
        want = G0(z)
    ops, net = G.runtime()
    img8, nfp8 = E.gen_forward_eval_fp8(ops, net, z.cuda())
    img16 = E.gen_forward_eval(ops, net, z.cuda())
    assert nfp8 == 3, nfp8                                     # first layer + 512 -> 256 -> 128; the 128 -> 64 block and the image layer: bf16
    e8, e16 = (img8.cpu() - want).abs(), (img16.cpu() - want).abs()
    print("fp8 vs oracle: mean %.4f max %.4f ; bf16 vs oracle: mean %.4f max %.4f" % (e8.mean(), e8.max(), e16.mean(), e16.max()))
    assert float(e8.mean()) <= 2e-2 and float(e8.max()) <= 0.35 and float(e16.mean()) <= 4e-3
    G.set_inference_fp8(True)
    assert torch.equal(G(z.cuda()), img8)
    # reference size, 4096 samples
    Gf = P.DCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    R.seeded_fill_(Gf, 3)
    Gf = Gf.set_precision("bf16").cuda().eval()
    ops, net = Gf.runtime()
    noise = torch.randn(4096, 2048, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    d_mean, d_max = 0.0, 0.0
    with torch.no_grad():
        for k, c in enumerate(torch.split(noise, 512)):
            img, nfp8 = E.gen_forward_eval_fp8(ops, net, c.contiguous())
            assert nfp8 == 5 and img.is_cuda and img.shape == (512, 3, 256, 256) and torch.isfinite(img).all()
            if k < 2:
                ref = E.gen_forward_eval(ops, net, c[:128].contiguous())
                d = (img[:128] - ref).abs()
                d_mean, d_max = max(d_mean, float(d.mean())), max(d_max, float(d.max()))
    print("fp8 vs bf16 at the reference size: mean |diff| %.4f, max %.4f" % (d_mean, d_max))
    assert d_mean <= 3e-2
    # configs[4] at FULL size against the ORACLE: 32 samples through the oracle generator in eval
    # mode on the CPU (fp32, the reference's arithmetic, src/gan_utils.py:217-221,236-242 with running statistics) -- not
    # against this product's own bf16 path.  Non-trivial running statistics, so the folded affine is exercised.
    G0f = R.seeded_fill_(R.OracleDCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2),
                                                last_nonlinearity=nn.Tanh()), 3)
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for m in G0f.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(0.05 * torch.randn(m.num_features, generator=gen))
                m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=gen))
    G0f = G0f.eval()
    Gq = P.DCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    Gq.load_state_dict(G0f.state_dict())
    Gq = Gq.set_precision("bf16").cuda().eval()
    z256 = R.synthetic_normal(256, 2048, seed=9)              # the fp8 kernels take whole 256-row tiles: a chunk of 256 samples,
    z32 = z256[:32]                                            # of which the first 32 go through the oracle (eval mode: per sample)
    with torch.no_grad():
        want = G0f(z32)
        ops, net = Gq.runtime()
        img8, nfp8 = E.gen_forward_eval_fp8(ops, net, z256.cuda())
        img16 = E.gen_forward_eval(ops, net, z256.cuda())
    assert nfp8 == 5
    e8, e16 = (img8[:32].cpu() - want).abs(), (img16[:32].cpu() - want).abs()
    print("reference size, 32 samples vs the ORACLE (fp32 CPU, eval mode): fp8 mean |diff| %.4f max %.4f ; bf16 mean %.4f max %.4f"
          % (e8.mean(), e8.max(), e16.mean(), e16.max()))
    # stated tolerance for images in [-1, 1]: fp8 e4m3 operands (3 mantissa bits, per-output-channel weight scales) through
    # 5 of the 7 layers: mean |diff| <= 3e-2, max <= 0.5; the bf16 path on the same inputs: mean <= 6e-3
    assert float(e8.mean()) <= 3e-2 and float(e8.max()) <= 0.5, (float(e8.mean()), float(e8.max()))
    assert float(e16.mean()) <= 6e-3, float(e16.mean())


def test_generator_forward_without_kept_context_fuses_last_batchnorm():
    """gen_forward(keep=False) applies the last BatchNorm + LeakyReLU inside the image layer (rg_last_up_pre): the images and
    the running statistics are bit-identical to the forward that materialises the normalised activation (keep=True)."""
    import torch.nn as nn
    import rna_gan_amd as P
    from rna_gan_amd import engine as E
    from oracle import ref_cpu as R
    for in_size, n in ((128, 4), (256, 8)):
        G0 = R.seeded_fill_(R.OracleDCGANGenerator(128, in_size, 3, 64, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.Tanh()), 7)
        out = []
        for keep in (True, False):
            G = P.DCGANGenerator(128, in_size, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
            G.load_state_dict(G0.state_dict())
            G = G.cuda().train()
            ops, gn = G.runtime()
            nz = R.synthetic_normal(n, 128, seed=200).cuda()
            img, _ = E.gen_forward(ops, gn, nz, keep=keep)
            torch.cuda.synchronize()
            out.append((img.clone(), {k: b.clone() for k, b in G.named_buffers()}))
        assert torch.equal(out[0][0], out[1][0])
        for k in out[0][1]:
            assert torch.equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize("in_size,n", [(64, 8), (256, 64)])
def test_generator_forward_pair_on_gpu(in_size, n):
    """engine.gen_forward_pair on the HIP path against two gen_forward calls: images within bf16 noise of a different tile
    shape, running statistics equal (first half first)."""
    import torch.nn as nn
    import rna_gan_amd as P
    from rna_gan_amd import engine as E
    from oracle import ref_cpu as R
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(128, in_size, 3, 64, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    za, zb = R.synthetic_normal(n, 128, seed=200).cuda(), R.synthetic_normal(n, 128, seed=201).cuda()
    res = []
    for pair in (False, True):
        G = P.DCGANGenerator(128, in_size, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
        G.load_state_dict(G0.state_dict())
        G = G.cuda().train()
        ops, gn = G.runtime()
        if pair:
            img = E.gen_forward_pair(ops, gn, torch.cat([za, zb]))
        else:
            img = torch.cat([E.gen_forward(ops, gn, za, keep=False)[0], E.gen_forward(ops, gn, zb, keep=False)[0]])
        torch.cuda.synchronize()
        res.append((img.clone(), {k: b.double().clone() for k, b in G.named_buffers()}))
    a, b = res[0][0], res[1][0]
    assert float((a - b).abs().mean()) <= 5e-3 and float((a - b).abs().max()) <= 0.25, (float((a - b).abs().mean()),
                                                                                          float((a - b).abs().max()))
    for k in res[0][1]:
        assert torch.allclose(res[0][1][k], res[1][1][k], rtol=2e-3, atol=1e-4), k


@pytest.mark.parametrize("in_size,enc,n", [(32, 128, 16), (64, 512, 64), (32, 128, 80)])
def test_generator_layer0_gradient_inside_the_adam_step(in_size, enc, n):
    """G.0's weight gradient formed inside the fused optimizer step (rg_g0_wgrad_adam: gradient in registers + Adam in one
    streaming pass, no dw round trip) against the two-kernel form (rg_g0_wgrad, rg_adam_step_dev) for three consecutive
    generator-loss train_ops: same losses, same parameters / Adam moments to fp32 round-off of a differently ordered 64-term
    sum, bf16 shadow consistent with the master.  n = 80: a ragged second 64-sample chunk."""
    step = 64
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    res = []
    keep = PL.G0_ADAM
    try:
        for fused in (True, False):
            PL.G0_ADAM = fused
            G, D, og, od = product_pair(in_size, step, enc, "bf16", G0, D0)
            lg = PL.WassersteinGeneratorLoss()
            losses = []
            for it in range(3):
                losses.append(lg.step(G, D, og, R.synthetic_normal(n, enc, seed=50 + it).cuda()).item())
            torch.cuda.synchronize()
            _, net = G.runtime()
            assert net.g0.pending_wgrad is None and not net.g0.fuse_step
            st = og.state_dict()["state"][0]
            res.append((losses, G.model[0][0].weight.detach().float().cpu().clone(), st["exp_avg"].float().cpu().clone(),
                        st["exp_avg_sq"].float().cpu().clone(), G.flat.shadow[:G.model[0][0].weight.numel()].float().cpu().clone(),
                        {k: v.detach().float().cpu().clone() for k, v in G.state_dict().items()}))
    finally:
        PL.G0_ADAM = keep
    (la, wa, ma, va, sa, sda), (lb, wb, mb, vb, sb, sdb) = res
    w0 = G0.model[0][0].weight.detach()
    assert max(abs(x - y) / (abs(y) + 0.1) for x, y in zip(la, lb)) < 2e-3, (la, lb)
    du_a, du_b = wa - w0, wb - w0
    cos = float((du_a * du_b).sum() / (du_a.norm() * du_b.norm()))
    assert cos > 0.999, cos                                   # three sign-like Adam steps: the updates coincide
    assert float((ma - mb).norm() / mb.norm()) < 2e-3 and float((va - vb).norm() / vb.norm()) < 4e-3
    assert torch.equal(sa.reshape(-1), wa.bfloat16().float().reshape(-1))      # the bf16 shadow IS the rounded master
    for k in sda:                                             # everything else of the generator went through the usual kernels
        if k != "model.0.0.weight" and sda[k].dtype.is_floating_point:
            assert float((sda[k] - sdb[k]).abs().max()) <= 5e-3 * float(sdb[k].abs().max() + 1e-6) + 2e-4, k


def test_unselected_inputs_loss_agreement_statistics():
    """smoke() and two tests in tests/test_engine_gpu.py CHOOSE their inputs (on the CPU oracle) for a
    clear LeakyReLU margin at the critic head, so the suite never said how often arbitrary inputs fall outside the stated
    tolerance.  This test takes 24 CONSECUTIVE seeds with no selection -- weights, tiles and draws all change with the seed --
    runs one iteration (G-loss, D-loss, penalty train_ops) on the HIP path in both precisions and on the CPU oracle, and
    reports the distribution of |hip - oracle| / (|oracle| + 0.1) per loss.  Asserted: fp32 kernels within 2e-3 on every seed
    (the reference's arithmetic: no tolerance for a flipped slope is needed at these sizes); bf16 kernels: median <= 2e-2 and at
    least 75 % of the 72 loss values within the stated 6e-2 -- the rest are the ill-conditioned inputs DESIGN 12.9 describes
    (a head pre-activation within bf16 noise of the kink), printed so that their frequency is on record."""
    in_size, step, enc, n = 32, 64, 128, 8
    seeds = list(range(101, 125))
    errs = {"fp32": [], "bf16": []}
    for seed in seeds:
        G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.Tanh()), seed)
        D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                       last_nonlinearity=nn.LeakyReLU(0.2)), seed + 1000)
        Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
        ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
        real = R.synthetic_images(n, in_size, seed=seed)
        noises = [R.synthetic_normal(n, enc, seed=100 * seed + 2 + j) for j in range(3)]
        ref = R.train_iteration(Go, Do, ogo, odo, real, noises, 0.4)
        for precision in ("fp32", "bf16"):
            G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
            D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
            G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
            G.set_precision(precision); D.set_precision(precision)
            G, D = G.cuda().train(), D.cuda().train()
            og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
            od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
            rd = real.cuda()
            got = {"g": PL._g_step(G, D, og, noises[0].cuda()).item(),
                   "d": PL._d_step(G, D, od, rd, noises[1].cuda(), None).item(),
                   "gp": PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.4, 10.0).item()}
            errs[precision].append([abs(got[k] - ref[k]) / (abs(ref[k]) + 0.1) for k in ("g", "d", "gp")])
    for precision in ("fp32", "bf16"):
        e = np.asarray(errs[precision])
        flat = np.sort(e.reshape(-1))
        print("%s: %d seeds x 3 losses: median %.2e, 90th percentile %.2e, max %.2e; per loss max (g, d, gp) %s; "
              "values beyond 6e-2: %d of %d (seeds %s)" % (
                  precision, len(seeds), np.median(flat), flat[int(0.9 * len(flat))], flat[-1], np.round(e.max(0), 4).tolist(),
                  int((flat > 6e-2).sum()), flat.size, [seeds[i] for i in np.where((e > 6e-2).any(1))[0]]))
    f32, b16 = np.asarray(errs["fp32"]), np.asarray(errs["bf16"])
    assert np.isfinite(f32).all() and np.isfinite(b16).all()
    assert float(f32.max()) <= 2e-3, f32.max()
    assert float(np.median(b16)) <= 2e-2, np.median(b16)
    assert float((b16 <= 6e-2).mean()) >= 0.75, (b16 <= 6e-2).mean()


def _loss_and_update_statistics(in_size, seeds, n=8, step=64, enc=128, precisions=("fp32", "bf16")):
    """One iteration (G-loss, D-loss, penalty train_ops) per seed -- weights, tiles and draws all change with the seed, nothing
    is selected -- on the HIP path in both precisions and on the CPU oracle.  Returns per precision: errs [seed][3] =
    |hip - oracle| / (|oracle| + 0.1) of the three losses, cos [seed][2] = cosine between the product's and the oracle's
    parameter UPDATES (all parameters of G / of D concatenated: the gate smoke() and tests/test_bench_step_gpu.py use)."""
    errs = {p: [] for p in precisions}
    coss = {p: [] for p in precisions}
    for seed in seeds:
        G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.Tanh()), seed)
        D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                       last_nonlinearity=nn.LeakyReLU(0.2)), seed + 1000)
        Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
        ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
        real = R.synthetic_images(n, in_size, seed=seed)
        noises = [R.synthetic_normal(n, enc, seed=100 * seed + 2 + j) for j in range(3)]
        ref = R.train_iteration(Go, Do, ogo, odo, real, noises, 0.4)
        upd_ref = [torch.cat([(a.detach() - b.detach()).reshape(-1).double() for a, b in zip(m1.parameters(), m0.parameters())])
                   for m1, m0 in ((Go, G0), (Do, D0))]
        for precision in precisions:
            G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
            D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
            G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
            G.set_precision(precision); D.set_precision(precision)
            G, D = G.cuda().train(), D.cuda().train()
            og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
            od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
            rd = real.cuda()
            got = {"g": PL._g_step(G, D, og, noises[0].cuda()).item(),
                   "d": PL._d_step(G, D, od, rd, noises[1].cuda(), None).item(),
                   "gp": PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.4, 10.0).item()}
            errs[precision].append([abs(got[k] - ref[k]) / (abs(ref[k]) + 0.1) for k in ("g", "d", "gp")])
            cs = []
            for mod, m0, ur in ((G, G0, upd_ref[0]), (D, D0, upd_ref[1])):
                sd0 = dict(m0.named_parameters())
                u = torch.cat([(p.detach().cpu().double() - sd0[k].detach().double()).reshape(-1)
                               for k, p in mod.named_parameters()])
                cs.append(float((u * ur).sum() / (u.norm() * ur.norm() + 1e-300)))
            coss[precision].append(cs)
    return {k: np.asarray(v) for k, v in errs.items()}, {k: np.asarray(v) for k, v in coss.items()}


def _describe(tag, e, c, seeds):
    flat = np.sort(e.reshape(-1))
    q = lambda f: flat[min(int(f * len(flat)), len(flat) - 1)]
    line = ("%s: %d seeds x 3 losses: median %.2e, 90th %.2e, 99th %.2e, max %.2e; per loss max (g, d, gp) %s; beyond 6e-2: %d of %d "
            "(seeds %s); update cosine G: min %.3f 1st-pct %.3f median %.3f, D: min %.3f 1st-pct %.3f median %.3f"
            % (tag, len(seeds), np.median(flat), q(0.9), q(0.99), flat[-1], np.round(e.max(0), 4).tolist(),
               int((flat > 6e-2).sum()), flat.size, [seeds[i] for i in np.where((e > 6e-2).any(1))[0]],
               c[:, 0].min(), np.sort(c[:, 0])[max(int(0.01 * len(c)), 0)], np.median(c[:, 0]),
               c[:, 1].min(), np.sort(c[:, 1])[max(int(0.01 * len(c)), 0)], np.median(c[:, 1])))
    print(line)
    return line


def test_unselected_inputs_statistics_at_64():
    """the tolerances as DISTRIBUTIONS on unselected inputs at a second size -- 64 x 64, 40 consecutive
    seeds in the suite; tools/tolerance_stats.py runs 200 seeds at 32 x 32 and 64 x 64 (profiles/round5_tolerance_statistics.txt,
    DESIGN 14.6) -- for the loss values AND for the update-cosine gate.  What the 200-seed runs say, and what is asserted here
    with headroom:
      fp32 kernels: every loss within 1e-2 (99th percentile 3.3e-3: a flipped LeakyReLU slope at batch 8 is the tail), update
        cosine >= 0.988 on every seed;
      bf16 kernels: median 5.7e-3, 90th percentile 3.6e-2; the stated 6e-2 holds on 95.3 % of the values at 64 x 64 (98 % at
        32 x 32) -- it is a ~95th-percentile bound on ARBITRARY inputs and an every-input bound only where the critic head keeps
        a LeakyReLU margin (how smoke() and the engine tests choose theirs); the tail is the penalty (||dD/dx|| - 1)^2 of 8
        samples, whose value moves by up to 10 x when ONE sample's head slope flips; update cosine median 0.90 (G) / 0.95 (D),
        minimum 0.65 -- against 0.92 / 0.97 and 0.67 at 32 x 32."""
    seeds = list(range(301, 341))
    errs, coss = _loss_and_update_statistics(64, seeds)
    for precision in ("fp32", "bf16"):
        _describe(precision + " 64x64", errs[precision], coss[precision], seeds)
    f32, b16 = errs["fp32"], errs["bf16"]
    assert np.isfinite(f32).all() and np.isfinite(b16).all()
    assert float(f32.max()) <= 1.5e-2, f32.max()
    assert float(np.sort(f32.reshape(-1))[int(0.9 * f32.size)]) <= 2.5e-3
    assert float(coss["fp32"].min()) >= 0.98, coss["fp32"].min()
    assert float(np.median(b16)) <= 1.5e-2, np.median(b16)
    assert float((b16 <= 6e-2).mean()) >= 0.90, (b16 <= 6e-2).mean()
    assert float(coss["bf16"].min()) >= 0.50 and float(np.median(coss["bf16"])) >= 0.85, (coss["bf16"].min(), np.median(coss["bf16"]))
