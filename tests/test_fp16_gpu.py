"""BASELINE.json configs[3] names fp16 storage.  precision="fp16" = the SAME kernels built with IEEE fp16 as the 16-bit storage
type (librnagan_hip_f16.so, v_mfma_f32_*_f16) + a static loss scale on the backward seeds that rna_gan_amd.optim.Adam removes
inside its kernels (the mechanism the reference's authors sketched and commented out: src/betaVAE.py:184,230-236).

  * the scaling plumbing is EXACT: with power-of-two scales forced onto the bf16 / fp32 paths (whose exponent range makes a
    power-of-two scaling exact) every parameter comes out bit-identical to the unscaled run -- every gradient path of the three
    train_ops carries the scale exactly once and every Adam kernel removes it;
  * the fp16 build's MFMA lane maps (device self-test with exact integers) and a conv layer against torch;
  * two iterations against the CPU oracle (the test of tests/test_train_gpu.py, fp16 column);
  * the reference model size, batch 8, against the oracle.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
import rna_gan_amd as P
from rna_gan_amd import losses as PL
from test_train_gpu import product_pair, l2rel


def _models(in_size, step, enc):
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 7)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 8)
    return G0, D0


def _run_iterations(G, D, og, od, in_size, enc, n, iters):
    out = []
    for it in range(iters):
        real = R.synthetic_images(n, in_size, seed=100 + it).cuda()
        noises = [R.synthetic_normal(n, enc, seed=200 + 3 * it + j).cuda() for j in range(3)]
        out.append((PL._g_step(G, D, og, noises[0]).item(), PL._d_step(G, D, od, real, noises[1], None).item(),
                    PL._gp_step(G, D, od, real, noises[2], 0.25 + 0.5 * it, 10.0).item()))
    return out


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_loss_scale_plumbing_is_exact(precision):
    """Scales 4096 = 64 * 64 forced onto a path where power-of-two scaling is exact: losses and ALL parameters, BatchNorm buffers
    and Adam moments after two iterations are bit-identical to the unscaled run."""
    in_size, step, enc, n = 32, 64, 128, 16
    G0, D0 = _models(in_size, step, enc)
    res = {}
    for scaled in (False, True):
        G, D, og, od = product_pair(in_size, step, enc, precision, G0, D0)
        for m in (G, D):
            ops, _ = m.runtime()
            if scaled:
                ops.loss_scale, ops.gp_seed_scale, ops.gp_tangent_scale = 4096.0, 64.0, 64.0
        losses = _run_iterations(G, D, og, od, in_size, enc, n, 2)
        res[scaled] = (losses, [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())],
                       [b.detach().clone() for b in list(G.buffers()) + list(D.buffers())],
                       [og._m.clone(), og._v.clone(), od._m.clone(), od._v.clone()])
    assert res[False][0] == res[True][0]
    for k in (1, 2, 3):
        for a, b in zip(res[False][k], res[True][k]):
            assert torch.equal(a, b)


def test_fp16_build_lane_maps_and_a_conv_layer():
    from rna_gan_amd.ops_hip import HipOps
    from rna_gan_amd.engine import ConvW
    ops = HipOps(torch.float16, "cuda:0")
    assert ops.selftest() == [0, 0]              # MFMA operand / accumulator maps and the transposed LDS read, exact integers
    g = torch.Generator().manual_seed(3)
    N, H, I, O = 8, 32, 128, 256
    x = torch.randn(N, H, H, I, generator=g).cuda().half()
    w = (torch.randn(O, I, 4, 4, generator=g) * 0.03).cuda()
    wt = w.permute(0, 2, 3, 1).contiguous()
    cw = ConvW(wt, None, torch.zeros_like(wt), None, "OHWI")
    y = ops.conv_down(x, cw)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.half().float(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert y.dtype == torch.float16
    assert float((y.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())     # one fp16 rounding of the result
    gy = torch.randn(N, H // 2, H // 2, O, generator=g).cuda().half()
    gx = ops.conv_up(gy, cw)
    refx = torch.nn.functional.conv_transpose2d(gy.float().permute(0, 3, 1, 2), w.half().float(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert float((gx.float() - refx).abs().max()) <= 2e-3 * float(refx.abs().max())
    ops.conv_wgrad(gy, x, cw, False)
    refw = torch.einsum("nhwo,nhwkli->okli", gy.float(),
                        torch.nn.functional.pad(x.float(), (0, 0, 1, 1, 1, 1)).unfold(1, 4, 2).unfold(2, 4, 2).permute(0, 1, 2, 4, 5, 3))
    assert l2rel(cw.dw.cpu().numpy(), refw.cpu().numpy()) <= 2e-3


def test_two_iterations_vs_oracle_fp16():
    """tests/test_train_gpu.py::test_two_iterations_vs_oracle in fp16: losses, first-iteration update directions, BatchNorm
    buffers.  Tolerances: the bf16 column's (4e-2 / cosine 0.75 / 3e-2); what fp16 measures is printed."""
    in_size, step, enc, n = 32, 64, 128, 16
    G0, D0 = _models(in_size, step, enc)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    G, D, og, od = product_pair(in_size, step, enc, "fp16", G0, D0)
    worst_loss, worst_cos = 0.0, 1.0
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
            err = abs(got - want) / (abs(want) + 0.5)
            worst_loss = max(worst_loss, err)
            assert np.isfinite(got) and err <= 4e-2, (it, nm, got, want)
        if it == 0:
            for mod, ref_mod, src in ((G, Go, G0), (D, Do, D0)):
                for (k, p), (_, q), (_, s) in zip(mod.named_parameters(), ref_mod.named_parameters(), src.named_parameters()):
                    du, dr = p.detach().cpu() - s.detach(), q.detach() - s.detach()
                    cos = float((du * dr).sum() / (du.norm() * dr.norm() + 1e-30))
                    worst_cos = min(worst_cos, cos)
                    assert cos >= 0.75, (k, cos)
    for mod, ref_mod in ((G, Go), (D, Do)):
        for (k, b), (_, q) in zip(mod.named_buffers(), ref_mod.named_buffers()):
            if k.endswith("num_batches_tracked"):
                assert int(b) == int(q), k
            else:
                assert l2rel(b.cpu().numpy(), q.numpy()) <= 3e-2, k
    print("fp16 two iterations: worst loss error %.2e, worst per-tensor update cosine %.4f" % (worst_loss, worst_cos))


def test_full_size_one_iteration_fp16():
    """Reference model size (enc 2048, step 64, 256 x 256), batch 8, fp16 storage: finite, losses near the fp32 CPU oracle, the
    update direction of every network against the oracle's."""
    in_size, step, enc, n = 256, 64, 2048, 8
    G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                               last_nonlinearity=nn.Tanh()), 17)
    D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.LeakyReLU(0.2)), 18)
    Go, Do = copy.deepcopy(G0).train(), copy.deepcopy(D0).train()
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    G, D, og, od = product_pair(in_size, step, enc, "fp16", G0, D0)
    real = R.synthetic_images(n, in_size, seed=300)
    noises = [R.synthetic_normal(n, enc, seed=400 + j) for j in range(3)]
    ref = R.train_iteration(Go, Do, ogo, odo, real, noises, 0.4)
    rd = real.cuda()
    lg = PL._g_step(G, D, og, noises[0].cuda()).item()
    ld = PL._d_step(G, D, od, rd, noises[1].cuda(), None).item()
    lp = PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.4, 10.0).item()
    print("full-size fp16 losses hip/ref:", lg, ref["g"], ld, ref["d"], lp, ref["gp"])
    for got, want in ((lg, ref["g"]), (ld, ref["d"]), (lp, ref["gp"])):
        assert np.isfinite(got) and abs(got - want) <= 6e-2 * (abs(want) + 0.1)
    for mod, ref_mod, src, nm in ((G, Go, G0, "G"), (D, Do, D0, "D")):
        du = torch.cat([(p.detach().cpu() - s.detach()).reshape(-1) for p, s in zip(mod.parameters(), src.parameters())])
        dr = torch.cat([(q.detach() - s.detach()).reshape(-1) for q, s in zip(ref_mod.parameters(), src.parameters())])
        cos = float((du * dr).sum() / (du.norm() * dr.norm() + 1e-30))
        print("full-size fp16 update cosine", nm, round(cos, 4))
        assert torch.isfinite(du).all() and cos >= 0.6, (nm, cos)
    img = G(noises[0].cuda())
    assert img.shape == (n, 3, 256, 256) and torch.isfinite(img).all() and float(img.abs().max()) <= 1.0


def test_fp16_stays_finite_and_tracks_fp32_over_150_iterations():
    """The static loss scale over a longer run on the same data and draws as an fp32 run (64 x 64, batch 16, 150 iterations, graphs
    replayed): every loss finite, parameters finite at the end, and the windowed loss means of the two runs agree (the
    trajectories diverge step by step as any two arithmetics do: the window is the statement)."""
    in_size, step, enc, n, iters = 64, 64, 128, 16, 150
    G0, D0 = _models(in_size, step, enc)
    curves = {}
    for precision in ("fp32", "fp16"):
        G, D, og, od = product_pair(in_size, step, enc, precision, G0, D0)
        out = []
        for it in range(iters):
            real = R.synthetic_images(n, in_size, seed=100 + (it % 8)).cuda()
            noises = [R.synthetic_normal(n, enc, seed=200 + 3 * it + j).cuda() for j in range(3)]
            out.append((PL._g_step(G, D, og, noises[0]).item(), PL._d_step(G, D, od, real, noises[1], None).item(),
                        PL._gp_step(G, D, od, real, noises[2], 0.5, 10.0).item()))
        curves[precision] = np.array(out)
        assert np.isfinite(curves[precision]).all(), precision
        for m in (G, D):
            assert all(torch.isfinite(v).all() for v in m.state_dict().values() if v.dtype.is_floating_point), precision
    a, b = curves["fp32"], curves["fp16"]
    for lo in range(0, iters, 50):
        wa, wb = a[lo:lo + 50].mean(0), b[lo:lo + 50].mean(0)
        print("iterations %d-%d: fp32 window means %s, fp16 %s" % (lo, lo + 50, np.round(wa, 4), np.round(wb, 4)))
        # D-loss and penalty windows within 15 %.  The G loss is the critic's raw output on fakes: a WGAN critic is defined up to
        # an additive constant that no loss term pins (the D loss is a difference, the penalty a gradient), so it drifts apart
        # between any two arithmetics (0.2 - 0.3 seen) and only a loose bound is asked of it
        assert np.all(np.abs(wa[1:] - wb[1:]) <= 0.15 * (np.abs(wa[1:]) + 0.25)), (lo, wa, wb)
        assert abs(wa[0] - wb[0]) <= 0.6, (lo, wa, wb)
