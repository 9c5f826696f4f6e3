"""GPU parity of the explicit training-step gradient computation (rna_gan_amd.engine on HipOps)
against the autograd oracle evaluated in float64 on the CPU (oracle/ref_cpu.py).

Tolerance policy (measured, see DESIGN.md "Numerics"):
  * LeakyReLU's derivative is discontinuous at 0, so ANY two fp32 evaluations (PyTorch CPU vs itself
    in another summation order, too) disagree on the mask of pre-activations closer to 0 than their
    rounding noise; one flipped element changes a whole BatchNorm channel's backward.
      - tiny fp32 cases: seeds are searched until no pre-activation of the fp64 oracle lies within
        MARGIN of 0  -> max-norm tolerance 5e-4 on every gradient;
      - mid fp32 cases (flips unavoidable: ~1e6 pre-activations): continuous quantities (losses,
        BN running statistics) to 2e-4, gradients to 2e-2 in relative L2;
  * bf16 path: operands of every MFMA are bf16; a bf16-rounding emulation of the same algorithm on
    the CPU (oracle/ops_ref.py with bfloat16) deviates 5-20 % (relative L2) from fp64 at these batch
    sizes, so gradients must agree to 0.35 relative L2 and cosine >= 0.93, losses to 5 %.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from rna_gan_amd import engine as E

MARGIN = 3e-5


def mk(in_size, step, enc, seed):
    G = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                              last_nonlinearity=nn.Tanh()), seed)
    D = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2),
                                                  last_nonlinearity=nn.LeakyReLU(0.2)), seed + 1)
    return G, D


def oracle64(G, D, real, noise, eps):
    """fp64 autograd results of the three steps + the smallest |LeakyReLU input| seen."""
    g, d = copy.deepcopy(G).double().train(), copy.deepcopy(D).double().train()
    r, z = real.double(), noise.double()
    margin = [float("inf")]

    def hook(_m, inp):
        margin[0] = min(margin[0], float(inp[0].detach().abs().min()))
    hs = [m.register_forward_pre_hook(hook) for mod in (g, d) for m in mod.modules() if isinstance(m, nn.LeakyReLU)]
    out = {}
    l = R.generator_loss(d(g(z))); l.backward()
    out["gl"] = float(l.detach()); out["G"] = {k: p.grad.clone() for k, p in g.named_parameters()}
    for p in d.parameters():
        p.grad = None
    l = R.discriminator_loss(d(r), d(g(z).detach())); l.backward()
    out["dl"] = float(l.detach()); out["D"] = {k: p.grad.clone() for k, p in d.named_parameters()}
    for p in d.parameters():
        p.grad = None
    xhat = eps * r + (1 - eps) * g(z)
    gp = R.gradient_penalty(xhat, d(xhat)); (10.0 * gp).backward()
    out["gp"] = float(gp.detach()); out["P"] = {k: p.grad.clone() for k, p in d.named_parameters()}
    out["bufG"] = {k: b.clone() for k, b in g.named_buffers()}
    out["bufD"] = {k: b.clone() for k, b in d.named_buffers()}
    for h in hs:
        h.remove()
    return out, margin[0]


def err(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    assert torch.isfinite(a).all(), "non-finite"
    mx = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    return mx, l2, cos


def check_grads(mod, ref, mode, what):
    for k, p in mod.named_parameters():
        if float(ref[k].abs().max()) == 0.0:      # e.g. d(beta) of the last BN layer in the GP step: exactly 0
            assert float(p.grad.abs().max()) <= 1e-6, f"{what} {k}: expected exact zero"
            continue
        mx, l2, cos = err(p.grad, ref[k])
        if mode == "tight":
            assert mx <= 5e-4, f"{what} {k}: max-rel {mx:.2e}"
        elif mode == "l2":
            assert l2 <= 2e-2, f"{what} {k}: l2-rel {l2:.2e}"
        else:
            assert l2 <= 0.35 and cos >= 0.93, f"{what} {k}: l2-rel {l2:.2e} cos {cos:.4f}"


def check_bufs(mod, ref, tol, what):
    for k, b in mod.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b.cpu()) == int(ref[k]), f"{what} {k}"
        else:
            assert err(b, ref[k])[0] <= tol, f"{what} {k}: {err(b, ref[k])[0]:.2e}"


CASES = [
    # in_size, step, enc, batch, dtype, mode
    (16, 4, 24, 5, torch.float32, "tight"),
    (32, 4, 16, 3, torch.float32, "tight"),
    (32, 64, 128, 4, torch.float32, "l2"),
    (32, 64, 128, 16, torch.bfloat16, "bf16"),     # MFMA kernels on the conv stack
    (64, 64, 128, 8, torch.bfloat16, "bf16"),
    (256, 64, 2048, 4, torch.float32, "l2"),       # the reference model's full size (fp32 parity mode), batch 4
]


@pytest.mark.parametrize("in_size,step,enc,n,dtype,mode,sync", [c + (False,) for c in CASES] +
                         [(32, 4, 16, 3, torch.float32, "tight", True), (32, 64, 128, 16, torch.bfloat16, "bf16", True)])
def test_three_steps_vs_autograd(in_size, step, enc, n, dtype, mode, sync):
    from rna_gan_amd.ops_hip import HipOps
    ops = HipOps(dtype, "cuda:0")
    if sync:
        # the split (local sums -> all-reduce -> apply) kernels of --sync-stats with a one-rank "all-reduce": must give
        # the single-process result like the fused kernels do
        ops.stat_reduce = lambda t: t
        ops.stat_world = 1
    eps = 0.3
    for seed in range(5, 45):
        G, D = mk(in_size, step, enc, seed)
        real = R.synthetic_images(n, in_size, seed=3 * seed)
        noise = R.synthetic_normal(n, enc, seed=3 * seed + 1)
        ref, margin = oracle64(G, D, real, noise, eps)
        if mode != "tight" or margin > MARGIN:
            break
    else:
        pytest.skip("no seed with the required LeakyReLU margin")
    Gg, Dg = copy.deepcopy(G).cuda().train(), copy.deepcopy(D).cuda().train()
    E.tap_major_(Gg), E.tap_major_(Dg)        # the HIP conv kernels take tap-major masters
    Gn, Dn = E.build_gen_net(Gg), E.build_disc_net(Dg)
    real_d, noise_d = real.cuda(), noise.cuda()
    ltol = 2e-4 if dtype == torch.float32 else 5e-2
    btol = 2e-4 if dtype == torch.float32 else 3e-2

    loss = float(E.gen_loss_grads(ops, Gn, Dn, noise_d).cpu())
    assert abs(loss - ref["gl"]) <= ltol * (abs(ref["gl"]) + 0.05), ("G loss", loss, ref["gl"])
    check_grads(Gg, ref["G"], mode, "G step")

    loss = float(E.disc_loss_grads(ops, Gn, Dn, real_d, noise_d).cpu())
    assert abs(loss - ref["dl"]) <= ltol * (abs(ref["dl"]) + 0.05), ("D loss", loss, ref["dl"])
    check_grads(Dg, ref["D"], mode, "D step")

    loss = float(E.gp_loss_grads(ops, Gn, Dn, real_d, noise_d, eps, 10.0).cpu())
    assert abs(loss - ref["gp"]) <= 10 * ltol * (abs(ref["gp"]) + 0.05), ("GP value", loss, ref["gp"])
    check_grads(Dg, ref["P"], mode, "GP step")
    check_bufs(Gg, ref["bufG"], btol, "G buffers")
    check_bufs(Dg, ref["bufD"], btol, "D buffers")


@pytest.mark.parametrize("in_size,step,enc,n", [(32, 64, 128, 16), (64, 64, 128, 8)])
def test_bf16_kernels_vs_bf16_rounding_twin(in_size, step, enc, n):
    """Is the bf16 (MFMA) path's distance from the fp64 oracle EXPLAINED by bf16 rounding?  Three evaluations of one
    iteration's gradients on the same inputs: fp64 autograd (oracle), the bf16-rounding twin (the same engine sequencing
    on the CPU over oracle/ops_ref.RefOps(bfloat16): bf16 operands / activations, fp32 accumulation) and the HIP
    kernels.  WGAN gradients have a large common-mode part that every BatchNorm backward subtracts, which amplifies the
    2^-9 roundings of the stored gradients to several per cent, differently for any two bf16 evaluations (the twin
    itself is 5-20 % from fp64); so, per parameter tensor:
        err(kernels, fp64) <= 1.5 x err(twin, fp64) + 1e-2   and   err(kernels, twin) <= 1.6 x err(twin, fp64) + 1e-2   (relative L2)
    -- a systematic error in one benchmarked kernel (a wrong tap, a dropped k-tile, a 10 % scale) breaks both.
    Losses within 3 % of the twin, BatchNorm buffers 5e-3."""
    from oracle.ops_ref import RefOps
    from rna_gan_amd.ops_hip import HipOps
    eps = 0.3
    # The critic ends in LeakyReLU(h_n): a sample whose head pre-activation h_n sits within bf16 noise of 0 flips its
    # whole backward contribution by a factor 5 in one evaluation and not in the other (seen: 38 % of the gradient norm
    # from ONE of 16 samples).  That is a property of the function, not of a kernel, so seeds are searched until every
    # h_n of the fp64 oracle (the discriminator calls of the iteration) is at least 8 % of the mean |h|.
    for seed in range(11, 160):
        G, D = mk(in_size, step, enc, seed)
        real = R.synthetic_images(n, in_size, seed=3 * seed)
        noise = R.synthetic_normal(n, enc, seed=3 * seed + 1)
        hs = []
        d64, g64 = copy.deepcopy(D).double().train(), copy.deepcopy(G).double().train()
        hk = d64.disc[0].register_forward_hook(lambda _m, _i, o: hs.append(o.detach().reshape(-1)))
        with torch.no_grad():
            fake = g64(noise.double())
            d64(fake); d64(real.double()); d64(eps * real.double() + (1 - eps) * fake)
        hk.remove()
        hcat = torch.cat(hs)
        if float(hcat.abs().min()) >= 0.08 * float(hcat.abs().mean()):
            break
    else:
        pytest.skip("no seed with a clear head margin")
    ref, _ = oracle64(G, D, real, noise, eps)
    res = {}
    for name, make_ops, dev in (("twin", lambda: RefOps(torch.bfloat16), "cpu"), ("hip", lambda: HipOps(torch.bfloat16, "cuda:0"), "cuda")):
        Gx, Dx = copy.deepcopy(G).to(dev).train(), copy.deepcopy(D).to(dev).train()
        E.tap_major_(Gx), E.tap_major_(Dx)
        Gn, Dn = E.build_gen_net(Gx), E.build_disc_net(Dx)
        ops = make_ops()
        r, z = real.to(dev), noise.to(dev)
        out = {}
        out["gl"] = float(E.gen_loss_grads(ops, Gn, Dn, z).cpu())
        out["G"] = {k: p.grad.detach().cpu().clone() for k, p in Gx.named_parameters()}
        out["dl"] = float(E.disc_loss_grads(ops, Gn, Dn, r, z).cpu())
        out["D"] = {k: p.grad.detach().cpu().clone() for k, p in Dx.named_parameters()}
        out["gp"] = float(E.gp_loss_grads(ops, Gn, Dn, r, z, eps, 10.0).cpu())
        out["P"] = {k: p.grad.detach().cpu().clone() for k, p in Dx.named_parameters()}
        out["buf"] = {"G." + k: b.detach().cpu().clone() for k, b in Gx.named_buffers()}
        out["buf"].update({"D." + k: b.detach().cpu().clone() for k, b in Dx.named_buffers()})
        res[name] = out
    t, h = res["twin"], res["hip"]
    for k in ("gl", "dl", "gp"):
        # (the D loss is a difference of two O(1) means: absolute floor 0.5)
        assert abs(h[k] - t[k]) <= 3e-2 * (abs(t[k]) + 0.5), (k, h[k], t[k])
    bad = []
    for grp in ("G", "D", "P"):
        for k in t[grp]:
            if float(ref[grp][k].abs().max()) == 0.0:
                continue
            _, e_h, _ = err(h[grp][k], ref[grp][k])
            _, e_t, _ = err(t[grp][k], ref[grp][k])
            _, e_ht, cos = err(h[grp][k], t[grp][k])
            line = f"  {grp} {k}: kernels-fp64 {e_h:.3f}  twin-fp64 {e_t:.3f}  kernels-twin {e_ht:.3f} cos {cos:.5f}"
            print(line)
            # two independent bf16 evaluations at distance e from fp64 sit ~sqrt(2) e apart; a 3-element image-bias gradient
            # is a sum with ~98 % cancellation and gets twice the allowance
            slack = 2.0 if t[grp][k].numel() <= 16 else 1.0
            if not (e_h <= slack * (1.5 * e_t + 1e-2) and e_ht <= slack * (1.6 * e_t + 1e-2)):
                bad.append(line)
    assert not bad, "\n".join(bad)
    for k in t["buf"]:
        if k.endswith("num_batches_tracked"):
            assert int(h["buf"][k]) == int(t["buf"][k])
        else:
            assert err(h["buf"][k], t["buf"][k])[0] <= 5e-3, k
