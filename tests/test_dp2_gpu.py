"""The data-parallel path at WORLD SIZE 2 on real kernels.  The GPU box has one device, and RCCL refuses two
ranks on one device, so the two rank processes share cuda:0 and all-reduce over gloo (which stages device tensors through
the host): everything else is the product's DP route exactly as an 8-GPU run takes it -- the loss plugins' step() ->
losses._Runner.run_dp: graph(prefix) / flush of the pending optimizer step / graph(rest) / gradient all-reduce started
eagerly, the optimizer step applied by the NEXT train_op after its prefix, 1/world folded into the backward seed, rank-local
BatchNorm statistics, the D-loss prefix = D(real) forward + backward.

Checked against a CPU emulation of DistributedDataParallel semantics on the oracle (two replicas with shared parameters and
their own BatchNorm buffers; per train_op: autograd on each replica's shard, gradients averaged, the same Adam step on both):
rank-local loss values, the parameter UPDATES (direction), the rank-local BatchNorm buffers; and both ranks must end with
bit-identical parameters.  fp32 kernels / fp32 wire for the tight comparison, then the bf16 kernels with the bf16 wire (when
this gloo build reduces bfloat16; fp32 wire otherwise) against the fp32 run -- that run also takes the gathered-factor route for
generator layer 0 (dist.G0_FACTORS: all-gather of z / gz0, the product formed over both ranks' samples inside the Adam step)."""
import copy
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R

IN_SIZE, STEP, ENC, N, ITERS = 32, 64, 128, 8, 3

WORKER = r'''
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from rna_gan_amd import dist as D_, losses as PL
import rna_gan_amd as P
from oracle import ref_cpu as R
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
D_.init_from_env(backend="gloo")
assert D_.world_size() == int(os.environ["WORLD_SIZE"]) and D_.active()
precision = os.environ["PRECISION"]
if precision in ("bf16", "fp16"):              # does this gloo build all-reduce 16-bit device tensors of the build's type?
    try:
        t = torch.ones(8, dtype=torch.bfloat16 if precision == "bf16" else torch.float16, device="cuda")
        dist.all_reduce(t)
        wire = ("bf16" if precision == "bf16" else "f16") if float(t[0]) == float(D_.world_size()) else "fp32"
    except Exception:
        wire = "fp32"
    if wire == "fp32":
        D_.COMPRESS = False
else:
    wire = "fp32"
in_size, step, enc, n, iters = [int(v) for v in os.environ["SHAPE"].split(",")]
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
G.set_precision(precision); D.set_precision(precision)
G, D = G.cuda().train(), D.cuda().train()
og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
lg, ld, lp = PL.WassersteinGeneratorLoss(), PL.WassersteinDiscriminatorLoss(), PL.WassersteinGradientPenalty()
losses = []
for it in range(iters):
    real = R.synthetic_images(n, in_size, seed=100 + 10 * it + rank).cuda()
    nz = [R.synthetic_normal(n, enc, seed=200 + 30 * it + 3 * rank + j).cuda() for j in range(3)]
    eps = torch.tensor([0.15 + 0.2 * it + 0.3 * rank], device="cuda")
    losses += [lg.step(G, D, og, nz[0]).item(), ld.step(G, D, od, real, nz[1]).item(), lp.step(G, D, od, real, nz[2], eps).item()]
PL.flush()
torch.cuda.synchronize()
fac = list(D_._factors.values())
torch.save({"losses": losses, "wire": wire, "factors": len(D_._factors) > 0,
            "factor_rows": [int(t.shape[0]) for f in fac for t in (f if isinstance(f, (tuple, list)) else [f]) if torch.is_tensor(t)],
            "G": {k: v.cpu() for k, v in G.state_dict().items()},
            "D": {k: v.cpu() for k, v in D.state_dict().items()}}, os.environ["OUT"] + str(rank))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_world(tmp_path, precision, world=2, n=N, iters=ITERS, extra_env=None, tag=""):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / ("dp%d_%s%s_rank" % (world, precision, tag)))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, REPO=repo, OUT=out, PRECISION=precision, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), RNAGAN_FORCE_DP="0",
                   SHAPE="%d,%d,%d,%d,%d" % (IN_SIZE, STEP, ENC, n, iters))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    for p in procs:
        try:
            _, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-3000:]
    return [torch.load(out + str(r)) for r in range(world)]


def _run_world2(tmp_path, precision, extra_env=None, tag=""):
    return _run_world(tmp_path, precision, 2, N, ITERS, extra_env, tag)


def _ddp_oracle(world=2, n=N, iters=ITERS):
    """DistributedDataParallel semantics on the CPU oracle: replicas share parameters, keep their own BatchNorm buffers."""
    mk_g = lambda: R.seeded_fill_(R.OracleDCGANGenerator(ENC, IN_SIZE, 3, STEP, nonlinearity=nn.LeakyReLU(0.2),
                                                          last_nonlinearity=nn.Tanh()), 7).double().train()
    mk_d = lambda: R.seeded_fill_(R.OracleDCGANDiscriminator(IN_SIZE, 3, STEP, nonlinearity=nn.LeakyReLU(0.2),
                                                              last_nonlinearity=nn.LeakyReLU(0.2)), 8).double().train()
    Gs, Ds = [mk_g() for _ in range(world)], [mk_d() for _ in range(world)]
    ogs = [R.make_adam(g.parameters(), 1e-4) for g in Gs]
    ods = [R.make_adam(d.parameters(), 4e-4) for d in Ds]

    def average_and_step(mods, opts):
        for ps in zip(*[list(m.parameters()) for m in mods]):
            g = sum(p.grad for p in ps) / len(ps)
            for p in ps:
                p.grad = g.clone()
        for o in opts:
            o.step()

    def zero(mods):
        for m in mods:
            for p in m.parameters():
                p.grad = None
    losses = [[] for _ in range(world)]
    for it in range(iters):
        data = []
        for r in range(world):
            real = R.synthetic_images(n, IN_SIZE, seed=100 + 10 * it + r).double()
            nz = [R.synthetic_normal(n, ENC, seed=200 + 30 * it + 3 * r + j).double() for j in range(3)]
            data.append((real, nz, 0.15 + 0.2 * it + 0.3 * r))
        zero(Gs + Ds)
        for r in range(world):
            l = R.generator_loss(Ds[r](Gs[r](data[r][1][0]))); l.backward(); losses[r].append(float(l.detach()))
        average_and_step(Gs, ogs)
        zero(Gs + Ds)
        for r in range(world):
            real, nz, _ = data[r]
            l = R.discriminator_loss(Ds[r](real), Ds[r](Gs[r](nz[1]).detach())); l.backward(); losses[r].append(float(l.detach()))
        average_and_step(Ds, ods)
        zero(Gs + Ds)
        for r in range(world):
            real, nz, eps = data[r]
            xhat = eps * real + (1 - eps) * Gs[r](nz[2])
            gp = R.gradient_penalty(xhat, Ds[r](xhat)); (10.0 * gp).backward(); losses[r].append(float(gp.detach()))
        average_and_step(Ds, ods)
    return Gs, Ds, losses


def test_world2_plugins_match_ddp_semantics(tmp_path):
    res = _run_world2(tmp_path, "fp32")
    Gs, Ds, want_losses = _ddp_oracle()
    init_g = R.seeded_fill_(R.OracleDCGANGenerator(ENC, IN_SIZE, 3, STEP, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.Tanh()), 7).state_dict()
    init_d = R.seeded_fill_(R.OracleDCGANDiscriminator(IN_SIZE, 3, STEP, nonlinearity=nn.LeakyReLU(0.2),
                                                       last_nonlinearity=nn.LeakyReLU(0.2)), 8).state_dict()
    for r in range(2):
        got = res[r]["losses"]
        for i, (a, b) in enumerate(zip(got, want_losses[r])):
            assert np.isfinite(a) and abs(a - b) <= 2e-3 * (abs(b) + 0.5), (r, i, got, want_losses[r])
        for name, sd, mod, init in (("G", res[r]["G"], Gs[r], init_g), ("D", res[r]["D"], Ds[r], init_d)):
            params = dict(mod.named_parameters())
            for k, v in sd.items():
                if k in params:
                    du = v.double() - init[k].double()
                    dr = params[k].detach().double() - init[k].double()
                    cos = float((du * dr).sum() / (du.norm() * dr.norm() + 1e-30))
                    # nine Adam steps of ~lr * sign(g): the cosine counts sign agreements, element by element
                    assert cos >= (0.98 if dr.numel() >= 4096 else 0.9), (r, name, k, cos)
                elif k.endswith("num_batches_tracked"):
                    assert int(v) == int(dict(mod.named_buffers())[k]), (r, k)
                else:       # BatchNorm running statistics are RANK-LOCAL: replica r's buffers, not an average
                    b = dict(mod.named_buffers())[k].double()
                    assert float((v.double() - b).norm() / (b.norm() + 1e-30)) <= 2e-3, (r, name, k)
    # one all-reduced gradient, one optimizer: the two ranks hold the same parameters bit for bit; their statistics differ
    for name in ("G", "D"):
        a, b = res[0][name], res[1][name]
        for k in a:
            if "running_" in k:
                continue
            assert torch.equal(a[k], b[k]), (name, k)
    assert any(not torch.equal(res[0]["D"][k], res[1]["D"][k]) for k in res[0]["D"] if "running_mean" in k)


def test_world2_bf16_kernels_and_wire(tmp_path):
    f32 = _run_world2(tmp_path, "fp32")
    b16 = _run_world2(tmp_path, "bf16")
    print("all-reduce wire of the bf16 run:", b16[0]["wire"])
    # generator layer 0's weight gradient travelled as all-gathered factors (dist.G0_FACTORS) in the bf16 run: both ranks formed
    # sum_r z_r^T gz0_r inside the fused Adam step; the fp32 run all-reduced the product
    assert b16[0]["factors"] and b16[1]["factors"] and not f32[0]["factors"]
    for name in ("G", "D"):
        for k in b16[0][name]:
            if "running_" not in k:
                assert torch.equal(b16[0][name][k], b16[1][name][k]), (name, k)
    for r in range(2):
        for i, (a, b) in enumerate(zip(b16[r]["losses"], f32[r]["losses"])):
            assert np.isfinite(a) and abs(a - b) <= (0.35 if i % 3 == 2 else 6e-2) * (abs(b) + 0.5), (r, i, a, b)
    for name in ("G", "D"):
        for k, v in b16[0][name].items():
            if v.dtype.is_floating_point and "running_" not in k:
                rel = float((v.double() - f32[0][name][k].double()).norm() / (f32[0][name][k].double().norm() + 1e-30))
                assert rel <= 1e-2, (name, k, rel)


def test_world2_fp16_kernels_wire_and_loss_scale(tmp_path):
    """BASELINE configs[3]'s arithmetic under data parallel: the fp16 build on two ranks -- fp32 all-reduce (the fp16 build's
    default: loss-scaled weight gradients overflow an fp16 wire, dist.F16_WIRE), all-gathered fp16 G.0 factors, the static loss
    scale (x 1/world in the same seed) removed inside the Adam kernels -- against the fp32 run: finite, rank-identical
    parameters; losses and updated parameters within the bf16 run's bounds or tighter."""
    f32 = _run_world2(tmp_path, "fp32")
    f16 = _run_world2(tmp_path, "fp16")
    print("all-reduce wire of the fp16 run:", f16[0]["wire"])
    assert f16[0]["factors"] and f16[1]["factors"]
    for name in ("G", "D"):
        for k in f16[0][name]:
            if "running_" not in k:
                assert torch.equal(f16[0][name][k], f16[1][name][k]), (name, k)
    for r in range(2):
        for i, (a, b) in enumerate(zip(f16[r]["losses"], f32[r]["losses"])):
            # (iterations 2 and 3 run on parameters that already differ by one Adam step of sign-like updates: 1.9e-2 seen)
            assert np.isfinite(a) and abs(a - b) <= (0.1 if i % 3 == 2 else 4e-2) * (abs(b) + 0.5), (r, i, a, b)
    worst = 0.0
    for name in ("G", "D"):
        for k, v in f16[0][name].items():
            if v.dtype.is_floating_point and "running_" not in k:
                rel = float((v.double() - f32[0][name][k].double()).norm() / (f32[0][name][k].double().norm() + 1e-30))
                worst = max(worst, rel)
                assert rel <= 1e-2, (name, k, rel)
    print("largest relative parameter difference fp16 vs fp32 after %d iterations: %.3g" % (ITERS, worst))


def test_world2_factors_with_an_optimizer_that_does_not_step_from_the_wire(tmp_path):
    """with generator layer 0's gradient travelling as gathered factors AND the optimizer not stepping from
    the bf16 wire (RNAGAN_DP_FUSED_WIDEN=0), the all-reduced TAIL of the generator's gradient must still be widened back into
    .grad -- otherwise every layer behind layer 0 steps from the rank-local gradient and the ranks drift apart silently.
    Both ranks must end bit-identical, and equal to the default route (Adam reading the wire) up to nothing at all: the two
    routes feed Adam the same bf16-rounded averaged gradient."""
    ref = _run_world2(tmp_path, "bf16")
    got = _run_world2(tmp_path, "bf16", extra_env={"RNAGAN_DP_FUSED_WIDEN": "0"}, tag="_nowire")
    assert got[0]["factors"] and got[1]["factors"]
    for name in ("G", "D"):
        for k in got[0][name]:
            if "running_" not in k:
                assert torch.equal(got[0][name][k], got[1][name][k]), ("ranks differ", name, k)
    if ref[0]["wire"] == "bf16":
        for k, v in got[0]["G"].items():
            if v.dtype.is_floating_point and "running_" not in k:
                rel = float((v.double() - ref[0]["G"][k].double()).norm() / (ref[0]["G"][k].double().norm() + 1e-30))
                assert rel <= 2e-3, (k, rel)


def test_world2_prefix_with_paired_weight_gradients(tmp_path):
    """RNAGAN_DP_PREFIX_BWD=2 (the default): the D-loss prefix runs D(real)'s forward and data-gradient chain only; each layer's
    conv weight gradient is one two-segment launch over the real and the fake half in the rest (engine.disc_loss_prefix_dgrad /
    disc_loss_rest_pairw).  Same mathematics as mode 1 (real half's weight gradients written in the prefix, fake half's
    accumulated): fp32 kernels, fp32 wire -- rank-identical parameters, losses equal to the other route's to rounding."""
    ref = _run_world2(tmp_path, "fp32", extra_env={"RNAGAN_DP_PREFIX_BWD": "1"}, tag="_mode1")
    got = _run_world2(tmp_path, "fp32", extra_env={"RNAGAN_DP_PREFIX_BWD": "2"}, tag="_pairw")
    for name in ("G", "D"):
        for k in got[0][name]:
            if "running_" not in k:
                assert torch.equal(got[0][name][k], got[1][name][k]), ("ranks differ", name, k)
    for r in range(2):
        for i, (a, b) in enumerate(zip(got[r]["losses"], ref[r]["losses"])):
            assert np.isfinite(a) and abs(a - b) <= 2e-3 * (abs(b) + 0.5), (r, i, a, b)
    for name in ("G", "D"):
        for k, v in got[0][name].items():
            if v.dtype.is_floating_point and "running_" not in k:
                rel = float((v.double() - ref[0][name][k].double()).norm() / (ref[0][name][k].double().norm() + 1e-30))
                assert rel <= 2e-3, (name, k, rel)



@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_world2_whole_route(tmp_path, precision):
    """RNAGAN_DP_ROUTE=whole: a train_op is the single process's gradient body as ONE graph (D(real) + D(fake) as one double
    batch, the penalty's fake batch out of the D-loss step's generator pass), then the all-reduce, waited for at once, then the
    optimizer step -- no prefix, nothing in flight.  Same DDP semantics as the prefix route: rank-identical parameters; fp32:
    losses and parameters equal to the prefix route's to rounding; bf16: within the bf16 tolerances."""
    ref = _run_world2(tmp_path, precision, tag="_prefix_" + precision)
    got = _run_world2(tmp_path, precision, extra_env={"RNAGAN_DP_ROUTE": "whole"}, tag="_whole_" + precision)
    for name in ("G", "D"):
        for k in got[0][name]:
            if "running_" not in k:
                assert torch.equal(got[0][name][k], got[1][name][k]), ("ranks differ", name, k)
    tol_l, tol_p = (2e-3, 2e-3) if precision == "fp32" else (6e-2, 2e-2)
    for r in range(2):
        for i, (a, b) in enumerate(zip(got[r]["losses"], ref[r]["losses"])):
            assert np.isfinite(a) and abs(a - b) <= (0.35 if precision == "bf16" and i % 3 == 2 else tol_l) * (abs(b) + 0.5), (r, i, a, b)
    for name in ("G", "D"):
        for k, v in got[0][name].items():
            if v.dtype.is_floating_point and "running_" not in k:
                rel = float((v.double() - ref[0][name][k].double()).norm() / (ref[0][name][k].double().norm() + 1e-30))
                assert rel <= tol_p, (name, k, rel)


@pytest.mark.parametrize("world", [4, 8])
def test_world4_and_world8_plugins_match_ddp_semantics(tmp_path, world):
    """nothing above world size 2 had ever executed.  4 and 8 rank processes share the box's GPU
    (gloo, device tensors staged through the host) and run the product's data-parallel route -- graph(prefix) / flush /
    graph(rest) / eager all-reduce start, 1 / world in the backward seed, rank-local BatchNorm statistics, the paired-weight-
    gradient D-loss prefix -- for two iterations at 4 samples per rank, fp32 kernels and wire: every rank's loss values against
    DistributedDataParallel semantics emulated on the CPU oracle (world replicas, gradients averaged, one Adam step), the
    parameter updates' direction, the rank-local running statistics, and bit-identical parameters on ALL ranks."""
    n, iters = 4, 2
    res = _run_world(tmp_path, "fp32", world, n, iters)
    Gs, Ds, want_losses = _ddp_oracle(world, n, iters)
    init_g = R.seeded_fill_(R.OracleDCGANGenerator(ENC, IN_SIZE, 3, STEP, nonlinearity=nn.LeakyReLU(0.2),
                                                   last_nonlinearity=nn.Tanh()), 7).state_dict()
    init_d = R.seeded_fill_(R.OracleDCGANDiscriminator(IN_SIZE, 3, STEP, nonlinearity=nn.LeakyReLU(0.2),
                                                       last_nonlinearity=nn.LeakyReLU(0.2)), 8).state_dict()
    for r in range(world):
        for i, (a, b) in enumerate(zip(res[r]["losses"], want_losses[r])):
            # G-loss / D-loss values: 2e-3.  The penalty (||dD/dxhat|| - 1)^2 of FOUR samples, evaluated behind the
            # discriminator's Adam step (whose sign-like first updates differ between fp32 and fp64 in the elements whose
            # averaged gradient is near zero -- more of them the more ranks are averaged): 2e-2 (measured at 8 ranks: 5e-3, 1.1e-2)
            # (second iteration's G-loss / D-loss, behind six sign-like Adam steps: measured up to 2.3e-3 at 8 ranks: 5e-3)
            tol = 2e-2 if i % 3 == 2 else (2e-3 if i < 3 else 5e-3)
            assert np.isfinite(a) and abs(a - b) <= tol * (abs(b) + 0.5), (r, i, res[r]["losses"], want_losses[r])
        for name, sd, mod, init in (("G", res[r]["G"], Gs[r], init_g), ("D", res[r]["D"], Ds[r], init_d)):
            params = dict(mod.named_parameters())
            for k, v in sd.items():
                if k in params:
                    du = v.double() - init[k].double()
                    dr = params[k].detach().double() - init[k].double()
                    cos = float((du * dr).sum() / (du.norm() * dr.norm() + 1e-30))
                    assert cos >= (0.97 if dr.numel() >= 4096 else 0.85), (r, name, k, cos)
                elif k.endswith("num_batches_tracked"):
                    assert int(v) == int(dict(mod.named_buffers())[k]), (r, k)
                else:
                    b = dict(mod.named_buffers())[k].double()
                    assert float((v.double() - b).norm() / (b.norm() + 1e-30)) <= 2e-3, (r, name, k)
    for r in range(1, world):
        for name in ("G", "D"):
            for k in res[0][name]:
                if "running_" not in k:
                    assert torch.equal(res[0][name][k], res[r][name][k]), (r, name, k)


def test_world8_bf16_kernels_wire_and_gathered_factors(tmp_path):
    """8 ranks, bf16 kernels, the bf16 wire (when this gloo build reduces bfloat16) and generator layer 0's gradient as
    all-gathered factors with K = 8 x n samples (dist.G0_FACTORS; the fused G.0 gradient + Adam kernel contracts over all
    ranks' samples): all ranks bit-identical, losses and weights within the bf16 tolerances of the fp32 run of the same world."""
    n, iters, world = 4, 2, 8
    f32 = _run_world(tmp_path, "fp32", world, n, iters)
    b16 = _run_world(tmp_path, "bf16", world, n, iters)
    print("all-reduce wire of the 8-rank bf16 run:", b16[0]["wire"], "factor buffers (rows):", b16[0]["factor_rows"])
    assert all(x["factors"] for x in b16) and not f32[0]["factors"]
    assert world * n in b16[0]["factor_rows"], b16[0]["factor_rows"]          # the gathered buffers hold ALL ranks' samples
    for r in range(1, world):
        for name in ("G", "D"):
            for k in b16[0][name]:
                if "running_" not in k:
                    assert torch.equal(b16[0][name][k], b16[r][name][k]), (r, name, k)
    for r in range(world):
        for i, (a, b) in enumerate(zip(b16[r]["losses"], f32[r]["losses"])):
            assert np.isfinite(a) and abs(a - b) <= (0.35 if i % 3 == 2 else 6e-2) * (abs(b) + 0.5), (r, i, a, b)
    for name in ("G", "D"):
        for k, v in b16[0][name].items():
            if v.dtype.is_floating_point and "running_" not in k:
                rel = float((v.double() - f32[0][name][k].double()).norm() / (f32[0][name][k].double().norm() + 1e-30))
                assert rel <= 1e-2, (name, k, rel)
