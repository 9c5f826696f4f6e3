"""The step bench.py TIMES, checked against the CPU oracle directly.

bench.hip_workload builds BASELINE configs[1] (batch 64, 256x256, bf16, wganvae plugins) and its one_step() is the timed
body: the three loss plugins' step() with one betaVAE encode per batch (losses._LatentCache), the D-loss step told the
penalty step's draw (one generator pass over the double batch, gen_forward_pair), D(real) + D(fake) as one double batch,
the last BatchNorm fused into the image layer, and -- from the third call of each train_op on -- HIP-graph replay.  Here
that very function runs, and an iteration is compared with oracle.ref_cpu.train_iteration (the restatement of
src/wgan_loss.py:96-129, :221-263, :354-389 pinned by fixture F5) on conditioned_noise(u, encode_latent(vae, rna)) for the
SAME draws: once while everything is still eager (iteration 1) and once when every train_op is a graph replay.  Before
each compared iteration the oracle is re-synchronised with the product's state (parameters, BatchNorm buffers, Adam
moments and step counts), so that each comparison is of ONE iteration from identical state (bf16 and fp32 trajectories
drift apart chaotically over several sign-like Adam steps; the single step is what has a tolerance).
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_state(sd):
    return {k: v.detach().cpu().contiguous().clone() for k, v in sd.items()}


def _cpu_optim_state(opt):
    sd = opt.state_dict()
    state = {i: {k: (v.detach().cpu().contiguous().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
             for i, st in sd["state"].items()}
    return {"state": state, "param_groups": copy.deepcopy(sd["param_groups"])}


def test_benchmarked_step_vs_oracle():
    sys.path.insert(0, ROOT)
    import bench
    from rna_gan_amd import graphed
    from rna_gan_amd import losses as PL
    if not graphed.ENABLED:
        pytest.skip("RNAGAN_GRAPHS=0: the benchmarked step replays HIP graphs")
    args = bench.parse_args([])                       # the defaults ARE the benchmarked configuration
    assert args.batch == 64 and args.precision == "bf16" and not args.api_path
    device = torch.device("cuda:0")
    one_step, flush, N, info = bench.hip_workload(args, 0, 1, device)
    h = info["handles"]
    G, D, og, od, (lg, ld, lp) = h["G"], h["D"], h["og"], h["od"], h["losses"]
    real_cpu, rna_cpu, gen = h["real"].cpu(), h["rna"].cpu(), h["host_generator"]

    Go = R.OracleDCGANGenerator(2048, 256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()).train()
    Do = R.OracleDCGANDiscriminator(256, 3, 64, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)).train()
    vae = R.OracleBetaVAE(19198, 2048, [6000, 4000, 2048], [4000, 6000]).eval()
    vae.load_state_dict(_cpu_state(lg.betavae.state_dict()))
    ogo, odo = R.make_adam(Go.parameters(), 1e-4), R.make_adam(Do.parameters(), 4e-4)
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    with torch.no_grad():
        z = R.encode_latent(vae, rna_cpu)             # frozen encoder, fixed RNA batch: one latent for every iteration

    def compared_iteration(tag, first):
        torch.cuda.synchronize()
        before_g, before_d = _cpu_state(G.state_dict()), _cpu_state(D.state_dict())
        Go.load_state_dict(before_g); Do.load_state_dict(before_d)
        if not first:                                  # Adam moments / step counts of the product -> the oracle's optimizers
            ogo.load_state_dict(_cpu_optim_state(og)); odo.load_state_dict(_cpu_optim_state(od))
        # the draws one_step() is about to make on the workload's CPU generator: three U(-0.3, 0.3) of (N, 2048), then eps
        twin = torch.Generator(device="cpu")
        twin.set_state(gen.get_state())
        us = [torch.empty(N, 2048).uniform_(-0.3, 0.3, generator=twin) for _ in range(3)]
        eps = float(torch.empty(1).uniform_(0.0, 1.0, generator=twin))
        got = [float(l.item()) for l in one_step()]
        torch.cuda.synchronize()
        assert torch.equal(twin.get_state(), gen.get_state()), "one_step() drew differently from the twin generator"
        with torch.no_grad():
            noises = [R.conditioned_noise(u, z) for u in us]
        ref = R.train_iteration(Go, Do, ogo, odo, real_cpu, noises, eps)
        want = [ref["g"], ref["d"], ref["gp"]]
        print("%s losses hip/oracle:" % tag, got, want)
        for g_, w_ in zip(got, want):
            assert np.isfinite(g_) and abs(g_ - w_) <= 6e-2 * (abs(w_) + 0.1), (tag, got, want)
        worst = (1.0, None)
        for name_, mod, modo, before in (("G", G, Go, before_g), ("D", D, Do, before_d)):
            sd, sdo = _cpu_state(mod.state_dict()), modo.state_dict()
            for k, _ in modo.named_parameters():
                du_hip = sd[k].double() - before[k].double()
                du_ref = sdo[k].double() - before[k].double()
                cos = float((du_hip * du_ref).sum() / (du_hip.norm() * du_ref.norm() + 1e-30))
                worst = min(worst, (cos, name_ + "." + k))
                if du_ref.numel() >= 100000:
                    print("%s %s.%-22s numel %9d  update cosine %.4f" % (tag, name_, k, du_ref.numel(), cos))
                # thresholds of test_full_size_batch64_bf16 (an Adam step is ~lr * sign(g): the cosine counts sign agreements)
                assert cos >= (0.8 if du_ref.numel() >= 4096 else 0.65), (tag, name_, k, cos)
            for k, b in modo.named_buffers():
                if k.endswith("num_batches_tracked"):
                    assert int(sd[k]) == int(b), (tag, k)
                else:
                    rel = float((sd[k].double() - b.double()).norm() / (b.double().norm() + 1e-30))
                    assert rel <= 3e-2, (tag, name_, k, rel)
        print("%s worst update cosine:" % tag, worst)

    compared_iteration("iteration 1 (eager)", first=True)
    assert all(sg.graph is None for r in (lg._runner, ld._runner, lp._runner) for sg in r._graphs.values())
    for _ in range(8):                                 # capture happens on a train_op's third call per staleness variant
        one_step()
    torch.cuda.synchronize()
    calls = {id(sg): sg.calls for sg in graphed._captured}
    assert len(calls) >= 3
    compared_iteration("iteration 10 (graph replay)", first=False)
    replayed = [sg for sg in graphed._captured if sg.calls == calls.get(id(sg), -1) + 1]
    per_plugin = [sum(1 for sg in r._graphs.values() if sg in replayed) for r in (lg._runner, ld._runner, lp._runner)]
    assert per_plugin == [1, 1, 1], "each train_op of the compared iteration must have been ONE graph replay: %s" % per_plugin
    # the D-loss graph is the look-ahead form (its generator pass also produced the penalty step's fake batch) and the
    # penalty graph the form that consumed it
    assert any("lookahead" in str(k) for k in ld._runner._graphs) and any("gpf" in str(k) for k in lp._runner._graphs)
    flush()
    PL.new_batch()
