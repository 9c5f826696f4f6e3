#!/usr/bin/env python3
# small-shape check of the fused split-K BatchNorm path: the three gradient computations of the smoke configuration with
# split_bn on / off on identical nets; reports the per-tensor relative difference of the gradients and the loss values
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import engine as E
from oracle import ref_cpu as R
in_size, step, enc, n = int(os.environ.get("IN", 32)), 64, 128, int(os.environ.get("N", 8))
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
real = R.synthetic_images(n, in_size, seed=1).cuda()
noises = [R.synthetic_normal(n, enc, seed=2 + j).cuda() for j in range(3)]
res = {}
for split in (0, 1):
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
    G.set_precision("bf16"); D.set_precision("bf16")
    G, D = G.cuda().train(), D.cuda().train()
    ops, gn = G.runtime(); _, dn = D.runtime()
    ops.split_bn = bool(split)
    out = {}
    l = E.gen_loss_grads(ops, gn, dn, noises[0]); torch.cuda.synchronize()
    out["g"] = (float(l), {k: p.grad.detach().clone() for k, p in G.named_parameters()})
    l = E.disc_loss_grads_batched(ops, gn, dn, real, noises[1]) if hasattr(E, "disc_loss_grads_batched") else None
    torch.cuda.synchronize()
    out["d"] = (float(l[0] if isinstance(l, (tuple, list)) else l), {k: p.grad.detach().clone() for k, p in D.named_parameters()})
    l = E.gp_loss_grads(ops, gn, dn, real, noises[2], 0.4, 10.0); torch.cuda.synchronize()
    out["p"] = (float(l), {k: p.grad.detach().clone() for k, p in D.named_parameters()})
    res[split] = out
for step_ in ("g", "d", "p"):
    a, b = res[0][step_], res[1][step_]
    print(step_, "loss unfused %.6f fused %.6f" % (a[0], b[0]))
    for k in a[1]:
        rel = float((a[1][k] - b[1][k]).norm() / (a[1][k].norm() + 1e-30))
        if rel > 1e-3:
            print("    %-28s rel diff %.3e  norm %.3e" % (k, rel, float(a[1][k].norm())))
