#!/usr/bin/env python3
# the smoke configuration through the plugins' steps with split_bn on / off: parameters after each optimizer step
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import losses as PL
from oracle import ref_cpu as R
in_size, step, enc, n = 32, 64, 128, 8
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
real = R.synthetic_images(n, in_size, seed=1).cuda()
noises = [R.synthetic_normal(n, enc, seed=2 + j).cuda() for j in range(3)]
snap = {}
for split in (0, 1):
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
    G.set_precision("bf16"); D.set_precision("bf16")
    G, D = G.cuda().train(), D.cuda().train()
    og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
    od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
    ops, _ = G.runtime()
    ops.split_bn = bool(split)
    s = {}
    s["init_G"] = {k: v.detach().clone() for k, v in G.named_parameters()}
    s["init_D"] = {k: v.detach().clone() for k, v in D.named_parameters()}
    s["lg"] = PL._g_step(G, D, og, noises[0]).item()
    s["G1"] = {k: v.detach().clone() for k, v in G.named_parameters()}
    with torch.no_grad():
        s["fake"] = G(noises[1]).clone()
    s["ld"] = PL._d_step(G, D, od, real, noises[1], None).item()
    s["D1"] = {k: v.detach().clone() for k, v in D.named_parameters()}
    snap[split] = s
a, b = snap[0], snap[1]
print("losses", a["lg"], b["lg"], a["ld"], b["ld"])
print("fake after the G step: rel diff", float((a["fake"] - b["fake"]).norm() / a["fake"].norm()))
for tag, init in (("G1", "init_G"), ("D1", "init_D")):
    for k in a[tag]:
        ua, ub = a[tag][k] - a[init][k], b[tag][k] - b[init][k]
        cos = float((ua * ub).sum() / (ua.norm() * ub.norm() + 1e-30))
        frac_nonzero_b = float((ub != 0).float().mean())
        print("  %s %-26s update cosine %.4f  |ua| %.3e |ub| %.3e  moved(b) %.3f" % (tag, k, cos, float(ua.norm()), float(ub.norm()), frac_nonzero_b))
