#!/usr/bin/env python3
# the smoke test's ORIGINAL inputs (head margin 0.108 of mean 0.80 at the penalty step) in bf16 under the round's switches:
# is the penalty value's deviation from the oracle a conditioning effect (any path) or one path's bug?
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import losses as PL
from oracle import ref_cpu as R
in_size, step, enc, n = 32, 64, 128, 8
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
real = R.synthetic_images(n, in_size, seed=1)
noises = [R.synthetic_normal(n, enc, seed=2 + j) for j in range(3)]
G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
G.set_precision("bf16"); D.set_precision("bf16")
G, D = G.cuda().train(), D.cuda().train()
og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
rd = real.cuda()
g = PL._g_step(G, D, og, noises[0].cuda()).item()
d = PL._d_step(G, D, od, rd, noises[1].cuda(), None).item()
# head pre-activations of the penalty step's forward on the product's CURRENT discriminator
ops, gn, dn = PL._nets(G, D)
from rna_gan_amd import engine as E
with torch.no_grad():
    fake = G(noises[2].cuda())
xhat = 0.4 * rd + 0.6 * fake
out, ctx = E.disc_forward(ops, dn, xhat.contiguous().float(), False)
print("env", {k: v for k, v in os.environ.items() if k.startswith("RNAGAN_")}, "g %.5f d %.5f" % (g, d), "head h:", [round(float(x), 3) for x in ctx.h.reshape(-1)])
gp = PL._gp_step(G, D, od, rd, noises[2].cuda(), 0.4, 10.0).item()
print("   gp %.4f (oracle 11.9111; oracle head h: 0.508 1.714 -1.236 0.184 -0.108 -0.467 -1.686 -0.523)" % gp)
