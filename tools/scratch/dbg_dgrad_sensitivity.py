#!/usr/bin/env python3
# how sensitive is the D-loss gradient (smoke configuration, fp32 kernels, after one generator step) to a 0.36 % perturbation
# of the fake batch?  (conditioning of  grad[ mean D(fake) - mean D(real) ]  through per-half BatchNorm statistics)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn, copy
from oracle import ref_cpu as R
in_size, step, enc, n = 32, 64, 128, 8
G = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7).double().train()
D = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8).double().train()
og = R.make_adam(G.parameters(), 1e-4)
real = R.synthetic_images(n, in_size, seed=1).double()
noises = [R.synthetic_normal(n, enc, seed=2 + j).double() for j in range(3)]
R.generator_loss(D(G(noises[0]))).backward(); og.step()
for p in D.parameters(): p.grad = None
fake = G(noises[1]).detach()
def dgrad(f):
    for p in D.parameters(): p.grad = None
    l = R.discriminator_loss(D(real), D(f)); l.backward()
    return float(l), torch.cat([p.grad.reshape(-1).clone() for p in D.parameters()])
l0, g0 = dgrad(fake)
gen = torch.Generator().manual_seed(0)
for rel in (1e-4, 1e-3, 3.6e-3, 1e-2):
    pert = fake + rel * fake.norm() / fake.numel() ** 0.5 * torch.randn(fake.shape, generator=gen, dtype=torch.float64)
    l1, g1 = dgrad(pert)
    print("fake perturbed by %.1e (rel L2): loss %.6f -> %.6f, D-gradient rel-L2 change %.4f" % (rel, l0, l1, float((g1 - g0).norm() / g0.norm())))
