#!/usr/bin/env python3
# ONE pair of nets, one generator step, then the D-loss gradients with split_bn off / on / off on the SAME weights
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import engine as E, losses as PL
from oracle import ref_cpu as R
in_size, step, enc, n = 32, 64, 128, 8
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
real = R.synthetic_images(n, in_size, seed=1).cuda()
noises = [R.synthetic_normal(n, enc, seed=2 + j).cuda() for j in range(3)]
G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
G.set_precision("bf16"); D.set_precision("bf16")
G, D = G.cuda().train(), D.cuda().train()
og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
ops, gn = G.runtime(); _, dn = D.runtime()
ops.split_bn = bool(int(os.environ.get("GSTEP_SPLIT", "0")))
PL._g_step(G, D, og, noises[0]).item()
def run(split):
    ops.split_bn = bool(split)
    img, _ = E._gen_fwd(ops, gn, noises[1], keep=False)
    l = E.disc_loss_grads_batched(ops, gn, dn, real, noises[1]); torch.cuda.synchronize()
    return float(l[0] if isinstance(l, (tuple, list)) else l), torch.cat([p.grad.detach().reshape(-1).clone() for p in D.parameters()]), img.clone()
r = [run(0), run(1), run(0), run(1)]
for i in (1, 2, 3):
    a, b = r[i][1].double(), r[0][1].double()
    print("call %d (split %d) vs call 0: loss %.6f vs %.6f, grad rel-L2 %.4f, image rel %.2e" %
          (i, i % 2, r[i][0], r[0][0], float((a - b).norm() / b.norm()), float((r[i][2] - r[0][2]).norm() / r[0][2].norm())))
