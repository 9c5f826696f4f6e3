import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import engine as E
from oracle import ref_cpu as R
in_size, step, enc, n = int(sys.argv[1]), 64, 128, int(sys.argv[2])
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
res = []
for batched in (0, 1):
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
    G, D = G.cuda().train(), D.cuda().train()
    ops, gn = G.runtime(); _, dn = D.runtime()
    real = R.synthetic_images(n, in_size, seed=100).cuda()
    nz = R.synthetic_normal(n, enc, seed=200).cuda()
    fn = E.disc_loss_grads_batched if batched else E.disc_loss_grads
    loss = fn(ops, gn, dn, real, nz)
    torch.cuda.synchronize()
    res.append((float(loss), {k: p.grad.detach().float().cpu().clone() for k, p in D.named_parameters()},
                {k: b.detach().float().cpu().clone() for k, b in D.named_buffers()}))
print("loss", res[0][0], res[1][0])
for k in res[0][1]:
    a, b = res[0][1][k], res[1][1][k]
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print("%-28s |a| %.3e |b| %.3e cos %.5f" % (k, a.norm(), b.norm(), cos))
for k in res[0][2]:
    a, b = res[0][2][k], res[1][2][k]
    print("buf %-24s diff %.3e" % (k, float((a - b).abs().max())))
