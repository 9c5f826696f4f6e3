import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import rna_gan_amd as P
from rna_gan_amd import losses as PL, engine as E
from oracle import ref_cpu as R
in_size, step, enc, n = int(sys.argv[1]), 64, 128, int(sys.argv[2])
G0 = R.seeded_fill_(R.OracleDCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 7)
D0 = R.seeded_fill_(R.OracleDCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2)), 8)
for batched in (0, 1):
    PL.D_BATCHED = bool(batched)
    G = P.DCGANGenerator(enc, in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh())
    D = P.DCGANDiscriminator(in_size, 3, step, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.LeakyReLU(0.2))
    G.load_state_dict(G0.state_dict()); D.load_state_dict(D0.state_dict())
    G, D = G.cuda().train(), D.cuda().train()
    og = P.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999)).bind(G)
    od = P.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999)).bind(D)
    lg, ld, lp = PL.WassersteinGeneratorLoss(), PL.WassersteinDiscriminatorLoss(), PL.WassersteinGradientPenalty()
    ops, gn = G.runtime(); _, dn = D.runtime()
    for it in range(4):
        real = R.synthetic_images(n, in_size, seed=100 + it).cuda()
        nz = [R.synthetic_normal(n, enc, seed=200 + 3 * it + j).cuda() for j in range(3)]
        eps = torch.tensor([0.1 + 0.2 * it], device="cuda")
        a = lg.step(G, D, og, nz[0]).item()
        b = ld.step(G, D, od, real, nz[1]).item()
        o_r, _ = E.disc_forward(ops, dn, real, update_running=False)
        c = lp.step(G, D, od, real, nz[2], eps).item()
        o_r2, _ = E.disc_forward(ops, dn, real, update_running=False)
        print("batched", batched, "it", it, "G %.4f D %.4f GP %.4f | mean D(real) after D-step %.4f after GP-step %.4f" %
              (a, b, c, float(o_r.mean()), float(o_r2.mean())), flush=True)
