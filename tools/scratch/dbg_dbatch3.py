import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rna_gan_amd import losses as PL, synth as R
dev = torch.device("cuda:0")
for batched in (0, 1):
    PL.D_BATCHED = bool(batched)
    G, Dm, og, od, (lg, ld, lp) = bench.build(dev, "bf16", 64, 19198, 0)
    real = R.synthetic_images(64, 256, seed=1234).to(dev)
    rna = R.synthetic_rna(64, 19198, seed=4321, distinct=16).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(0)
    for it in range(5):
        PL.new_batch()
        u = [torch.empty(64, 2048).uniform_(-0.3, 0.3, generator=gen).to(dev) for _ in range(3)]
        eps = torch.empty(1).uniform_(0, 1, generator=gen).to(dev)
        a = lg.step(G, Dm, og, rna, u[0]).item()
        b = ld.step(G, Dm, od, real, rna, u[1]).item()
        c = lp.step(G, Dm, od, real, rna, u[2], eps).item()
        print("batched", batched, "it", it, "G %.4f D %.4f GP %.4f" % (a, b, c), flush=True)
