"""Which replay clobbers the G-step / D-step loss tensors?  The three plugin steps by hand, value of every loss tensor read
after EACH step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rna_gan_amd import losses as PL
args = bench.parse_args([])
dev = torch.device("cuda:0")
one_step, flush, N, info = bench.hip_workload(args, 0, 1, dev)
h = info["handles"]
G, D, og, od, (lg, ld, lp) = h["G"], h["D"], h["og"], h["od"], h["losses"]
real, rna = h["real"], h["rna"]
gen = torch.Generator().manual_seed(1)
for it in range(10):
    PL.new_batch()
    us = [torch.empty(N, 2048).uniform_(-0.3, 0.3, generator=gen).to(dev) for _ in range(3)]
    eps = torch.empty(1).uniform_(0, 1, generator=gen).to(dev)
    a = lg.step(G, D, og, rna, us[0]); torch.cuda.synchronize(); va = a.item()
    b = ld.step(G, D, od, real, rna, us[1], next_u=us[2]); torch.cuda.synchronize(); vb = b.item(); va2 = a.item()
    c = lp.step(G, D, od, real, rna, us[2], eps); torch.cuda.synchronize(); vc = c.item(); va3 = a.item(); vb3 = b.item()
    if it in (4, 6):
        for nm, mod in (("G", G), ("D", D)):
            o, _ = mod.runtime()
            for w in [o._wsbuf] + o._ws_retired:
                if w is not None:
                    print("   ", nm, "ws", w.data_ptr(), w.data_ptr() + w.numel(), flush=True)
    print(it, "G after G/D/GP: %.6f %.6f %.6f | D after D/GP: %.6f %.6f | GP %.6f" % (va, va2, va3, vb, vb3, vc),
          "ptrs", a.data_ptr(), b.data_ptr(), c.data_ptr(), flush=True)
