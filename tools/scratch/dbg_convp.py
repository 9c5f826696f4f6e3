import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rna_gan_amd import _abi
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps
lib = _abi.load(); hip = HipOps(torch.bfloat16, "cuda:0")
def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed); return torch.from_numpy((g.standard_normal(size=shape) * scale).astype(np.float32))
for (N, Ws) in [(1, 64), (5, 64), (1, 16), (2, 32), (64, 64)]:
    O, I = 128, 64
    w = rnd((O, 4, 4, I), 1, (2.0 / (O * 4)) ** 0.5).cuda()
    cw = ConvW(w, None, torch.zeros_like(w), None, "OHWI")
    g = rnd((N, Ws, Ws, O), 3).cuda().to(torch.bfloat16)
    for rep in range(3):
        outs = []
        for on in (1, 0):
            lib.rg_set_option(b"convp", on)
            u, su = hip.conv_up(g, cw, want_stats=True)
            torch.cuda.synchronize(); outs.append(u.float())
        lib.rg_set_option(b"convp", -1)
        bad = ~torch.isfinite(outs[0])
        d = (outs[0] - outs[1]).abs()
        d[bad] = 1e9
        nz = (d > 0).nonzero()
        print(N, Ws, "rep", rep, "nonfinite", int(bad.sum()), "mismatch", nz.shape[0], "first", nz[:6].tolist())
