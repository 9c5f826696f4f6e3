import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rna_gan_amd import _abi
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps
lib = _abi.load(); hip = HipOps(torch.bfloat16, "cuda:0")
def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed); return torch.from_numpy((g.standard_normal(size=shape) * scale).astype(np.float32))
for (N, Ws) in [(1, 16), (1, 64)]:
    O, I = 128, 64
    w = rnd((O, 4, 4, I), 1, (2.0 / (O * 4)) ** 0.5).cuda()
    cw = ConvW(w, None, torch.zeros_like(w), None, "OHWI")
    g = rnd((N, Ws, Ws, O), 3).cuda().to(torch.bfloat16)
    m = rnd((N, 2 * Ws, 2 * Ws, I), 21).cuda().to(torch.bfloat16)
    outs = []
    for on in (1, 0):
        lib.rg_set_option(b"convp", on)
        md = m.clone()
        if on:
            md._rg_sign_bits = hip.sign_pack(md)
        um = hip.conv_up(g, cw, md, 0.2)
        u, su = hip.conv_up(g, cw, want_stats=True)
        torch.cuda.synchronize(); outs.append((um.float(), u.float()))
    lib.rg_set_option(b"convp", -1)
    for tag, k in (("masked", 0), ("plain", 1)):
        a, b = outs[0][k], outs[1][k]
        d = (a - b).abs()
        nz = (d > 0).nonzero()
        print(N, Ws, tag, "mismatch", nz.shape[0], "of", a.numel(), "max", float(d.max()))
        if nz.shape[0]:
            ch = torch.bincount(nz[:, 3], minlength=64)
            print("   by channel:", ch.tolist())
            px = torch.bincount(nz[:, 2] % 32, minlength=32)
            print("   by x % 32:", px.tolist())
            print("   first:", nz[:8].tolist(), [float(a[tuple(i)]) for i in nz[:4]], [float(b[tuple(i)]) for i in nz[:4]])
            # ratio new / ref for mismatching elements (slope mix-up shows as 5 or 0.2)
            r = (a[d > 0] / b[d > 0])
            print("   ratio quantiles:", [float(x) for x in torch.quantile(r, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], device=r.device))])
