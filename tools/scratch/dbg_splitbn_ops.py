#!/usr/bin/env python3
# op-level scan at small shapes: conv (down / up) with deferred split-K slabs + BatchNorm forward / backward (1 / 2 groups),
# fused slab kernels vs the separate-launch path
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps
dev = torch.device("cuda:0")
def relmax(a, b): return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
gen = torch.Generator().manual_seed(1)
for (I, O, hs, n, groups) in [(64, 128, 16, 8, 1), (128, 256, 8, 8, 1), (64, 128, 16, 16, 2), (128, 256, 8, 16, 2),
                              (256, 512, 4, 8, 1), (64, 128, 32, 8, 1), (128, 256, 16, 8, 1), (128, 256, 16, 16, 2)]:
    fu, se = HipOps(torch.bfloat16, dev), HipOps(torch.bfloat16, dev)
    se.split_bn = False
    w = (torch.randn(O, 4, 4, I, generator=gen) * (2.0 / (I * 16)) ** 0.5).bfloat16().float().to(dev)
    cf, cs = ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI"), ConvW(w.clone(), None, torch.zeros_like(w), None, "OHWI")
    ho = hs // 2
    x = torch.randn(n, hs, hs, I, generator=gen).bfloat16().to(dev)          # high-res, I channels
    y = torch.randn(n, ho, ho, O, generator=gen).bfloat16().to(dev)          # low-res, O channels
    gO, bO = (1 + 0.1 * torch.randn(O, generator=gen)).to(dev), (0.1 * torch.randn(O, generator=gen)).to(dev)
    gI, bI = (1 + 0.1 * torch.randn(I, generator=gen)).to(dev), (0.1 * torch.randn(I, generator=gen)).to(dev)
    def fwd(ops, z, C, gam, bet):
        f = ops.bn_forward if groups == 1 else ops.bn_forward2
        return f(z, gam, bet, 0.2, 1e-5, 0.1)
    line = "I %4d O %4d hs %3d n %3d g %d:" % (I, O, hs, n, groups)
    # (1) conv_down -> bn_forward (D forward)
    res = []
    for ops, cw in ((fu, cf), (se, cs)):
        z, _ = ops.conv_down(x, cw, want_stats=True, defer=groups)
        used = getattr(z, "_rg_slabs", None) is not None
        a, m, iv = fwd(ops, z, O, gO, bO)
        res.append((z, a, m, iv, used))
    torch.cuda.synchronize()
    line += " down>fwd[%d] z %.1e a %.1e mean %.1e |" % (res[0][4], relmax(res[0][0], res[1][0]), relmax(res[0][1], res[1][1]), relmax(res[0][2], res[1][2]))
    # (2) conv_up -> bn_forward (G forward)
    res = []
    for ops, cw in ((fu, cf), (se, cs)):
        out = ops.conv_up(y, cw, want_stats=True, defer=groups) if "want_stats" in ops.conv_up.__code__.co_varnames else ops.conv_up(y, cw, defer=groups)
        z = out[0] if isinstance(out, tuple) else out
        used = getattr(z, "_rg_slabs", None) is not None
        a, m, iv = fwd(ops, z, I, gI, bI)
        res.append((z, a, m, iv, used))
    torch.cuda.synchronize()
    line += " up>fwd[%d] z %.1e a %.1e mean %.1e |" % (res[0][4], relmax(res[0][0], res[1][0]), relmax(res[0][1], res[1][1]), relmax(res[0][2], res[1][2]))
    # (3) conv_up -> bn_act_bwd (D backward) and (4) conv_down -> bn_act_bwd (G backward)
    for tag, src, C, gam, bet, conv, zshape in (("up>bwd", y, I, gI, bI, "conv_up", (n, hs, hs, I)), ("down>bwd", x, O, gO, bO, "conv_down", (n, ho, ho, O))):
        zb = (torch.randn(*zshape, generator=gen) * 1.3 + 0.2).bfloat16().to(dev)
        res = []
        for ops, cw in ((fu, cf), (se, cs)):
            _, mean, inv = fwd(ops, zb.clone(), C, gam, bet)
            ga = getattr(ops, conv)(src, cw, defer=groups)
            ga = ga[0] if isinstance(ga, tuple) else ga
            used = getattr(ga, "_rg_slabs", None) is not None
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            if groups == 1:
                gz, _, _ = ops.bn_act_bwd(zb, ga, mean, inv, gam, bet, 0.2, dg, db, False, keep_ga=False)
            else:
                gz = ops.bn_act_bwd2(zb, ga, mean, inv, gam, bet, 0.2, dg, db, False)
            res.append((gz, dg, db, used))
        torch.cuda.synchronize()
        line += " %s[%d] gz %.1e dgamma %.1e |" % (tag, res[0][3], relmax(res[0][0], res[1][0]), relmax(res[0][1], res[1][1]))
    print(line)
