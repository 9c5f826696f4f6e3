#!/usr/bin/env python3
"""Per-layer timing of the MFMA conv kernels at the reference model's shapes (batch 64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ops = HipOps(torch.bfloat16, "cuda:0")
dev = torch.device("cuda:0")
REP = 10

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e-3

rows = []
c, s = 64, 128
tot = {"down": [0, 0], "up": [0, 0], "wgrad": [0, 0]}
for l in range(5):
    I, O, hs = c, 2 * c, s            # D layer l+1: x[N,hs,hs,I] -> [N,hs/2,hs/2,O]
    w = torch.randn(O, I, 4, 4, device=dev) * (2.0 / (I * 16)) ** 0.5
    wt = w.permute(0, 2, 3, 1).contiguous()
    cw = ConvW(wt, None, torch.zeros_like(wt), None, "OHWI")
    x = torch.randn(N, hs, hs, I, device=dev).to(torch.bfloat16)
    g = torch.randn(N, hs // 2, hs // 2, O, device=dev).to(torch.bfloat16)
    flops = 2.0 * N * (hs // 2) ** 2 * O * I * 16
    for kind, fn in (("down", lambda: ops.conv_down(x, cw)), ("up", lambda: ops.conv_up(g, cw)),
                     ("wgrad", lambda: ops.conv_wgrad(g, x, cw, False))):
        t = timeit(fn)
        tot[kind][0] += flops; tot[kind][1] += t
        print(f"I={I:5d} O={O:5d} hi={hs:4d}  {kind:6s} {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF/s")
    c *= 2; s //= 2
for k, (f, t) in tot.items():
    print(f"{k}: total {t*1e3:.3f} ms, {f/t/1e12:.1f} TF/s")
