#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW
N = 64
ops = HipOps(torch.bfloat16, "cuda:0"); dev = torch.device("cuda:0")
def timeit(fn, rep=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e-3
w = torch.randn(64, 3, 4, 4, device=dev) * 0.1; b = torch.randn(64, device=dev) * 0.1; b3 = torch.randn(3, device=dev) * 0.1
cw = ConvW(w)
x = torch.randn(N, 3, 256, 256, device=dev)
a = torch.randn(N, 128, 128, 64, device=dev).to(torch.bfloat16)
dw = torch.zeros_like(w)
for name, fn, bytes_ in (("first_down", lambda: ops.first_down(x, cw, b, 0.2), x.numel() * 4 + a.numel() * 2),
                         ("last_up", lambda: ops.last_up(a, cw, b3, True), x.numel() * 4 + a.numel() * 2),
                         ("skinny_wgrad", lambda: ops.skinny_wgrad(a, x, dw, False), x.numel() * 4 + a.numel() * 2)):
    t = timeit(fn)
    print(f"{name:14s} {t*1e6:8.1f} us   {bytes_/t/1e12:6.2f} TB/s algorithmic")
