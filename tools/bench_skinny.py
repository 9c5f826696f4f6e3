#!/usr/bin/env python3
"""The three image-side row kernels at the benchmark's shape (64 x 3 x 256 x 256 <-> 64 x 128 x 128 x 64 bf16): time and
algorithmic TB/s, the general row-staged kernels (skinny128 = 0) and the control-flow-free forms (skinny128 = 1) interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rna_gan_amd import _abi
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ops = HipOps(torch.bfloat16, "cuda:0"); dev = torch.device("cuda:0")
lib = _abi.load()
def timeit(fn, rep=20):
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
img = torch.tanh(torch.randn(N, 3, 256, 256, device=dev))
dw = torch.zeros_like(w)
a0 = ops.first_down(x, cw, b, 0.2)
cases = (("first_down+bits", lambda: ops.first_down(x, cw, b, 0.2)),
         ("first_down raw", lambda: ops.first_down(x, cw, None, 1.0)),
         ("first_down tangent", lambda: ops.first_down_tangent(x, cw, a0, 0.2)),
         ("last_up tanh", lambda: ops.last_up(a, cw, b3, True)),
         ("last_up post", lambda: ops.last_up_post(a, cw, None)),
         ("last_up post tb", lambda: ops.last_up_post(a, cw, img)),
         ("skinny_wgrad", lambda: ops.skinny_wgrad(a, x, dw, False)))
bytes_ = x.numel() * 4 + a.numel() * 2
res = {}
for rnd in range(3):
    for opt in (0, 1):
        lib.rg_set_option(b"skinny128", opt)
        for name, fn in cases:
            res.setdefault((name, opt), []).append(timeit(fn))
lib.rg_set_option(b"skinny128", -1)
for name, _ in cases:
    t0, t1 = min(res[(name, 0)]), min(res[(name, 1)])
    print(f"{name:20s} general {t0*1e6:7.1f} us {bytes_/t0/1e12:5.2f} TB/s   rows128 {t1*1e6:7.1f} us {bytes_/t1/1e12:5.2f} TB/s")
