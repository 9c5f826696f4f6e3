#!/usr/bin/env python3
# rg_g0_wgrad_adam at the reference generator's size (E = C = 2048, 67 M parameters) for K = 64 .. 512 samples
# (K = world x 64 when the data-parallel path hands over gathered factors)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rna_gan_amd import _abi
lib = _abi.load(); dev = torch.device("cuda:0")
E = C = 2048
p = torch.randn(E, C, 4, 4, device=dev) * 0.02; m = torch.zeros_like(p); v = torch.zeros_like(p)
sh = torch.zeros(E, C, 4, 4, dtype=torch.bfloat16, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev); hyper = torch.zeros(8, device=dev)
st = torch.cuda.current_stream().cuda_stream
_abi.check(lib.rg_adam_hyper_dev(step.data_ptr(), 1e-4, 0.5, 0.999, 1e-8, 0.0, hyper.data_ptr(), st), "hyper")
for N in (64, 128, 256, 512):
    z = torch.randn(N, E, device=dev); gy = (torch.randn(N, 4, 4, C, device=dev) * 0.05).bfloat16()
    f = lambda: _abi.check(lib.rg_g0_wgrad_adam(z.data_ptr(), gy.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr(), hyper.data_ptr(),
                                               sh.data_ptr(), N, E, C, _abi.RG_BF16, st), "g0adam")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("K = %3d samples: %.1f us  (%.2f TB/s of the 26 B / parameter stream)" % (N, us, E * C * 16 * 26 / us / 1e6))
