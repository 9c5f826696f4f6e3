#!/usr/bin/env python3
"""BatchNorm forward / backward timing at the small deep-layer shapes (RNAGAN_BN_FUSED=0/1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rna_gan_amd.ops_hip import HipOps
ops = HipOps(torch.bfloat16, "cuda:0")
def timeit(fn, rep=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3
for M, C in ((1024, 2048), (4096, 1024), (16384, 512), (65536, 256), (262144, 128), (1048576, 64)):
    z = torch.randn(M, 1, 1, C, device="cuda").to(torch.bfloat16).view(1, M, 1, C)
    ga = torch.randn_like(z)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    a, mean, inv = ops.bn_forward(z, g, b, 0.2, 1e-5, 0.1)
    tf = timeit(lambda: ops.bn_forward(z, g, b, 0.2, 1e-5, 0.1))
    tb = timeit(lambda: ops.bn_act_bwd(z, ga, mean, inv, g, b, 0.2))
    mean, inv = ops.bn_stats_finalize(z, 1e-5, 0.1)
    ta = timeit(lambda: ops.bn_act(z, mean, inv, g, b, 0.2))
    nb = M * C * 2
    print(f"M={M:7d} C={C:5d}  fwd(stats+apply) {tf:7.1f} us   apply only {ta:7.1f} us ({2 * nb / ta / 1e6:5.2f} TB/s)   "
          f"bwd(reduce+apply) {tb:7.1f} us ({5 * nb / tb / 1e6:5.2f} TB/s)")
