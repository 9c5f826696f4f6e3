#!/usr/bin/env python3
# head weight gradient microbenchmark (N x 4 x 4 x 2048 bf16 activation)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rna_gan_amd.ops_hip import HipOps
dev = torch.device("cuda:0"); ops = HipOps(torch.bfloat16, dev)
for N in (64, 128):
    a = torch.randn(N, 4, 4, 2048, device=dev).bfloat16(); gh = torch.randn(N, device=dev); dw = torch.zeros(1, 2048, 4, 4, device=dev)
    ops.head_wgrad(gh, a, dw, False); torch.cuda.synchronize()
    ref = torch.einsum("n,nhwc->chw", gh.double(), a.double())
    err = float((dw[0].double() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.head_wgrad(gh, a, dw, False)
    e1.record(); torch.cuda.synchronize()
    print("head_wgrad N=%d: %.1f us, rel err %.1e" % (N, e0.elapsed_time(e1) / 50 * 1e3, err))
