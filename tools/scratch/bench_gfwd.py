import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rna_gan_amd import engine as E
dev = torch.device("cuda:0")
G, Dm, og, od, losses = bench.build(dev, "bf16", 64, 19198, 0)
ops, gn = G.runtime()
def run(n, reps):
    nz = torch.randn(n, 2048, device=dev)
    f = lambda: E.gen_forward(ops, gn, nz, keep=False)
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
a = run(64, 2); b = run(128, 1)
print("2 x batch 64: %.3f ms   1 x batch 128: %.3f ms" % (a, b))
