#!/usr/bin/env python3
# soak: 600 iterations of the benchmark's step and 300 of the Trainer loop; losses finite, device memory flat
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
args = bench.parse_args([])
dev = torch.device("cuda:0")
one_step, flush, N, info = bench.hip_workload(args, 0, 1, dev)
for _ in range(12): one_step()
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
vals = []
for i in range(600):
    ls = one_step()
    if i % 100 == 99:
        torch.cuda.synchronize(); vals.append([round(float(l), 4) for l in ls])
torch.cuda.synchronize(); m1 = torch.cuda.memory_allocated(); r1 = torch.cuda.memory_reserved()
print("600 steps: losses every 100:", vals)
print("allocated %.1f -> %.1f MB, reserved %.1f -> %.1f MB" % (m0 / 2**20, m1 / 2**20, r0 / 2**20, r1 / 2**20))
assert all(abs(x) < 1e6 and x == x for v in vals for x in v) and m1 - m0 < 64 * 2**20
h = info["handles"]
step, _ = bench.api_path_step(args, h["G"], h["D"], h["og"], h["od"], h["losses"], dev, 0)
for _ in range(12): step()
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
for i in range(300): ls = step()
torch.cuda.synchronize(); m1 = torch.cuda.memory_allocated()
print("Trainer loop 300 iterations: last losses", [round(float(l), 4) for l in ls], "allocated %.1f -> %.1f MB" % (m0 / 2**20, m1 / 2**20))
assert m1 - m0 < 64 * 2**20
print("soak OK")
