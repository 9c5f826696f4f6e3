#!/usr/bin/env python3
"""The bf16 / fp32 parity tolerances as distributions over UNSELECTED inputs: N consecutive seeds at
32 x 32 and 64 x 64, one iteration each on the HIP path in both precisions against the CPU oracle -- loss errors
|hip - oracle| / (|oracle| + 0.1) and the update-cosine gate.  Writes the lines DESIGN 14.6 quotes.

    python tools/tolerance_stats.py [--seeds 200] [--precisions fp32,bf16,fp16] [--out profiles/round6_tolerance_statistics.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=200)
ap.add_argument("--sizes", default="32,64")
ap.add_argument("--out", default=None)
ap.add_argument("--precisions", default="fp32,bf16", help="comma list of fp32, bf16, fp16")
a = ap.parse_args()
import test_train_gpu as T      # noqa: E402  (the statistics helper lives next to the test that asserts its percentiles)

lines = []
for size in [int(x) for x in a.sizes.split(",")]:
    seeds = list(range(1001, 1001 + a.seeds))
    precs = tuple(a.precisions.split(","))
    errs, coss = T._loss_and_update_statistics(size, seeds, precisions=precs)
    for precision in precs:
        lines.append(T._describe("%s %dx%d batch 8" % (precision, size, size), errs[precision], coss[precision], seeds))
if a.out:
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
