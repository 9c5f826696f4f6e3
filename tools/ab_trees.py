#!/usr/bin/env python3
"""Interleaved A/B of whole source trees on one box: `bench.py --no-cpu-baseline --no-roofline --no-extras` of each tree in
turn, R rounds (A B A B ...), one fresh process per run.  For changes that cannot be switched inside one process (kernels
changed at build time, ABI changes): e.g. the round-4 tree (git archive of the previous round's HEAD, library built in place)
against the working tree.  tools/ab_step.py is the in-process form for run-time options.

    python tools/ab_trees.py --trees r4=tools/scratch/_r4tree,head=. --rounds 5 --steps 40 [--json out.json]
"""
import argparse
import json
import os
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--trees", required=True)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--json", default=None)
ap.add_argument("--cleanup", action="store_true",
                help="afterwards remove every tree that lives under tools/scratch/ (extracted copies of other commits: they are not "
                     "source of this tree and inflate line counts / the gpurun snapshot)")
a = ap.parse_args()
trees = [t.split("=", 1) for t in a.trees.split(",")]
res = {n: [] for n, _ in trees}
for r in range(a.rounds):
    for name, path in (trees if r % 2 == 0 else trees[::-1]):
        path = os.path.abspath(path)
        out = subprocess.run([sys.executable, os.path.join(path, "bench.py"), "--no-cpu-baseline", "--no-roofline", "--no-extras",
                              "--steps", str(a.steps)], cwd=path, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print("[ab_trees] %s failed (rc %d)" % (name, out.returncode), file=sys.stderr)
            sys.exit(1)
        res[name].append(json.loads(line[-1])["ms_per_step"])
    print("[ab_trees] round %d: %s" % (r, "  ".join("%s %.3f" % (n, res[n][-1]) for n, _ in trees)), file=sys.stderr, flush=True)
base = trees[0][0]
summary = {"steps": a.steps, "rounds": a.rounds, "unit": "ms per iteration", "trees": []}
for n, p in trees:
    d = [x - y for x, y in zip(res[n], res[base])]
    summary["trees"].append({"name": n, "path": p, "ms": res[n], "mean": round(sum(res[n]) / len(res[n]), 3), "min": min(res[n]),
                             "delta_vs_first": {"mean": round(sum(d) / len(d), 3), "min": round(min(d), 3), "max": round(max(d), 3)}})
line = json.dumps(summary)
print(line)
if a.json:
    open(a.json, "w").write(line + "\n")

if a.cleanup:
    import shutil
    scratch = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch") + os.sep
    for _, p in trees:
        p = os.path.abspath(p) + os.sep
        if p.startswith(scratch) and p != scratch:
            shutil.rmtree(p, ignore_errors=True)
            print("[ab_trees] removed %s" % p, file=sys.stderr)
