#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (csv output) and group them by family."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", round(tot / 1e6, 3), "file", f)
fam = {}
for r in rows:
    n = r["Name"]
    key = ("gather" if ("gather_gemm" in n or "conv8_kernel" in n or "convp_kernel" in n or "convd_kernel" in n) else "wgrad" if "wgrad" in n else "bn" if ("rowreduce" in n or "rowapply" in n or "colfinish" in n or "bn_" in n or "slab_bn" in n)
           else "reduce" if "reduce_" in n else "pack" if ("pack" in n or "widen" in n) else "adam" if "adam" in n.lower()
           else "image-side" if ("first_down" in n or "last_up" in n or "skinny" in n) else "other")
    fam[key] = fam.get(key, 0.0) + float(r["TotalDurationNs"])
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print("  family %-12s %9.3f ms %6.2f%%" % (k, v / 1e6, 100 * v / tot))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-100s %6s %9.3f ms %6.2f%% avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
