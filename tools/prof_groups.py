#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace csv by (kernel, grid): launches, mean / min duration.  A kernel that serves several layers
shows as one line in --stats; this separates the layers (their grids differ).

    python3 tools/prof_groups.py <dir with *_kernel_trace.csv> [name filter] [top N] [--seq K]

--seq K also prints each group's first K durations in launch order (which of a layer's launches per iteration are the slow ones)."""
import csv, glob, os, sys, collections

def main():
    seq = 0
    if "--seq" in sys.argv:
        i = sys.argv.index("--seq"); seq = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    groups = collections.defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"]
                if flt and flt not in name:
                    continue
                wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
                grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]),
                        int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
                groups[(name, grid, wg)].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    order = {k: [d for _, d in sorted(v)] for k, v in groups.items()}
    groups = {k: order[k] for k in groups}
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for v in groups.values())
    print("total %.3f ms over %d groups" % (tot / 1e3, len(rows)))
    for (name, grid, wg), v in rows[:top]:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        print("%7d x %9.1f us (min %9.1f)  %6.2f%%  grid %-18s %s" % (len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot, "x".join(map(str, grid)), short[:150]))
        if seq:
            print("          in launch order: " + " ".join("%.0f" % x for x in v[-seq:]))

if __name__ == "__main__":
    main()
