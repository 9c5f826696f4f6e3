#!/usr/bin/env python3
"""Per-dispatch-shape summary of rocprofv3 --pmc passes (csv output directories).

Rows are keyed by (short kernel name, grid size, workgroup size): in the conv microbench (tools/ab_conv.py) and in
bench.py every conv layer has its own grid, so a row is one layer of one kernel.  For each counter the mean per
dispatch is printed; derived columns (MI355X_MICROARCH.md):
  us          mean dispatch duration from the timestamps
  mfma_util   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)      (busy cycles / elapsed cycles)
  clk_GHz     GRBM_GUI_ACTIVE / 8 / duration
  fetch_MB    2 x FETCH_SIZE (KiB -> bytes; gfx950 tallies 128-B requests at 64 B), write_MB = WRITE_SIZE
  l2_hit      TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
Usage: pmc_layers.py <dir> [<dir> ...] [--match substr] [--csv out.csv]"""
import collections
import csv
import glob
import re
import sys

dirs, match, out_csv = [], None, None
it = iter(sys.argv[1:])
for a in it:
    if a == "--match":
        match = next(it)
    elif a == "--csv":
        out_csv = next(it)
    else:
        dirs.append(a)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


rows = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if match and match not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            c = rows[key][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
            did = (f, r["Dispatch_Id"])
            if did not in seen:
                seen.add(did)
                dur[key][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                dur[key][1] += 1

table = []
for key in sorted(rows, key=lambda k: (k[0], k[1])):
    cs = {k: v[0] / max(v[1], 1) for k, v in rows[key].items()}
    us = dur[key][0] / max(dur[key][1], 1)
    row = {"kernel": key[0], "grid": key[1], "wg": key[2], "n": dur[key][1], "us": round(us, 1)}
    if "GRBM_GUI_ACTIVE" in cs:
        cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
        row["clk_GHz"] = round(cyc / (us * 1e3), 2) if us > 0 else None
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            row["mfma_util"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc), 3)
    if "SQ_WAVE_CYCLES" in cs:
        wc = cs["SQ_WAVE_CYCLES"]
        for k, nm in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active"),
                      ("SQ_WAIT_INST_LDS", "wait_lds")):
            if k in cs:
                row[nm] = round(cs[k] / wc, 3)
    if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs and cs["SQ_LDS_IDX_ACTIVE"] > 0:
        row["lds_conflict"] = round(cs["SQ_LDS_BANK_CONFLICT"] / cs["SQ_LDS_IDX_ACTIVE"], 3)
    if "FETCH_SIZE" in cs:
        row["fetch_MB"] = round(2.0 * cs["FETCH_SIZE"] * 1024 / 1e6, 1)
    if "WRITE_SIZE" in cs:
        row["write_MB"] = round(cs["WRITE_SIZE"] * 1024 / 1e6, 1)
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        row["l2_hit"] = round(cs["TCC_HIT_sum"] / max(cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"], 1.0), 3)
    table.append(row)

cols = ["kernel", "grid", "wg", "n", "us", "clk_GHz", "mfma_util", "wait_any", "wait_inst", "active", "wait_lds",
        "lds_conflict", "fetch_MB", "write_MB", "l2_hit"]
cols = [c for c in cols if any(c in r for r in table)]
w = csv.writer(sys.stdout)
w.writerow(cols)
for r in table:
    w.writerow([r.get(c, "") for c in cols])
if out_csv:
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for r in table:
            w.writerow([r.get(c, "") for c in cols])
