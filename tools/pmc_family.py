#!/usr/bin/env python3
"""Time-weighted MFMA utilisation per kernel family from a tools/pmc_layers.py csv (MFMA-busy cycles / elapsed cycles,
weighted by n x us).  Usage: pmc_family.py <layers.csv>  -> JSON"""
import csv, json, sys

fam_of = lambda n: ("gather_gemm" if ("gather_gemm_dma_kernel<0" in n or "gather_gemm_dma_kernel<1" in n or "conv8" in n or "convp_kernel" in n or "convd_kernel" in n)
                    else "wgrad_dma" if ("wgrad8_kernel" in n or "wgrad8n_kernel" in n or "wgrad_dma_kernel" in n) else None)   # the MFMA conv weight gradients
# (g0_wgrad_adam_kernel is an HBM-streaming Adam pass with 16 MFMAs per tile, head_wgrad a VALU reduction: neither belongs here)
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    f = fam_of(r["kernel"])
    if f is None or not r.get("mfma_util"):
        continue
    w = float(r["n"]) * float(r["us"])
    a = acc.setdefault(f, {"w": 0.0, "u": 0.0, "kernels": {}})
    a["w"] += w
    a["u"] += w * float(r["mfma_util"])
    a["kernels"]["%s grid %s" % (r["kernel"], r["grid"])] = {"launches": int(r["n"]), "avg_us": float(r["us"]),
                                                               "mfma_util": float(r["mfma_util"]),
                                                               "clk_GHz": float(r["clk_GHz"]) if r.get("clk_GHz") else None}
print(json.dumps({f: {"mfma_util": round(a["u"] / a["w"], 4), "kernels": a["kernels"]} for f, a in acc.items()}, indent=1))
