#!/usr/bin/env python3
"""A/B of the conv kernels inside ONE process (CDNA guide rule 24): for every 4x4 stride-2 layer shape of the
reference model, conv_down / conv_up (and the weight gradient with --wgrad) are run under each option set,
outputs compared with the first set (max |diff|, and against an fp32 torch convolution on a sample slice), then
timed in interleaved rounds.

  python tools/ab_conv.py [--batch 64] [--rounds 5] [--sets "conv8=0;conv8=1;conv8=1,conv8_blocks=512"] [--wgrad]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from rna_gan_amd import _abi
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--rep", type=int, default=10)
ap.add_argument("--sets", default="conv8=0;conv8=1")
ap.add_argument("--wgrad", action="store_true")
ap.add_argument("--layers", default="0,1,2,3,4")
ap.add_argument("--kinds", default="down,up")
ap.add_argument("--check", type=int, default=1)
args = ap.parse_args()

lib = _abi.load()
dev = torch.device("cuda:0")
ops = HipOps(torch.bfloat16, "cuda:0")
N = args.batch
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv) for s in args.sets.split(";")]
all_keys = sorted({k for s in sets for k in s})


def apply(s):
    for k in all_keys:
        _abi.check(lib.rg_set_option(k.encode(), s.get(k, -1)), "rg_set_option")


def timeit(fn, rep):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e-3


def ref_slice(kind, x, g, w, n_s=2):
    """fp32 torch convolution of the bf16-rounded operands on the first n_s samples (host-independent check)."""
    wf = w.to(torch.bfloat16).float()
    if kind == "down":
        xs = x[:n_s].float().permute(0, 3, 1, 2)
        return F.conv2d(xs, wf, stride=2, padding=1).permute(0, 2, 3, 1)
    gs = g[:n_s].float().permute(0, 3, 1, 2)
    return F.conv_transpose2d(gs, wf, stride=2, padding=1).permute(0, 2, 3, 1)


layers = [int(v) for v in args.layers.split(",")]
kinds = [k for k in args.kinds.split(",") if k] + (["wgrad", "wgrad2"] if args.wgrad else [])
tot = {(k, i): [0.0, 0.0] for k in kinds for i in range(len(sets))}
torch.manual_seed(0)
for l in range(5):
    c, s = 64 << l, 128 >> l
    if l not in layers:
        continue
    I, O, hs = c, 2 * c, s
    w = torch.randn(O, I, 4, 4, device=dev) * (2.0 / (I * 16)) ** 0.5
    wt = w.permute(0, 2, 3, 1).contiguous()
    cw = ConvW(wt, None, torch.zeros_like(wt), None, "OHWI")
    x = torch.randn(N, hs, hs, I, device=dev).to(torch.bfloat16)
    g = torch.randn(N, hs // 2, hs // 2, O, device=dev).to(torch.bfloat16)
    flops = 2.0 * N * (hs // 2) ** 2 * O * I * 16
    x2, g2 = x.flip(0).contiguous(), g.flip(0).contiguous()
    fns = {"down": lambda: ops.conv_down(x, cw, want_stats=True), "up": lambda: ops.conv_up(g, cw, want_stats=True),
           "wgrad": lambda: ops.conv_wgrad(g, x, cw, False), "wgrad2": lambda: ops.conv_wgrad2(g, x, g2, x2, cw, False)}
    for kind in kinds:
        fn = fns[kind]
        base = None
        notes = []
        if args.check:
            for i, st in enumerate(sets):
                apply(st)
                if kind.startswith("wgrad"):
                    fn()
                    y, stt = cw.dw.clone().float(), None
                else:
                    y, stt = fn()
                    y = y.float()
                torch.cuda.synchronize()
                if base is None:
                    base = y
                    if not kind.startswith("wgrad"):
                        r = ref_slice(kind, x, g, w)
                        err = (y[:r.shape[0]] - r).abs().max().item() / (r.abs().max().item() + 1e-9)
                        notes.append("set0 vs torch fp32 (2 samples): rel-max %.2e" % err)
                else:
                    d = (y - base).abs().max().item() / (base.abs().max().item() + 1e-9)
                    extra = ""
                    if stt is not None:
                        cs = stt.view(-1, 2, stt.shape[-1]).sum(0)
                        ref = torch.stack([base.view(-1, base.shape[-1]).sum(0), (base.view(-1, base.shape[-1]) ** 2).sum(0)])
                        extra = " stats rel %.1e" % ((cs - ref).abs().max() / ref.abs().max()).item()
                    notes.append("set%d vs set0: rel-max %.2e%s" % (i, d, extra))
        times = [[] for _ in sets]
        for i, st in enumerate(sets):      # warm-up
            apply(st)
            timeit(fn, 3)
        for _ in range(args.rounds):
            for i, st in enumerate(sets):
                apply(st)
                times[i].append(timeit(fn, args.rep))
        line = "L%d I=%4d O=%4d hi=%3d %-5s" % (l + 1, I, O, hs, kind)
        for i in range(len(sets)):
            med = sorted(times[i])[len(times[i]) // 2]
            fl = flops * (2 if kind == "wgrad2" else 1)
            tot[(kind, i)][0] += fl
            tot[(kind, i)][1] += med
            line += " | set%d %7.1f us %7.1f TF (min %6.1f)" % (i, med * 1e6, fl / med / 1e12, min(times[i]) * 1e6)
        print(line, " ; ".join(notes), flush=True)
for i, st in enumerate(sets):
    print("set%d = %s" % (i, st))
for (kind, i), (f, t) in sorted(tot.items()):
    if t > 0:
        print("%-5s set%d: total %.3f ms, %.1f TF/s" % (kind, i, t * 1e3, f / t / 1e12))
