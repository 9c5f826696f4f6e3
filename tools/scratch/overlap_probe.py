#!/usr/bin/env python3
"""Probe: do independent kernels on two HIP streams (eager, and as forked branches of one captured graph) overlap on
this device?  Chain A = conv_down layer 3 x nA (MFMA-bound), chain B = streaming fp32 add x nB (HBM-bound) or a
second conv chain / a chain of tiny kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rna_gan_amd.engine import ConvW
from rna_gan_amd.ops_hip import HipOps

dev = torch.device("cuda:0")
opsA = HipOps(torch.bfloat16, "cuda:0")
opsB = HipOps(torch.bfloat16, "cuda:0")
N = 64


def mk(l):
    c, s = 64 << l, 128 >> l
    w = torch.randn(2 * c, 4, 4, c, device=dev) * 0.02
    return ConvW(w, None, torch.zeros_like(w), None, "OHWI"), torch.randn(N, s, s, c, device=dev).to(torch.bfloat16)


cwA, xA = mk(2)
cwB, xB = mk(3)
big1 = torch.randn(100_000_000, device=dev)
big2 = torch.randn(100_000_000, device=dev)
small = torch.randn(4096, device=dev)


def chainA(n=16):
    for _ in range(n):
        opsA.conv_down(xA, cwA, want_stats=True)


def chain_conv2(n=16):
    for _ in range(n):
        opsB.conv_down(xB, cwB, want_stats=True)


def chain_stream(n=4):
    for _ in range(n):
        big1.add_(big2)


def chain_tiny(n=100):
    for _ in range(n):
        small.mul_(1.0001)


def t(fn, rep=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep


side = torch.cuda.Stream()


def both(a, b):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            b()
        a()
        cur.wait_stream(side)
    return f


def graphed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


for name, b in (("conv2", chain_conv2), ("stream", chain_stream), ("tiny", chain_tiny)):
    ta, tb = t(chainA), t(b)
    tab = t(both(chainA, b))
    ga, gb, gab = graphed(chainA), graphed(b), graphed(both(chainA, b))
    gser = graphed(lambda: (chainA(), b()))
    print("%-7s eager: A %.3f  B %.3f  A||B %.3f ms | graph: A %.3f  B %.3f  A;B %.3f  A||B %.3f ms"
          % (name, ta, tb, tab, t(ga), t(gb), t(gser), t(gab)), flush=True)
