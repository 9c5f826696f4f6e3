"""On-box ceilings for the roofline object of bench.py (SURVEY 8d / BASELINE.md 4: every fraction "against both nominal and
measured peak (MFMA-loop and stream-copy microbenchmarks, re-measured on the box)").

Measurement kernels of librnagan_hip.so (rna_gan_amd/csrc/rg_probe.hip), timed here with HIP events on the current stream.
Protocol (MI355X_MICROARCH.md, "DVFS give-back" items 6 and 7): random operands, >= `settle_s` seconds of back-to-back launches
of the SAME kernel before the timed launches, the timed launches back to back behind them.  Not a product path: nothing
under rna_gan_amd/ calls this module except bench.py and tools/.
"""
from __future__ import annotations

import ctypes
import time

import torch

from . import _abi
from ._abi import check


def _timed(launch, settle_s: float, n_timed: int, device):
    """Run `launch` back to back for about settle_s seconds, then time n_timed more launches with one event pair."""
    stream = torch.cuda.current_stream(device)
    launch(); torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); launch(); e1.record(stream); torch.cuda.synchronize(device)
    one_ms = max(e0.elapsed_time(e1), 1e-3)
    n_settle = max(int(settle_s * 1e3 / one_ms), 1)
    t0 = time.perf_counter()
    for i in range(n_settle):
        launch()
        if i % 64 == 63 and time.perf_counter() - t0 > 2.0 * settle_s + 1.0:    # (the queue must not run away from the host)
            break
    e0.record(stream)
    for _ in range(n_timed):
        launch()
    e1.record(stream)
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / n_timed


def measure_ceilings(device, settle_s: float = 2.0, n_timed: int = 8, copy_mb: int = 1024):
    """{name: value}: TFLOP/s of the bare bf16 MFMA loops (both shapes, one and two waves per SIMD), TFLOP/s of the product's
    8-wave conv k-loop fed from LDS-resident stages (both shapes), GB/s of a float4 stream copy (read + write bytes)."""
    lib = _abi.load()
    st = torch.cuda.current_stream(device).cuda_stream
    out = {"protocol": "random operands; %.1f s of back-to-back launches of the same kernel, then %d timed launches (one HIP "
                       "event pair on the launch stream)" % (settle_s, n_timed)}
    scratch = torch.zeros(64, dtype=torch.float32, device=device)
    fl = ctypes.c_double(0.0)
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    for shape in (16, 32):
        for wps in (1, 2):
            iters = 40000 // wps          # ~20-40 ms per launch
            def launch(shape=shape, wps=wps, iters=iters):
                check(lib.rg_probe_mfma_bare(shape, wps, ncu, iters, scratch.data_ptr(), ctypes.addressof(fl), st),
                      "rg_probe_mfma_bare")
            ms = _timed(launch, settle_s, n_timed, device)
            out["mfma_bare_%s_%dwave_tflops" % ("16x16x32" if shape == 16 else "32x32x16", wps)] = round(fl.value / (ms * 1e-3) / 1e12, 1)
    a = torch.empty((ncu * 256, 128), dtype=torch.bfloat16, device=device)
    b = torch.empty((256, 128), dtype=torch.bfloat16, device=device)
    c = torch.empty((ncu * 256, 256), dtype=torch.bfloat16, device=device)
    check(lib.rg_probe_fill_bf16(a.data_ptr(), a.numel(), 11, st), "rg_probe_fill_bf16")
    check(lib.rg_probe_fill_bf16(b.data_ptr(), b.numel(), 23, st), "rg_probe_fill_bf16")
    for shape in (16, 32):
        iters = 16384                      # k-tiles per workgroup: ~15-25 ms per launch
        def launch(shape=shape, iters=iters):
            check(lib.rg_probe_lds_mfma(shape, ncu, iters, a.data_ptr(), b.data_ptr(), c.data_ptr(), ctypes.addressof(fl), st),
                  "rg_probe_lds_mfma")
        ms = _timed(launch, settle_s, n_timed, device)
        out["conv8_loop_lds_fed_%s_tflops" % ("16x16x32" if shape == 16 else "32x32x16")] = round(fl.value / (ms * 1e-3) / 1e12, 1)
    del a, b, c
    nbytes = copy_mb << 20
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    check(lib.rg_probe_fill_bf16(src.data_ptr(), nbytes // 2, 5, st), "rg_probe_fill_bf16")

    best = None
    for variant, blocks in ((0, 2048), (1, 2048), (0, 4096), (1, 4096)):
        def launch_copy(variant=variant, blocks=blocks):
            check(lib.rg_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, variant, blocks, st), "rg_probe_copy")
        ms = _timed(launch_copy, min(settle_s, 0.5), n_timed, device)
        gbps = round(2.0 * nbytes / (ms * 1e-3) / 1e9, 1)
        out["stream_copy_%s_%d_gbps" % ("nt" if variant else "plain", blocks)] = gbps
        best = gbps if best is None else max(best, gbps)
    out["stream_copy_gbps"] = best
    out["stream_copy_what"] = ("float4 copy of %d MiB (read + write bytes counted; best of plain / non-temporal at 2048 / 4096 "
                               "workgroups of 256 threads, 8 loads in flight per thread)" % copy_mb)
    out["compute_units"] = ncu
    return out


if __name__ == "__main__":
    import json
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(measure_ceilings(dev), indent=1))
