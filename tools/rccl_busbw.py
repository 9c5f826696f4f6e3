#!/usr/bin/env python3
"""Bus bandwidth of the collectives one training iteration issues, measured with torch.distributed (RCCL) on the ranks of a
torchrun launch -- part of tools/dp_first_run.sh.

Messages: the bf16 gradient tail of the generator (90 MB with G.0's gradient travelling as factors; 224 MB
without), the discriminator's 89 MB, and the all-gather of the G.0 factors (4.7 MB per rank).  Bus bandwidth as nccl-tests
defines it: all-reduce 2 (W - 1) / W x bytes / t, all-gather (W - 1) / W x total bytes / t.  DESIGN's 8-rank estimate assumed
~300 GB/s for the >= 64 MB all-reduces."""
import json
import os
import time

import torch
import torch.distributed as dist


def main():
    dist.init_process_group("nccl", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rows = []
    cases = [("allreduce bf16 generator tail (factors on)", "ar", 44_900_000, torch.bfloat16),
             ("allreduce bf16 generator whole (factors off)", "ar", 111_815_555, torch.bfloat16),
             ("allreduce bf16 discriminator", "ar", 44_739_392, torch.bfloat16),
             ("allreduce fp32 discriminator", "ar", 44_739_392, torch.float32),
             ("allgather G.0 factors gz0 (bf16, 64 x 16 x 2048 per rank)", "ag", 64 * 16 * 2048, torch.bfloat16),
             ("allgather G.0 factors z (fp32, 64 x 2048 per rank)", "ag", 64 * 2048, torch.float32)]
    for name, kind, n, dt in cases:
        if kind == "ar":
            buf = torch.ones(n, dtype=dt, device="cuda")
            fn = lambda: dist.all_reduce(buf)
            nbytes = n * buf.element_size()
            factor = 2.0 * (world - 1) / world
        else:
            mine = torch.ones(n, dtype=dt, device="cuda")
            full = torch.empty(n * world, dtype=dt, device="cuda")
            fn = lambda: dist.all_gather_into_tensor(full, mine)
            nbytes = n * world * mine.element_size()
            factor = (world - 1) / world
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        iters = 20
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / iters
        t = torch.tensor([dt_s], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
        rows.append({"collective": name, "bytes": nbytes, "us": round(dt_s * 1e6, 1),
                     "algbw_GBps": round(nbytes / dt_s / 1e9, 1), "busbw_GBps": round(factor * nbytes / dt_s / 1e9, 1)})
    if rank == 0:
        print(json.dumps({"world": world, "backend": "rccl (torch.distributed nccl)", "rows": rows}, indent=1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
