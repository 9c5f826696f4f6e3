"""Data-parallel glue: one process per GPU, gradients summed with RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI.  The reference has no distributed path; semantics are plain
DDP: rank-local BatchNorm statistics, rank-local latent standardisation / eps / penalty norm, and
the gradient of the MEAN over ranks of the rank-local losses.

The 1/world scaling is folded into the backward seed (engine ``grad_scale``), so the collective is
a pure SUM over large flat fp32 buckets: no extra pass over the gradients.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# 64 MiB fp32 buckets: large enough to be bandwidth- rather than latency-bound on xGMI, small enough
# that the first bucket can start while the later ones are still being produced
BUCKET_ELEMS = 16 * 1024 * 1024


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (no-op for a single process)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")


def grad_scale() -> float:
    return 1.0 / world_size()


def allreduce_sum_(flat: torch.Tensor):
    """In-place SUM all-reduce of a flat gradient buffer in fixed-size buckets (async, then wait)."""
    if world_size() == 1:
        return
    works = []
    n = flat.numel()
    for off in range(0, n, BUCKET_ELEMS):
        works.append(dist.all_reduce(flat[off:min(n, off + BUCKET_ELEMS)], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()


def broadcast_(t: torch.Tensor, src: int = 0):
    if world_size() > 1:
        dist.broadcast(t, src=src)
