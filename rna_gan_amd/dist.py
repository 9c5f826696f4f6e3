"""Data-parallel glue: one process per GPU, gradients summed with RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI.  The reference has no distributed path; semantics are plain
DDP: rank-local BatchNorm statistics, rank-local latent standardisation / eps / penalty norm, and
the gradient of the MEAN over ranks of the rank-local losses.

* The 1/world scaling is folded into the backward seed (engine ``grad_scale``), so the collective is
  a pure SUM over the flat gradient buffer: no extra pass over the gradients.
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): the all-reduce is bandwidth-bound per link, so
  the volume is what matters.  In bf16 precision mode the flat fp32 gradient is compressed to bf16 for
  the wire (805 MB -> 403 MB per iteration and rank; rna_gan_amd.optim.Adam steps straight from the wire buffer);
  the MFMA operands that produced it were bf16 already (DESIGN "Numerics").  fp32 mode sends fp32.
* The generator's layer-0 weight gradient (60 % of its parameters) is a rank-(batch) product: its FACTORS are all-gathered
  (4.7 MB per rank) and the product is formed over all ranks' samples inside the fused Adam step (G0_FACTORS below):
  403 -> 269 MB on the wire per iteration and rank.
* Collectives are never captured into HIP graphs: for world > 1 each train_op is graph(prefix) / graph(rest) /
  eager all-reduce start, and the wait + optimizer-step graph are issued by the NEXT train_op after its prefix
  (which reads the other network), so the transfer overlaps with compute (losses._Runner.run_dp).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# One collective per gradient buffer (G: 224 MB, D: 89 MB on the bf16 wire): the all-reduce of a train_op is started
# once its whole backward is done (90 % of the generator's gradient bytes are written by the last two launches of the
# backward, so buckets in backward order would have nothing to overlap with), and every extra RCCL launch costs
# 20-50 us per rank.  RNAGAN_DP_BUCKET_MB splits the buffer into fixed-size buckets for experiments.
BUCKET_BYTES = int(os.environ.get("RNAGAN_DP_BUCKET_MB", "1024")) * 1024 * 1024
FORCE = os.environ.get("RNAGAN_FORCE_DP", "0") == "1"     # take the DP code path even with one rank (testing)
COMPRESS = os.environ.get("RNAGAN_DP_BF16", "1") != "0"
# fp16 build: the gradients carry the static loss scale (x 4096) and a weight gradient of 16 / world ends fp16's range -- the
# critic head's penalty-step gradient reached 48 in tests/test_dp2_gpu.py and went onto an fp16 wire as inf.  So the fp16 build
# all-reduces fp32 by default (what torch AMP + DistributedDataParallel does: fp32 gradients of fp32 master weights);
# RNAGAN_DP_F16_WIRE=1 opts into the 16-bit wire (half the bytes, no overflow guard).
F16_WIRE = os.environ.get("RNAGAN_DP_F16_WIRE", "0") == "1"
# --sync-stats: BatchNorm statistics (forward, backward, tangent, double backward), the latent
# standardisation and the penalty norm are taken over the GLOBAL batch (tiny all-reduces of per-channel sums), so an
# N-rank run reproduces the single-process reference step at batch N x n exactly.  Collectives then sit inside the
# step: HIP graphs are not used in this mode.
SYNC_STATS = os.environ.get("RNAGAN_SYNC_STATS", "0") == "1"


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def active() -> bool:
    """True when gradients have to be all-reduced."""
    return world_size() > 1 or (FORCE and dist.is_available() and dist.is_initialized())


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (no-op for a single process)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if (ws <= 1 and not FORCE) or (dist.is_available() and dist.is_initialized()):
        return
    if "RANK" not in os.environ:
        return
    if backend is None:
        # RNAGAN_DIST_BACKEND=gloo: functional runs of the multi-rank path where RCCL cannot be used -- several ranks SHARING
        # one device (RCCL refuses duplicate devices; gloo stages device tensors through the host: correct, not fast)
        backend = os.environ.get("RNAGAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            # RCCL shares buffers between the rank processes through dmabuf IPC on this driver stack; the HSA runtime reads the
            # variable when HIP initialises, i.e. it has to be in the environment BEFORE the first torch.cuda call of the
            # process (bench.py and the CLI export it themselves; a user script under torchrun must do the same)
            import warnings
            warnings.warn("rna_gan_amd.dist: HSA_ENABLE_IPC_MODE_LEGACY=0 is not exported; RCCL's inter-process buffer "
                          "sharing fails with 'hipIpcGetMemHandle: invalid argument' on hosts that only support dmabuf IPC "
                          "(export it before python starts)")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")


def set_sync_stats(on: bool):
    """Switch --sync-stats on/off programmatically (before the modules' runtimes are built)."""
    global SYNC_STATS
    SYNC_STATS = bool(on)


def sync_stats() -> bool:
    return SYNC_STATS and active()


def allreduce_small_(t: torch.Tensor):
    """Blocking in-place SUM all-reduce of a small statistics tensor."""
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def attach_sync(ops):
    """Give an ops object (HipOps / the CPU twin) the global-statistics hooks when --sync-stats is active."""
    if sync_stats():
        ops.stat_reduce = allreduce_small_
        ops.stat_world = world_size()
    return ops


def grad_scale() -> float:
    return 1.0 / world_size()


def gp_grad_scale() -> float:
    """Backward seed of the gradient penalty: the mean of per-rank penalties (plain DDP semantics) or, with
    synchronised statistics, ONE whole-batch penalty whose per-rank gradient contributions simply add up."""
    return 1.0 if sync_stats() else 1.0 / world_size()


_wire = {}
# Generator layer 0's weight gradient is a rank-(batch) product dW = z^T gz0 (K = the batch): the ranks all-gather its
# FACTORS (per rank 64 x 2048 fp32 + 64 x 32768 bf16 = 4.7 MB) instead of all-reducing the 67 M-element product (134 MB on the
# bf16 wire, 60 % of the generator's collective), and every rank forms sum_r z_r^T gz0_r inside its fused Adam step
# (rg_g0_wgrad_adam with K = world x batch, fp32 accumulation over ALL ranks' samples -- more accurate than a bf16 ring sum).
G0_FACTORS = os.environ.get("RNAGAN_DP_G0_FACTORS", "1") != "0"
_factors = {}


def factor_buffers(key, n, E, C, gy_dtype, device):
    """Persistent gathered-factor buffers of one module: (z_all [W n, E] fp32, gy_all [W n, 4, 4, C]) and this rank's slices."""
    W, r = world_size(), rank()
    k = (key, n, E, C, gy_dtype, device, W)
    buf = _factors.get(k)
    if buf is None:
        z_all = torch.zeros(W * n, E, dtype=torch.float32, device=device)
        gy_all = torch.zeros(W * n, 4, 4, C, dtype=gy_dtype, device=device)
        buf = _factors[k] = (z_all, gy_all, z_all[r * n:(r + 1) * n], gy_all[r * n:(r + 1) * n])
    return buf


def allgather_start(full: torch.Tensor, mine: torch.Tensor):
    """Start the all-gather of per-rank row blocks into ``full`` (``mine`` = this rank's block, a slice of ``full``)."""
    if world_size() == 1:
        return None
    try:
        return dist.all_gather_into_tensor(full, mine, async_op=True)
    except (RuntimeError, NotImplementedError):          # a backend without the flat form: per-rank views
        return dist.all_gather(list(full.chunk(world_size(), dim=0)), mine.clone(), async_op=True)


# RNAGAN_DEBUG_HOG="blocks,microseconds": measurement aid for a single-GPU box -- where an all-reduce would run, a kernel that
# holds `blocks` CUs for `microseconds` is launched on its own stream (behind the gradients, waited for by allreduce_finish as
# the collective's work object would be): the CU footprint and duration of a collective without a second GPU.  DESIGN 12.7.
_HOG = [None]


def _hog_start(device):
    spec = os.environ.get("RNAGAN_DEBUG_HOG")
    if not spec:
        return None
    import ctypes
    blocks, usec = [int(v) for v in spec.split(",")]
    if _HOG[0] is None:
        # the stand-in kernel lives in a diagnostic library of its own (tools/debug/), not in the product ABI
        from .build import build_debug_library
        dbg = ctypes.CDLL(build_debug_library())
        dbg.rgdbg_hold_cus.restype = ctypes.c_int
        dbg.rgdbg_hold_cus.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _HOG[0] = (torch.cuda.Stream(device), torch.zeros(4, device=device), dbg)
    stream, sink, dbg = _HOG[0]
    stream.wait_stream(torch.cuda.current_stream(device))
    rc = dbg.rgdbg_hold_cus(blocks, usec, sink.data_ptr(), stream.cuda_stream)
    if rc != 0:
        raise RuntimeError("rgdbg_hold_cus failed: %d" % rc)
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


class _HogWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        if self.ev is not None:
            torch.cuda.current_stream().wait_event(self.ev)


def _half_of(compress):
    """compress argument -> "bf16" / "f16" (the build of the library whose 16-bit type the wire carries) or None."""
    if not compress:
        return None
    return "f16" if compress == "f16" else "bf16"


def wire_for(flat: torch.Tensor, half: str = "bf16"):
    """The persistent 16-bit wire buffer (bf16, or fp16 for the fp16 build) of a flat fp32 gradient buffer (one per device,
    length and type), or None when the gradients travel as fp32."""
    if not (COMPRESS and flat.is_cuda and flat.dtype == torch.float32):
        return None
    key = (flat.device, flat.numel(), flat.data_ptr(), half)
    wire = _wire.get(key)
    if wire is None:
        wire = _wire[key] = torch.empty(flat.numel(), dtype=torch.float16 if half == "f16" else torch.bfloat16,
                                        device=flat.device)
    return wire


def allreduce_start(flat: torch.Tensor, compress=False, head: int = 0, table=None):
    """Start the in-place SUM all-reduce of a flat fp32 gradient buffer (fixed-size buckets, asynchronous: RCCL runs
    on its own stream behind everything enqueued so far on the current one).  Returns a handle for
    allreduce_finish(); None when there is nothing to reduce.  `flat` must not be written before the finish.
    head: the first `head` elements are NOT reduced (their gradient travels as gathered factors, see G0_FACTORS).
    table (bf16 wire only): segment table over ``[head:]`` -- rows (offset, length, slab address or 0, nsplit, slab dtype) as for
    rg_adam_step_slabs -- of layers whose weight gradient was NOT reduced into ``flat`` in this pass: split-K slabs are summed
    straight onto the wire, nsplit = -1 marks a slice its weight-gradient launch already wrote there (rg_grad_to_wire)."""
    if not active():
        return None
    n = flat.numel()
    if compress and COMPRESS and flat.is_cuda and flat.dtype == torch.float32:
        from . import _abi
        half = _half_of(compress)
        lib = _abi.load(half)
        wire = wire_for(flat, half)
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        if table:
            import ctypes as C
            k = len(table)
            offs = (C.c_ulonglong * k)(*[t[0] for t in table])
            lens = (C.c_ulonglong * k)(*[t[1] for t in table])
            slabs = (C.c_void_p * k)(*[t[2] or None for t in table])
            nsp = (C.c_int * k)(*[t[3] for t in table])
            sdts = (C.c_int * k)(*[t[4] for t in table])
            _abi.check(lib.rg_grad_to_wire(flat.data_ptr() + 4 * head, wire.data_ptr() + 2 * head, n - head, k, C.addressof(offs),
                                           C.addressof(lens), C.addressof(slabs), C.addressof(nsp), C.addressof(sdts), stream),
                       "rg_grad_to_wire")
        else:
            _abi.check(lib.rg_cast_pad(flat.data_ptr() + 4 * head, wire.data_ptr() + 2 * head, 1, n - head, n - head,
                                       _abi.RG_F16 if half == "f16" else _abi.RG_BF16, stream), "rg_cast_pad")
        works = _launch_buckets(wire[head:], BUCKET_BYTES // 2)
        hog = _hog_start(flat.device) if flat.is_cuda else None
        return (wire, flat, works + ([_HogWork(hog)] if hog is not None else []), head)
    works = _launch_buckets(flat[head:], BUCKET_BYTES // flat.element_size())
    hog = _hog_start(flat.device) if flat.is_cuda else None
    return (None, flat, works + ([_HogWork(hog)] if hog is not None else []), head)


def allreduce_finish(handle, widen=True):
    """Make the current stream wait for the all-reduce and widen the bf16 wire buffer back into the fp32 gradients
    (widen=False: the caller's optimizer reads the wire buffer itself, see wire_of).  Only the reduced part
    ``[head:]`` is widened: the first ``head`` elements never travelled on the wire (allreduce_start's ``head``)."""
    if handle is None:
        return
    wire, flat, works, head = handle
    for w in works:
        w.wait()
    if wire is not None and widen and flat.numel() > head:
        from . import _abi
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        _abi.check(_abi.load("f16" if wire.dtype == torch.float16 else "bf16").rg_widen_bf16(wire.data_ptr() + 2 * head, flat.data_ptr() + 4 * head, flat.numel() - head, stream),
                   "rg_widen_bf16")


def wire_of(handle):
    """The bf16 wire buffer of a compressed all-reduce (None for an fp32 one)."""
    return None if handle is None else handle[0]


def allreduce_sum_(flat: torch.Tensor, compress: bool = False):
    """In-place SUM all-reduce of a flat fp32 gradient buffer, in fixed-size buckets."""
    allreduce_finish(allreduce_start(flat, compress))


def _launch_buckets(t: torch.Tensor, elems: int):
    works = []
    n = t.numel()
    for off in range(0, n, elems):
        works.append(dist.all_reduce(t[off:min(n, off + elems)], op=dist.ReduceOp.SUM, async_op=True))
    return works


_flush_hook = [None]


def set_flush_hook(fn):
    _flush_hook[0] = fn


def flush():
    """Apply an optimizer step that a data-parallel train_op left in flight (losses.flush); no-op otherwise."""
    if _flush_hook[0] is not None:
        _flush_hook[0]()


def broadcast_(t: torch.Tensor, src: int = 0):
    if world_size() > 1:
        dist.broadcast(t, src=src)
