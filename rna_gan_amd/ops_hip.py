"""HipOps: the product op backend -- thin tensor->pointer marshalling onto the C ABI
(include/rnagan_hip.h).  PyTorch is used for device memory (caching allocator) and the stream
only; every arithmetic op below is one or two hand-written HIP kernels.  No fallback paths.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from contextlib import contextmanager

import torch

from . import _abi
from ._abi import RG_BF16, RG_F16, RG_F32, check
from .engine import ConvW


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class SlabRef:
    """The fp32 split-K slabs a deferred conv launch left in the workspace (conv_down / conv_up with defer=groups): attached
    to the (not yet written) output tensor as ``t._rg_slabs`` and consumed by the BatchNorm op that follows, which sums the
    slabs itself and writes the tensor (rg_bn_forward_slabs / rg_bn_act_bwd_slabs).  Holds the workspace tensor alive."""

    __slots__ = ("ws", "nsplit", "stride", "groups", "dtype")

    def __init__(self, ws, nsplit, stride, groups, dtype=RG_F32):
        self.ws, self.nsplit, self.stride, self.groups, self.dtype = ws, nsplit, stride, groups, dtype


_LIVE_OPS = weakref.WeakSet()


def _handoff_flag(device=None):
    """int32[1] device tensor = the largest error word of the live backends (optionally of one device), or None when no
    backend has launched a fused split-K BatchNorm kernel yet.  No host sync."""
    words = [ops._sb_sync[0:1] for ops in list(_LIVE_OPS)
             if ops._sb_sync is not None and (device is None or ops.device == device)]
    if not words:
        return None
    flag = words[0].clone()
    for w in words[1:]:
        flag = torch.maximum(flag, w.to(flag.device))
    return flag


def _rearm():
    for ops in list(_LIVE_OPS):
        if ops._sb_sync is not None:
            ops._sb_sync[0].zero_()


_HANDOFF_MSG = ("rna_gan_amd: a fused split-K BatchNorm launch timed out waiting for its workgroups%s: the results of that "
                "launch are invalid.  The kernels need all their <= 256 workgroups co-resident; another kernel (a "
                "collective?) held CUs.  Set RNAGAN_SPLIT_BN_DP=0 / RNAGAN_SPLIT_BN=0 to use the separate launches.")


def check_handoffs():
    """Host-side check of the fused split-K BatchNorm kernels' error word (rg_splitbn.hip: a workgroup whose bounded spin
    on the in-launch rendezvous timed out sets sync[SB_ERR] and goes on with incomplete sums -- the launch's statistics and
    activations are then garbage).  A HOST SYNC, so it is called where the host waits anyway (Trainer.save_model at the end
    of an epoch; bench.py behind the timed region; the tests); inside an epoch the Trainer polls the same word every
    ``handoff_check_every`` iterations without a sync (handoffs_poll_start / handoffs_poll_finish).
    RANK-COLLECTIVE under data parallelism: the word is MAX-all-reduced, so EVERY rank raises when ANY rank's launch timed
    out -- the gradient all-reduce has already spread that rank's garbage, rank 0 must not checkpoint it, and a rank that
    raised alone would leave the others hanging in their next collective.  Every rank must therefore call it at the same
    point (save_model, bench.py and the poll do).  Raises and re-arms."""
    from . import dist as D_
    flag = _handoff_flag()
    if D_.active():
        if flag is None:                     # no fused launch on this rank yet: still take part in the collective
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
    if flag is not None and int(flag.item()) != 0:
        _rearm()
        raise RuntimeError(_HANDOFF_MSG % (" on at least one rank" if D_.active() else ""))


def handoffs_poll_start():
    """The check above without the host sync: enqueue (rank-collective under data parallelism) the MAX of the error words
    and its copy into a pinned host slot; returns the handle handoffs_poll_finish takes an iteration or more later."""
    from . import dist as D_
    flag = _handoff_flag()
    if flag is None:
        if not D_.active() or not torch.cuda.is_available():
            return None
        flag = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    if D_.active():
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
    slot = torch.empty(1, dtype=torch.int32, pin_memory=flag.is_cuda)
    slot.copy_(flag, non_blocking=True)
    ev = None
    if flag.is_cuda:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flag.device))
    return slot, ev, flag


def handoffs_poll_finish(handle):
    if handle is None:
        return
    from . import dist as D_
    slot, ev, _flag = handle
    if ev is not None:
        ev.synchronize()
    if int(slot.item()) != 0:
        _rearm()
        raise RuntimeError(_HANDOFF_MSG % (" on at least one rank" if D_.active() else ""))


class HipOps:
    name = "hip"

    def __init__(self, act_dtype=torch.bfloat16, device="cuda:0", algo=_abi.ALGO_AUTO):
        if not torch.cuda.is_available():
            raise RuntimeError("rna_gan_amd.HipOps needs a ROCm GPU (no CPU fallback exists)")
        if act_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("act_dtype must be torch.float32, torch.bfloat16 or torch.float16")
        # torch.float16 selects the fp16 BUILD of the library (librnagan_hip_f16.so: the same kernels with IEEE fp16 as the
        # 16-bit storage type, BASELINE configs[3]); its callers pass RG_F16 where the bf16 build takes RG_BF16
        self.half = "f16" if act_dtype == torch.float16 else "bf16"
        self.lib = _abi.load(self.half)
        self.device = torch.device(device)
        self.act_dtype = act_dtype
        self.h16 = torch.float16 if self.half == "f16" else torch.bfloat16       # torch dtype of the build's 16-bit type
        self.H16 = RG_F16 if self.half == "f16" else RG_BF16                      # ... and its dtype code
        self.dt = RG_F32 if act_dtype == torch.float32 else self.H16
        # loss scaling (fp16 only; src/betaVAE.py:184,230-236 is the reference authors' commented-out GradScaler): the backward
        # seeds of the three train_ops carry loss_scale, the penalty's first backward gp_seed_scale and its tangent direction
        # gp_tangent_scale with gp_seed_scale * gp_tangent_scale == loss_scale, so EVERY parameter gradient of a step arrives
        # scaled by loss_scale and rna_gan_amd.optim.Adam unscales inside its kernels (hyper[8]).  Powers of two: exact.
        self.loss_scale = 1.0
        self.gp_seed_scale = 1.0
        self.gp_tangent_scale = 1.0
        if self.half == "f16":
            ls = float(os.environ.get("RNAGAN_F16_LOSS_SCALE", "4096"))
            rt = ls ** 0.5
            if ls < 1.0 or rt != int(rt) or (int(rt) & (int(rt) - 1)) != 0:
                raise ValueError("RNAGAN_F16_LOSS_SCALE must be the square of a power of two (1, 4, 16, ..., 4096, 16384, ...)")
            self.loss_scale, self.gp_seed_scale, self.gp_tangent_scale = ls, rt, rt
        self.algo = algo
        self._wsbuf = None
        # optional per-launch timing (bench.py roofline): list of (kernel family, algorithmic FLOPs,
        # start event, end event); events are recorded on the stream the kernels are enqueued on
        self.timing = None
        # weight-gradient launches are forked onto a side stream (they are off the backward chain's critical
        # path and MFMA-bound, so they co-run with the HBM-bound BatchNorm passes); own workspace; inputs are
        # kept alive until join() so the caching allocator cannot hand their memory to the main stream early
        self.side_stream = None
        self._wsbuf_side = None
        self._ws_retired = []
        self._in_side = False
        self._keep = []
        self.use_side = os.environ.get("RNAGAN_SIDE_STREAM", "0") != "0"
        # tanh backward / channel sums / squared norm of D's input gradient inside the kernel that writes it
        self.fuse_input_post = os.environ.get("RNAGAN_INPUT_POST", "1") != "0"
        self.pack_from_shadow = os.environ.get("RNAGAN_PACK_FROM_SHADOW", "1") != "0"
        # a deferred weight gradient without split-K is formed AND stepped by one launch of the optimizer (rg_conv_wgrad_adam)
        self.wgrad_in_step = os.environ.get("RNAGAN_WGRAD_ADAM", "1") != "0"
        # {dw address: ConvW} of the image-side layers whose weight-gradient partials the optimizer sums (set by the train_op runner
        # for the duration of one gradient pass, with ConvW.defer_slabs)
        self._skinny_defer = None
        # the image-side weight-gradient kernel also forms the layer's bias gradient (a column of ones in its patch operand)
        self.skinny_bias = os.environ.get("RNAGAN_SKINNY_BIAS", "1") != "0"
        self.epilogue_stats = os.environ.get("RNAGAN_EPILOGUE_STATS", "1") != "0"
        # synchronised (global-batch) statistics in a data-parallel run (dist.attach_sync): an in-place SUM all-reduce
        # for small fp32 tensors and the number of ranks; None = rank-local statistics (plain DDP semantics)
        self.stat_reduce = None
        self.stat_world = 1
        # split-K conv -> BatchNorm without the intermediate passes (rg_splitbn.hip): the conv leaves its slabs, the BatchNorm
        # kernel reduces them (RNAGAN_SPLIT_BN=0: conv + slab reduction + statistics + finisher + apply as separate launches)
        self.split_bn = os.environ.get("RNAGAN_SPLIT_BN", "1") != "0"
        self._split_bn_dp = None       # decided at the first use: fused split-K BatchNorm kernels only outside data-parallel runs
        # data-gradient convs can also produce the BatchNorm-backward sums of the block they feed in their epilogue
        # (rg_conv_*_bnbwd).  Built, exact, and NOT faster: the extra z read lands in the conv's tail, where every workgroup of
        # the chip is in its epilogue at once (11.95-12.09 ms with the separate reduction pass vs 12.03-12.07 ms with the
        # fused form, same box; the conv family drops from 0.408 to 0.382 of peak) -- off by default, RNAGAN_BWD_EPILOGUE=1
        self.bwd_epilogue = os.environ.get("RNAGAN_BWD_EPILOGUE", "0") != "0"
        # fp32 storage: the stride-2 convs / transposed convs / weight gradients on the bf16 matrix cores from operands split ONCE
        # PER TENSOR into bf16 planes (rg_conv8f.hip): products per fp32 product -- 6 (every term down to 2^-24: fp32-grade, the
        # default), 3 (2^-16 per product), 0: the per-tile kernels of rg_generic.hip (f32mma option)
        self.f32_planes = int(os.environ.get("RNAGAN_F32_PLANES", "6")) if self.dt == RG_F32 else 0
        if self.f32_planes not in (0, 3, 6):
            raise ValueError("RNAGAN_F32_PLANES must be 0, 3 or 6")
        self._slabs_pending = None    # the tensor whose deferred split-K slabs currently occupy the workspace
        self._sb_sync = None          # hand-off words of the fused kernels: zeroed once, left zero by every launch
        self._sb_scratch = None
        self._sb_retired = []
        _LIVE_OPS.add(self)

    # ------------------------------------------------------------------ plumbing
    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _ws(self, nbytes: int):
        """Caller-owned workspace of the C ABI calls.  A buffer that is outgrown is RETIRED, not freed: captured HIP
        graphs replay launches that hold its address (growth is geometric, so the retired list stays short)."""
        nbytes = max(int(nbytes), 256)
        if self._slabs_pending is not None and not self._in_side:
            # a deferred conv's split-K slabs live in this buffer until the BatchNorm op that was promised consumes them
            raise RuntimeError("rna_gan_amd: a conv launched with defer= left its split-K slabs in the workspace and another "
                               "op asked for the workspace before the BatchNorm op consumed them")
        if self._in_side:
            if self._wsbuf_side is None or self._wsbuf_side.numel() < nbytes:
                if self._wsbuf_side is not None:
                    self._ws_retired.append(self._wsbuf_side)
                self._wsbuf_side = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
            return self._wsbuf_side
        if self._wsbuf is None or self._wsbuf.numel() < nbytes:
            if self._wsbuf is not None:
                self._ws_retired.append(self._wsbuf)
            self._wsbuf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
        return self._wsbuf

    def _sb_bufs(self, M, C, groups):
        """(scratch, sync) of the fused split-K BatchNorm kernels; an outgrown scratch buffer is retired, not freed (graphs)."""
        if self._sb_sync is None:
            self._sb_sync = torch.zeros(int(self.lib.rg_slab_bn_sync_words()), dtype=torch.int32, device=self.device)
        need = int(self.lib.rg_slab_bn_scratch_bytes(int(M), int(C), int(groups)))
        if self._sb_scratch is None or self._sb_scratch.numel() < need:
            if self._sb_scratch is not None:
                self._sb_retired.append(self._sb_scratch)
            self._sb_scratch = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._sb_scratch, self._sb_sync

    def _defer_split(self, up, N, Hl, Wl, O, I, rows_out, C, groups):
        """Split factor if this conv launch can hand its slabs to the fused BatchNorm kernel (groups batch groups), else 0."""
        if not groups or not self.split_bn or self.dt == RG_F32 or self.stat_reduce is not None or rows_out % groups:
            return 0
        # The fused kernels rendezvous across ALL their workgroups (<= 256 blocks of 1024 threads).  In a data-parallel run
        # RCCL's kernels hold some CUs for the length of a collective, and blocks that cannot be placed keep the resident
        # ones spinning.  Measured with a STAND-IN kernel holding 32 CUs for 500 us per collective: the fused
        # form stays ahead with a light stand-in (12.6 vs 12.8 ms) and level with a register-heavy one (12.8 vs 12.8) -- but
        # that is not RCCL, and a timed-out rendezvous yields garbage, so with more than one rank the separate launches are
        # the default until a real multi-GPU run has A/B'd it (tools/dp_first_run.sh does): RNAGAN_SPLIT_BN_DP=1 turns the
        # fused kernels on there; check_handoffs() reports a timeout either way.  One rank (RNAGAN_FORCE_DP) has no collective
        # kernel beside it and keeps them.
        if self._split_bn_dp is None:
            from . import dist as D_
            dflt = "1" if D_.world_size() <= 1 else "0"
            self._split_bn_dp = (not D_.active()) or os.environ.get("RNAGAN_SPLIT_BN_DP", dflt) != "0"
        if not self._split_bn_dp:
            return 0
        ns = int(self.lib.rg_conv_split(up, N, Hl, Wl, O, I, self.dt, self.algo))
        if ns <= 1 or not self.lib.rg_slab_bn_supported(rows_out // groups, C, groups, ns):
            return 0
        return ns

    @contextmanager
    def side(self, *tensors):
        """Run the enclosed ops on the side stream, ordered after everything enqueued so far on the
        current stream.  ``tensors``: operands to keep alive until join()."""
        if not self.use_side or self.timing is not None:
            yield
            return
        if self.side_stream is None:
            self.side_stream = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        self.side_stream.wait_stream(main)
        self._keep.extend(tensors)
        self._in_side = True
        try:
            with torch.cuda.stream(self.side_stream):
                yield
        finally:
            self._in_side = False

    def join(self):
        """Make the current stream wait for the side stream's work (before gradients are consumed)."""
        if self.side_stream is not None and self.use_side:
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
        self._keep.clear()

    def _timed(self, key, flops, thunk, cw=None):
        if self.timing is None:
            return thunk()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.device))
        r = thunk()
        e1.record(torch.cuda.current_stream(self.device))
        # (family, algorithmic FLOPs, start, end, network the layer belongs to: "G" / "D" / None)
        self.timing.append((key, float(flops), e0, e1, getattr(cw, "owner", None)))
        return r

    def _act(self, *shape):
        return torch.empty(shape, dtype=self.act_dtype, device=self.device)

    def _f32(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _planes(self, t):
        """bf16 planes [3][numel] of an fp32 tensor (v = h + m + l exactly, rg_split_planes), cached on the tensor object: an
        activation is split once for the conv that consumes it and for the weight gradient that reads it again."""
        # (keyed by the tensor's version counter: a torch in-place op on it invalidates the planes.  The library's own kernels
        # write through raw pointers without touching that counter -- every op of this class returns a FRESH tensor, and a caller
        # that overwrites an activation through the C ABI has to `del t._rg_planes`)
        c = getattr(t, "_rg_planes", None)
        if c is None or c[0] != t._version:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 8 == 0
            p = c[1] if c is not None else torch.empty((3, t.numel()), dtype=torch.bfloat16, device=self.device)
            check(self.lib.rg_split_planes(_ptr(t), _ptr(p), t.numel(), self.stream), "rg_split_planes")
            c = t._rg_planes = (t._version, p)
        return c[1]

    def _plane_packs(self, cw: ConvW):
        """(planes of wdn[O][16][I], planes of wup[16][I][O]) of a tap-major fp32 master, rebuilt when its version changes."""
        if cw.packs is None or cw.packs_version != cw.version:
            O, I = cw.O, cw.I
            if cw.packs is None:
                cw.packs = (torch.empty((3, O, 16, I), dtype=torch.bfloat16, device=self.device),
                            torch.empty((3, 16, I, O), dtype=torch.bfloat16, device=self.device))
            check(self.lib.rg_split_planes(_ptr(cw.w), _ptr(cw.packs[0]), cw.w.numel(), self.stream), "rg_split_planes")
            for p in range(3):          # the transposed conv's operand image, plane by plane
                check(self.lib.rg_pack_conv_wup_from_bf16(_ptr(cw.packs[0][p]), _ptr(cw.packs[1][p]), O, I, self.stream),
                      "rg_pack_conv_wup_from_bf16")
            cw.packs_version = cw.version
        return cw.packs

    def _conv_planes(self, up, x, cw: ConvW, N, Hl, Wl, O, I, y, want_stats, mask=None, mslope=1.0):
        """The fp32 conv on bf16 planes (rg_f32p_conv) when the shape has that kernel; returns the statistics partials (or None),
        or False when the caller has to use the RG_F32 entry point."""
        if not self.f32_planes or not self.lib.rg_f32p_conv_supported(up, N, Hl, Wl, O, I, self.f32_planes):
            return False
        wp = self._plane_packs(cw)[1 if up else 0]
        xp = self._planes(x)
        st = None
        if want_stats and self.epilogue_stats and self.stat_reduce is None:
            rows = self.lib.rg_f32p_conv_stats_rows(up, N, Hl, Wl, O, I, self.f32_planes)
            st = self._f32(rows, 2, I if up else O) if rows > 0 else None
        ws = self._ws(self.lib.rg_f32p_conv_workspace_bytes(up, N, Hl, Wl, O, I, self.f32_planes))
        self._timed("conv_fwd_dgrad", 2.0 * N * Hl * Wl * O * I * 16, lambda: check(
            self.lib.rg_f32p_conv(up, _ptr(xp), _ptr(wp), _ptr(y), N, Hl, Wl, O, I, self.f32_planes, _ptr(st), _ptr(mask), float(mslope),
                                  _ptr(ws), ws.numel(), self.stream), "rg_f32p_conv"), cw=cw)
        return st

    def _packs(self, cw: ConvW):
        if self.dt == RG_F32:
            return None, None
        if cw.packs is None or cw.packs_version != cw.version:
            O, I = cw.O, cw.I
            if cw.packs is None:
                wdn = cw.shadow if cw.shadow is not None else torch.empty((O, 16, I), dtype=self.h16,
                                                                         device=self.device)
                cw.packs = (wdn, torch.empty((16, I, O), dtype=self.h16, device=self.device))
            # wdn is a cast of the tap-major master: skipped when the fused Adam already wrote it (shadow), and then
            # wup is transposed from that bf16 copy (half the read traffic of the fp32 master)
            need_wdn = cw.shadow is None or cw.shadow_version != cw.version or not self.pack_from_shadow
            if need_wdn:
                check(self.lib.rg_pack_conv_weight(_ptr(cw.w), _ptr(cw.packs[0]), _ptr(cw.packs[1]), O, I, self.H16,
                                                   self.stream), "rg_pack_conv_weight")
            else:
                check(self.lib.rg_pack_conv_wup_from_bf16(_ptr(cw.shadow), _ptr(cw.packs[1]), O, I, self.stream),
                      "rg_pack_conv_wup_from_bf16")
            cw.packs_version = cw.version
            if cw.shadow is not None:
                cw.shadow_version = cw.version
        return cw.packs

    def refresh_packs(self, cws):
        """Rebuild the stale transposed-conv weight images (wup) of several layers in ONE launch when the fused Adam left
        their bf16 shadows current (the common case after an optimizer step: 5 transposes of 10-40 us per network became
        one).  Layers that do not qualify are left to the lazy per-layer path (_packs)."""
        if self.dt == RG_F32 or not self.pack_from_shadow or os.environ.get("RNAGAN_PACK_MULTI", "1") == "0":
            return
        todo = []
        for cw in cws:
            if (cw.layout == "OHWI" and cw.shadow is not None and cw.shadow_version == cw.version and
                    (cw.packs is None or cw.packs_version != cw.version) and cw.O % 64 == 0 and (16 * cw.I) % 128 == 0):
                if cw.packs is None:
                    cw.packs = (cw.shadow, torch.empty((16, cw.I, cw.O), dtype=self.h16, device=self.device))
                if cw.packs[0] is cw.shadow:
                    todo.append(cw)
        if len(todo) < 2:
            return
        import ctypes as C
        for i in range(0, len(todo), 8):
            part = todo[i:i + 8]
            n = len(part)
            src = (C.c_void_p * n)(*[cw.shadow.data_ptr() for cw in part])
            dst = (C.c_void_p * n)(*[cw.packs[1].data_ptr() for cw in part])
            Os = (C.c_int * n)(*[cw.O for cw in part])
            Is = (C.c_int * n)(*[cw.I for cw in part])
            check(self.lib.rg_pack_conv_wup_from_bf16_multi(n, src, dst, Os, Is, self.stream), "rg_pack_conv_wup_from_bf16_multi")
            for cw in part:
                cw.packs_version = cw.version

    # ------------------------------------------------------------------ conv family
    @staticmethod
    def _tap_major(cw: ConvW):
        if cw.layout != "OHWI" or not cw.w.is_contiguous():
            raise ValueError("the HIP conv kernels take tap-major masters w[O][4][4][I]: re-home the module's 4x4 "
                             "conv weights with rna_gan_amd.models.tap_major_(module) (or build the handle with "
                             "ConvW.from_param on such a parameter)")

    def _stats_buf(self, up, N, Hl, Wl, O, I, C):
        """Buffer for the BatchNorm partial sums a conv epilogue can write, or None (split-K / generic kernel)."""
        if not self.epilogue_stats or self.stat_reduce is not None:
            return None
        rows = self.lib.rg_conv_stats_rows(up, N, Hl, Wl, O, I, self.dt, self.algo)
        return self._f32(rows, 2, C) if rows > 0 else None

    def _bn_bwd_fused(self, up, y, cw_ptr, x, dims, bn_bwd, flops):
        """Launch the conv with the consumer's BatchNorm-backward sums in its epilogue when this shape has that form; the
        partial rows ride on the result as ``_rg_bwd_partials`` for bn_act_bwd / bn_act_bwd2.  True when launched."""
        z, mean, invstd, gamma, beta, slope, groups = bn_bwd
        N, Hl, Wl, O, I = dims
        if (self.dt == RG_F32 or self.stat_reduce is not None or not self.bwd_epilogue or z.shape != y.shape or
                z.dtype != y.dtype or not z.is_contiguous()):
            return False
        rows = int(self.lib.rg_conv_bnbwd_rows(up, N, Hl, Wl, O, I, groups, self.dt, self.algo))
        if rows <= 0:
            return False
        C = y.shape[-1]
        part = self._f32(rows, 2, C)
        ws = self._ws(self.lib.rg_conv_workspace_bytes(up, N, Hl, Wl, O, I, self.dt, self.algo))
        fn = self.lib.rg_conv_up_bnbwd if up else self.lib.rg_conv_down_bnbwd
        a = (N, Hl, Wl, O, I) if up else (N, 2 * Hl, 2 * Wl, I, O)
        self._timed("conv_fwd_dgrad", flops, lambda: check(
            fn(_ptr(x), _ptr(cw_ptr), _ptr(y), *a, _ptr(z), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), float(slope),
               int(groups), _ptr(part), self.dt, self.algo, _ptr(ws), ws.numel(), self.stream),
            "rg_conv_up_bnbwd" if up else "rg_conv_down_bnbwd"))
        y._rg_bwd_partials = (part, 4 if up else 1, groups)
        return True

    def conv_down(self, x, cw: ConvW, want_stats=False, defer=0, bn_bwd=None):
        """Stride-2 conv.  want_stats: also return the per-tile column sums of y and y^2 written by the MFMA epilogue
        (None when this shape cannot produce them) for bn_forward(..., partials=...).
        defer = g > 0: the caller promises that the NEXT op on the result is the train-mode BatchNorm op (forward: bn_forward /
        bn_forward2, backward: bn_act_bwd / bn_act_bwd2) over g batch groups.  If this launch runs split-K, only the slab
        launch is issued; the result tensor is returned UNWRITTEN with ``_rg_slabs`` attached and the BatchNorm op reduces
        the slabs (and writes the tensor) itself."""
        N, Hi, Wi, I = x.shape
        O = cw.O
        self._tap_major(cw)
        assert cw.I == I and x.is_contiguous()
        y = self._act(N, Hi // 2, Wi // 2, O)
        if self.f32_planes:
            st = self._conv_planes(0, x, cw, N, Hi // 2, Wi // 2, O, I, y, want_stats)
            if st is not False:
                return (y, st) if want_stats else y
        wdn, _ = self._packs(cw)
        ns = self._defer_split(0, N, Hi // 2, Wi // 2, O, I, N * (Hi // 2) * (Wi // 2), O, defer)
        if ns:
            ws = self._ws(self.lib.rg_conv_workspace_bytes(0, N, Hi // 2, Wi // 2, O, I, self.dt, self.algo))
            self._timed("conv_fwd_dgrad", 2.0 * N * (Hi // 2) * (Wi // 2) * O * I * 16, lambda: check(
                self.lib.rg_conv_down_partial(_ptr(x), _ptr(wdn), N, Hi, Wi, I, O, self.dt, self.algo, _ptr(ws), ws.numel(),
                                              self.stream), "rg_conv_down_partial"), cw=cw)
            y._rg_slabs = SlabRef(ws, ns, y.numel(), defer,
                                  self.lib.rg_conv_slab_dtype(0, N, Hi // 2, Wi // 2, O, I, self.dt, self.algo))
            self._slabs_pending = y
            return (y, None) if want_stats else y
        if bn_bwd is not None and not want_stats and self._bn_bwd_fused(
                0, y, wdn, x, (N, Hi // 2, Wi // 2, O, I), bn_bwd, 2.0 * N * (Hi // 2) * (Wi // 2) * O * I * 16):
            return y
        st = self._stats_buf(0, N, Hi // 2, Wi // 2, O, I, O) if want_stats else None
        ws = self._ws(self.lib.rg_conv_workspace_bytes(0, N, Hi // 2, Wi // 2, O, I, self.dt, self.algo))
        self._timed("conv_fwd_dgrad", 2.0 * N * (Hi // 2) * (Wi // 2) * O * I * 16, lambda: check(
            self.lib.rg_conv_down(_ptr(x), _ptr(cw.w), _ptr(wdn), _ptr(y), N, Hi, Wi, I, O, _ptr(st), self.dt,
                                  self.algo, _ptr(ws), ws.numel(), self.stream), "rg_conv_down"), cw=cw)
        return (y, st) if want_stats else y

    def conv_up(self, x, cw: ConvW, mask_act=None, slope=1.0, want_stats=False, defer=0, bn_bwd=None):
        """Transposed conv; with mask_act (same shape as the result) the LeakyReLU backward
        ``y *= (mask_act > 0 ? 1 : slope)`` is applied in the kernel's epilogue.  want_stats, defer: as conv_down."""
        N, Ho, Wo, O = x.shape
        I = cw.I
        self._tap_major(cw)
        assert cw.O == O and x.is_contiguous()
        y = self._act(N, 2 * Ho, 2 * Wo, I)
        if self.f32_planes:
            assert mask_act is None or (mask_act.shape == y.shape and mask_act.dtype == y.dtype and mask_act.is_contiguous())
            fused = mask_act is not None and bool(self.lib.rg_f32p_conv_mask_supported(1, N, Ho, Wo, O, I, self.f32_planes))
            st = self._conv_planes(1, x, cw, N, Ho, Wo, O, I, y, want_stats and mask_act is None,
                                   mask=mask_act if fused else None, mslope=slope)
            if st is not False:
                if mask_act is not None and not fused:   # the consumer's LeakyReLU backward as one elementwise pass behind the conv
                    y = self.lrelu_bwd(y, mask_act, slope)
                return (y, st) if want_stats else y
        _, wup = self._packs(cw)
        ns = self._defer_split(1, N, Ho, Wo, O, I, N * 4 * Ho * Wo, I, defer) if mask_act is None else 0
        if ns:
            ws = self._ws(self.lib.rg_conv_workspace_bytes(1, N, Ho, Wo, O, I, self.dt, self.algo))
            self._timed("conv_fwd_dgrad", 2.0 * N * Ho * Wo * O * I * 16, lambda: check(
                self.lib.rg_conv_up_partial(_ptr(x), _ptr(wup), N, Ho, Wo, O, I, self.dt, self.algo, _ptr(ws), ws.numel(),
                                            self.stream), "rg_conv_up_partial"), cw=cw)
            y._rg_slabs = SlabRef(ws, ns, y.numel(), defer,
                                  self.lib.rg_conv_slab_dtype(1, N, Ho, Wo, O, I, self.dt, self.algo))
            self._slabs_pending = y
            return (y, None) if want_stats else y
        if bn_bwd is not None and mask_act is None and not want_stats and self._bn_bwd_fused(
                1, y, wup, x, (N, Ho, Wo, O, I), bn_bwd, 2.0 * N * Ho * Wo * O * I * 16):
            return y
        assert mask_act is None or (mask_act.shape == y.shape and mask_act.dtype == y.dtype and mask_act.is_contiguous())
        st = self._stats_buf(1, N, Ho, Wo, O, I, I) if (want_stats and mask_act is None) else None
        ws = self._ws(self.lib.rg_conv_workspace_bytes(1, N, Ho, Wo, O, I, self.dt, self.algo))
        bits = getattr(mask_act, "_rg_sign_bits", None) if mask_act is not None else None
        if bits is not None and self.lib.rg_conv_up_maskbits_supported(N, Ho, Wo, O, I, self.dt, self.algo):
            self._timed("conv_fwd_dgrad", 2.0 * N * Ho * Wo * O * I * 16, lambda: check(
                self.lib.rg_conv_up_maskbits(_ptr(x), _ptr(wup), _ptr(y), N, Ho, Wo, O, I, _ptr(bits), float(slope), self.dt,
                                             self.algo, _ptr(ws), ws.numel(), self.stream), "rg_conv_up_maskbits"), cw=cw)
            return (y, None) if want_stats else y
        self._timed("conv_fwd_dgrad", 2.0 * N * Ho * Wo * O * I * 16, lambda: check(
            self.lib.rg_conv_up(_ptr(x), _ptr(cw.w), _ptr(wup), _ptr(y), N, Ho, Wo, O, I, _ptr(mask_act), float(slope),
                                _ptr(st), self.dt, self.algo, _ptr(ws), ws.numel(), self.stream), "rg_conv_up"), cw=cw)
        return (y, st) if want_stats else y

    def conv_up_affine(self, x, cw: ConvW, scale, shift, slope: float):
        """Transposed conv with the eval-mode BatchNorm affine and LeakyReLU fused into its epilogue (bf16 MFMA path):
        lrelu(conv_up(x) * scale[c] + shift[c], slope), rounded once.  None when this shape / precision has no fused form."""
        if self.dt == RG_F32:
            return None
        N, Ho, Wo, O = x.shape
        I = cw.I
        if O % 64 != 0 or I % 8 != 0 or I < 64 or (Ho & (Ho - 1)) or (Wo & (Wo - 1)):
            return None
        self._tap_major(cw)
        _, wup = self._packs(cw)
        y = self._act(N, 2 * Ho, 2 * Wo, I)
        ws = self._ws(self.lib.rg_conv_workspace_bytes(1, N, Ho, Wo, O, I, self.dt, self.algo))
        sc, sh = scale.float().contiguous(), shift.float().contiguous()
        self._timed("conv_fwd_dgrad", 2.0 * N * Ho * Wo * O * I * 16, lambda: check(
            self.lib.rg_conv_up_affine(_ptr(x), _ptr(wup), _ptr(y), N, Ho, Wo, O, I, _ptr(sc), _ptr(sh), float(slope),
                                       _ptr(ws), ws.numel(), self.stream), "rg_conv_up_affine"), cw=cw)
        return y

    def g0_fwd_affine(self, z, cw: ConvW, scale, shift, slope: float):
        """Generator layer 0 with the folded BatchNorm affine + LeakyReLU in the GEMM epilogue (scale / shift per channel
        c, expanded here to the layer's 16*C (tap, c) columns).  None when there is no fused form."""
        N, E = z.shape
        C = cw.w.shape[1]
        if self.dt == RG_F32 or E % 64 != 0 or C % 8 != 0:
            return None
        self.g0_pack(cw)
        y = self._act(N, 4, 4, C)
        ws = self._ws(self.lib.rg_g0_workspace_bytes(N, E, C, self.dt, self.algo))
        sc, sh = scale.float().repeat(16).contiguous(), shift.float().repeat(16).contiguous()
        check(self.lib.rg_g0_fwd_affine(_ptr(z), _ptr(cw.packs[0]), _ptr(y), N, E, C, _ptr(sc), _ptr(sh), float(slope),
                                        _ptr(ws), ws.numel(), self.stream), "rg_g0_fwd_affine")
        return y

    # ------------------------------------------------------------------ fp8 (e4m3) inference operands
    def cast_fp8(self, t, mul=1.0):
        """fp32 tensor -> fp8 e4m3 bytes (uint8 tensor of the same shape), fp8(t * mul)."""
        t = t.float().contiguous()
        out = torch.empty(t.shape, dtype=torch.uint8, device=self.device)
        check(self.lib.rg_cast_fp8(_ptr(t), _ptr(out), t.numel(), float(mul), self.stream), "rg_cast_fp8")
        return out

    def fp8_supported(self, M, K, Ncols, taps):
        return bool(self.lib.rg_fp8_supported(int(M), int(K), int(Ncols), int(taps)))

    def fp8_pack_up(self, cw: ConvW):
        """(wup8[16][I][O] fp8 bytes, colscale[I]) of a transposed-conv weight: per output channel i the weights are
        divided by s_i = max |w[., i, .]| / 448 before the cast, s_i is folded into the epilogue scale.  Cached per
        master version (inference weights do not change).  The re-layout is host-side plumbing on the device."""
        key = ("fp8up", cw.version)
        if getattr(cw, "_fp8", None) is not None and cw._fp8[0] == key:
            return cw._fp8[1], cw._fp8[2]
        w = cw.w if cw.layout == "OHWI" else cw.w.permute(0, 2, 3, 1)          # [O][4][4][I]
        wt = w.permute(1, 2, 3, 0).reshape(16, cw.I, cw.O).float()               # [tap][i][o]
        s = wt.abs().amax(dim=(0, 2)).clamp_min(1e-30) / 448.0
        q = self.cast_fp8((wt / s[None, :, None]).contiguous())
        cw._fp8 = (key, q, s.contiguous())
        return q, cw._fp8[2]

    def fp8_pack_g0(self, cw: ConvW):
        """(b8[(tap, c)][E] fp8 bytes, colscale[16*C]) of the generator's first weight w[E][C][4][4]."""
        key = ("fp8g0", cw.version)
        if getattr(cw, "_fp8", None) is not None and cw._fp8[0] == key:
            return cw._fp8[1], cw._fp8[2]
        E, C = cw.w.shape[0], cw.w.shape[1]
        b = cw.w.permute(2, 3, 1, 0).reshape(16 * C, E).float()
        s = b.abs().amax(dim=1).clamp_min(1e-30) / 448.0
        q = self.cast_fp8((b / s[:, None]).contiguous())
        cw._fp8 = (key, q, s.contiguous())
        return q, cw._fp8[2]

    def conv_up_fp8(self, x8, cw: ConvW, scale, shift, slope: float, out_fp8: bool):
        N, Ho, Wo, O = x8.shape
        I = cw.I
        assert x8.dtype == torch.uint8 and x8.is_contiguous() and cw.O == O
        q, s = self.fp8_pack_up(cw)
        y = torch.empty((N, 2 * Ho, 2 * Wo, I), dtype=torch.uint8 if out_fp8 else torch.bfloat16, device=self.device)
        sc, sh = (scale.float() * s).contiguous(), shift.float().contiguous()
        self._timed("conv_fwd_dgrad", 2.0 * N * Ho * Wo * O * I * 16, lambda: check(
            self.lib.rg_conv_up_fp8(_ptr(x8), _ptr(q), _ptr(y), N, Ho, Wo, O, I, _ptr(sc), _ptr(sh), float(slope),
                                    int(out_fp8), self.stream), "rg_conv_up_fp8"), cw=cw)
        return y

    def g0_fwd_fp8(self, z8, cw: ConvW, scale, shift, slope: float, out_fp8: bool):
        N, E = z8.shape
        C = cw.w.shape[1]
        q, s = self.fp8_pack_g0(cw)
        y = torch.empty((N, 4, 4, C), dtype=torch.uint8 if out_fp8 else torch.bfloat16, device=self.device)
        sc, sh = (scale.float().repeat(16) * s).contiguous(), shift.float().repeat(16).contiguous()
        check(self.lib.rg_gemm_fp8(_ptr(z8), _ptr(q), _ptr(y), N, E, 16 * C, _ptr(sc), _ptr(sh), float(slope), int(out_fp8),
                                   self.stream), "rg_gemm_fp8")
        return y

    def selftest_fp8(self):
        d = torch.zeros(2, dtype=torch.int32, device=self.device)
        check(self.lib.rg_selftest_fp8(_ptr(d), self.stream), "rg_selftest_fp8")
        return d.cpu().tolist()

    def _wgrad_slabs(self, low0, high0, low1, high1, cw: ConvW, accumulate: bool, flops):
        """The weight gradient of a layer whose optimizer step follows immediately (cw.defer_slabs, set by the train_op runner):
        a split-K launch leaves its fp32 partial slabs in a buffer of the layer's own (cw.pending_slabs) and the reduction into
        dw is skipped -- rna_gan_amd.optim.Adam sums the slabs inside its step (rg_adam_step_slabs).  True when launched."""
        if not cw.defer_slabs or self.dt == RG_F32 or self.stat_reduce is not None or self._in_side:
            return False
        if accumulate or cw.pending_slabs is not None or cw.pending_wgrad is not None:
            raise RuntimeError("rna_gan_amd: a second weight-gradient contribution for a layer whose first one is still deferred "
                               "split-K slabs (defer_slabs expects ONE weight-gradient launch per layer and pass)")
        N, Ho, Wo, O = low0.shape
        I = high0.shape[3]
        dw = cw.dw
        if cw.w.data_ptr() % 16 or dw.data_ptr() % 16:
            return False
        if cw.wire_slot is not None:
            if self.lib.rg_conv_wgrad_adam_supported(N, Ho, Wo, O, I, int(low1 is not None), self.dt, self.algo):
                # data parallel, a plan without split-K: the tile goes onto the wire as bf16, once (rg_conv_wgrad_wire)
                self._timed("conv_wgrad", flops, lambda: check(
                    self.lib.rg_conv_wgrad_wire(_ptr(low0), _ptr(high0), _ptr(low1), _ptr(high1), _ptr(cw.wire_slot), N, Ho, Wo,
                                                O, I, self.dt, self.algo, self.stream), "rg_conv_wgrad_wire"), cw=cw)
                cw.pending_slabs = (None, -1, 0)
                return True
        elif (self.wgrad_in_step and cw.shadow is not None and
                self.lib.rg_conv_wgrad_adam_supported(N, Ho, Wo, O, I, int(low1 is not None), self.dt, self.algo)):
            # a plan WITHOUT split-K (the two 33.5 M-parameter layers at batch 64): nothing is launched here -- the operands
            # stay on the handle and the optimizer forms the gradient tile and applies its step to it in one launch
            # (rg_conv_wgrad_adam: the 4 bytes per parameter written here and read back by the streaming Adam disappear)
            cw.pending_wgrad = ("conv", low0, high0, low1, high1, (N, Ho, Wo, O, I), self.dt, self.algo, flops)
            return True
        nb = int(self.lib.rg_conv_wgrad_workspace_bytes(N, Ho, Wo, O, I, self.dt, self.algo))
        if nb <= 0:
            return False                           # no split-K plan for this shape (or no matrix-core kernel): nothing to defer
        if cw._slab_ws is None or cw._slab_ws.numel() < nb:
            # persistent: captured graphs (keyed per batch shape) and the data-parallel wire table hold its address, so a buffer
            # that a larger batch outgrows is RETIRED, never freed -- a replay of the smaller batch's graph still writes / reads it
            if cw._slab_ws is not None:
                self._ws_retired.append(cw._slab_ws)
            cw._slab_ws = torch.empty(nb + 4096, dtype=torch.uint8, device=self.device)
        ns, sdt = ctypes.c_int(0), ctypes.c_int(0)
        self._timed("conv_wgrad", flops, lambda: check(
            self.lib.rg_conv_wgrad_slabs(_ptr(low0), _ptr(high0), _ptr(low1), _ptr(high1), _ptr(dw), N, Ho, Wo, O, I, self.dt,
                                         self.algo, _ptr(cw._slab_ws), cw._slab_ws.numel(), ctypes.addressof(ns),
                                         ctypes.addressof(sdt), self.stream),
            "rg_conv_wgrad_slabs"), cw=cw)
        # (buffer, number of partial slabs, their element type: RG_F32 / RG_BF16)
        cw.pending_slabs = (cw._slab_ws, int(ns.value), int(sdt.value)) if ns.value > 1 else None
        return True

    def _wgrad_planes(self, low0, high0, low1, high1, cw: ConvW, accumulate: bool, flops):
        """fp32 storage: the weight gradient on bf16 planes of its operands (rg_f32p_wgrad); True when launched."""
        if not self.f32_planes:
            return False
        N, Ho, Wo, O = low0.shape
        I = high0.shape[3]
        if not self.lib.rg_f32p_wgrad_supported(N, Ho, Wo, O, I, self.f32_planes):
            return False
        two = low1 is not None
        pl0, ph0 = self._planes(low0), self._planes(high0)
        pl1, ph1 = (self._planes(low1), self._planes(high1)) if two else (None, None)
        ws = self._ws(self.lib.rg_f32p_wgrad_workspace_bytes(N, Ho, Wo, O, I, self.f32_planes, int(two)))
        self._timed("conv_wgrad", flops, lambda: check(
            self.lib.rg_f32p_wgrad(_ptr(pl0), _ptr(ph0), _ptr(pl1), _ptr(ph1), _ptr(cw.dw), N, Ho, Wo, O, I, self.f32_planes,
                                   int(accumulate), _ptr(ws), ws.numel(), self.stream), "rg_f32p_wgrad"), cw=cw)
        return True

    def conv_wgrad(self, low, high, cw: ConvW, accumulate: bool):
        N, Ho, Wo, O = low.shape
        I = high.shape[3]
        self._tap_major(cw)
        dw = cw.dw
        assert high.shape[1] == 2 * Ho and tuple(dw.shape) == (O, 4, 4, I) and dw.is_contiguous()
        if self._wgrad_slabs(low, high, None, None, cw, accumulate, 2.0 * N * Ho * Wo * O * I * 16):
            return
        if self._wgrad_planes(low, high, None, None, cw, accumulate, 2.0 * N * Ho * Wo * O * I * 16):
            return
        nb = self.lib.rg_conv_wgrad_workspace_bytes(N, Ho, Wo, O, I, self.dt, self.algo)
        ws = self._ws(nb)
        self._timed("conv_wgrad", 2.0 * N * Ho * Wo * O * I * 16, lambda: check(
            self.lib.rg_conv_wgrad(_ptr(low), _ptr(high), _ptr(dw), N, Ho, Wo, O, I, self.dt, int(accumulate),
                                   self.algo, _ptr(ws), ws.numel(), self.stream), "rg_conv_wgrad"), cw=cw)

    def conv_wgrad2(self, low0, high0, low1, high1, cw: ConvW, accumulate: bool):
        """dw (+)= wgrad(low0, high0) + wgrad(low1, high1) in one launch (one split-K reduction)."""
        N, Ho, Wo, O = low0.shape
        I = high0.shape[3]
        self._tap_major(cw)
        dw = cw.dw
        assert low1.shape == low0.shape and high1.shape == high0.shape
        assert tuple(dw.shape) == (O, 4, 4, I) and dw.is_contiguous()
        if self._wgrad_slabs(low0, high0, low1, high1, cw, accumulate, 4.0 * N * Ho * Wo * O * I * 16):
            return
        if self._wgrad_planes(low0, high0, low1, high1, cw, accumulate, 4.0 * N * Ho * Wo * O * I * 16):
            return
        nb = self.lib.rg_conv_wgrad_workspace_bytes(N, Ho, Wo, O, I, self.dt, self.algo)
        ws = self._ws(nb)
        self._timed("conv_wgrad", 4.0 * N * Ho * Wo * O * I * 16, lambda: check(
            self.lib.rg_conv_wgrad2(_ptr(low0), _ptr(high0), _ptr(low1), _ptr(high1), _ptr(dw), N, Ho, Wo, O, I,
                                    self.dt, int(accumulate), self.algo, _ptr(ws), ws.numel(), self.stream),
            "rg_conv_wgrad2"), cw=cw)

    def u8_to_norm(self, u8, mean=0.5, std=0.5):
        """uint8 tiles (any shape, CHW order kept) -> fp32 (x / 255 - mean) / std on the device: the input transform of
        src/histopathology_gan.py:106-109, bit-identical to the host version."""
        assert u8.dtype == torch.uint8 and u8.is_contiguous() and u8.is_cuda
        y = self._f32(*u8.shape)
        check(self.lib.rg_u8_to_norm(_ptr(u8), _ptr(y), u8.numel(), float(mean), float(std), self.stream), "rg_u8_to_norm")
        return y

    def export_images_nhwc(self, img_nchw):
        """NCHW fp32 in [-1,1] -> NHWC fp32 in [0,1] (un-normalise + permute, src/gan_utils.py:236-241)."""
        N, C, H, W = img_nchw.shape
        assert img_nchw.dtype == torch.float32 and img_nchw.is_contiguous()
        y = self._f32(N, H, W, C)
        check(self.lib.rg_export_images_nhwc(_ptr(img_nchw), _ptr(y), N, C, H, W, self.stream), "rg_export_images_nhwc")
        return y

    # ------------------------------------------------------------------ resize-conv block (DCGANUpGenerator)
    def upconv3(self, x, cw: ConvW, bias, out_nchw=False):
        """Conv3x3(ReflectionPad(1)(bilinear x2 (x))) + bias; NHWC activation out, or the NCHW fp32 image."""
        N, H, W, Cin = x.shape
        Cout = cw.w.shape[0]
        assert tuple(cw.w.shape) == (Cout, Cin, 3, 3) and cw.w.is_contiguous() and x.is_contiguous()
        y = self._f32(N, Cout, 2 * H, 2 * W) if out_nchw else self._act(N, 2 * H, 2 * W, Cout)
        ws = self._ws(self.lib.rg_upconv3_workspace_bytes(N, H, W, Cin, Cout))
        check(self.lib.rg_upconv3_fwd(_ptr(x), _ptr(cw.w), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, int(out_nchw),
                                      self.dt, self.algo, _ptr(ws), ws.numel(), self.stream), "rg_upconv3_fwd")
        return y

    def _up_dims(self, gy, cw, gy_nchw):
        Cout, Cin = cw.w.shape[0], cw.w.shape[1]
        if gy_nchw:
            N, C, H2, W2 = gy.shape
            assert gy.dtype == torch.float32
        else:
            N, H2, W2, C = gy.shape
        assert C == Cout and gy.is_contiguous()
        return N, H2 // 2, W2 // 2, Cin, Cout

    def upconv3_bwd_data(self, gy, cw: ConvW, gy_nchw=False):
        N, H, W, Cin, Cout = self._up_dims(gy, cw, gy_nchw)
        gx = self._act(N, H, W, Cin)
        ws = self._ws(self.lib.rg_upconv3_workspace_bytes(N, H, W, Cin, Cout))
        check(self.lib.rg_upconv3_bwd_data(_ptr(gy), int(gy_nchw), _ptr(cw.w), _ptr(gx), N, H, W, Cin, Cout, self.dt,
                                           self.algo, _ptr(ws), ws.numel(), self.stream), "rg_upconv3_bwd_data")
        return gx

    def upconv3_wgrad(self, gy, x, cw: ConvW, accumulate: bool, gy_nchw=False):
        N, H, W, Cin, Cout = self._up_dims(gy, cw, gy_nchw)
        assert tuple(x.shape) == (N, H, W, Cin) and x.is_contiguous() and cw.dw.is_contiguous()
        ws = self._ws(self.lib.rg_upconv3_workspace_bytes(N, H, W, Cin, Cout))
        check(self.lib.rg_upconv3_wgrad(_ptr(gy), int(gy_nchw), _ptr(x), _ptr(cw.dw), N, H, W, Cin, Cout, self.dt,
                                        self.algo, int(accumulate), _ptr(ws), ws.numel(), self.stream), "rg_upconv3_wgrad")

    def first_down(self, x_nchw, cw: ConvW, bias, slope: float, out=None):
        """out: (activation view, sign-bit view or None) to write into -- a batch slice of a larger tensor (the D step runs
        D(real) and D(fake) as one double batch from layer 1 on); the sign bits then stay with the caller."""
        N, I, H, W = x_nchw.shape
        O = cw.w.shape[0]
        assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and cw.w.shape[1] == I
        if out is not None:
            y, bits = out
            assert y.shape == (N, H // 2, W // 2, O) and y.is_contiguous() and y.dtype == self.act_dtype
            if bits is not None:
                check(self.lib.rg_first_down_bits(_ptr(x_nchw), _ptr(cw.w), _ptr(bias), _ptr(y), _ptr(bits), N, H, W, I, O,
                                                  float(slope), self.dt, self.stream), "rg_first_down_bits")
            else:
                check(self.lib.rg_first_down(_ptr(x_nchw), _ptr(cw.w), _ptr(bias), _ptr(y), N, H, W, I, O, float(slope),
                                             self.dt, self.stream), "rg_first_down")
            return y
        y = self._act(N, H // 2, W // 2, O)
        # Discriminator layer 0 (slope != 1): the kernel also writes the packed sign bits of its output, one uint64 per
        # pixel, when the data-gradient conv of layer 1 can take its fused LeakyReLU mask in that form (conv_up below
        # picks them up from the tensor: 8 B instead of 128 B per pixel, and the patch-resident kernel).
        # (the caller of layer 1's data gradient decides by rg_conv_up_maskbits_supported with ITS channel counts whether
        # the bits are used; a 64 -> 128 second layer is the only shape that has the kernel)
        if (self.dt != RG_F32 and O == 64 and slope != 1.0 and H % 4 == 0 and W % 4 == 0 and
                self.lib.rg_conv_up_maskbits_supported(N, H // 4, W // 4, 128, 64, self.dt, self.algo)):
            bits = torch.empty((N, H // 2, W // 2), dtype=torch.int64, device=self.device)
            check(self.lib.rg_first_down_bits(_ptr(x_nchw), _ptr(cw.w), _ptr(bias), _ptr(y), _ptr(bits), N, H, W, I, O,
                                              float(slope), self.dt, self.stream), "rg_first_down_bits")
            y._rg_sign_bits = bits
            return y
        check(self.lib.rg_first_down(_ptr(x_nchw), _ptr(cw.w), _ptr(bias), _ptr(y), N, H, W, I, O, float(slope),
                                     self.dt, self.stream), "rg_first_down")
        return y

    def first_down_tangent(self, v_nchw, cw: ConvW, a0, slope: float):
        """lrelu'(a0) * conv(v) (no bias): the tangent of discriminator layer 0.  One kernel when a0 carries its packed sign
        bits (first_down), else the conv followed by a masking pass."""
        N, I, H, W = v_nchw.shape
        O = cw.w.shape[0]
        bits = getattr(a0, "_rg_sign_bits", None)
        if bits is not None and self.lib.rg_first_down_masked_supported(H, W, I, O, self.dt):
            y = self._act(N, H // 2, W // 2, O)
            check(self.lib.rg_first_down_masked(_ptr(v_nchw), _ptr(cw.w), _ptr(y), _ptr(bits), float(slope), N, H, W, I, O,
                                                self.dt, self.stream), "rg_first_down_masked")
            return y
        return self.lrelu_bwd(self.first_down(v_nchw, cw, None, 1.0), a0, slope)

    def sign_pack(self, a):
        """Packed sign bits (uint64 per pixel, bit c = a[pixel][c] > 0) of a bf16 activation [..., 64]."""
        assert a.dtype == self.h16 and a.shape[-1] == 64 and a.is_contiguous()
        bits = torch.empty(a.shape[:-1], dtype=torch.int64, device=a.device)
        check(self.lib.rg_sign_pack(_ptr(a), _ptr(bits), a.numel() // 64, 64, self.dt, self.stream), "rg_sign_pack")
        return bits

    def last_up_bn(self, z, partials, bn, slope: float, cw: ConvW, bias, tanh: bool, update_running=True):
        """last_up(lrelu(bn_train(z))) without materialising the normalised activation (a forward that keeps nothing for a
        backward pass): statistics from the conv epilogue's partial sums (or one reduction pass), BatchNorm + LeakyReLU applied
        while last_up stages its input rows.  None when this shape has no fused form."""
        N, Ho, Wo, O = z.shape
        I = cw.w.shape[1]
        if self.stat_reduce is not None or not self.lib.rg_last_up_pre_supported(Wo, O, I, self.dt):
            return None
        M, C = self._mc(z)
        rm, rv, nbt = (bn.running_mean, bn.running_var, bn.nbt) if update_running else (None, None, None)
        if partials is not None:
            mean, invstd = self._f32(C), self._f32(C)
            ws = self._ws(32 * 2 * C * 4)
            check(self.lib.rg_bn_finalize_partials(_ptr(partials), partials.shape[0], M, C, float(bn.eps), float(bn.momentum),
                                                   _ptr(mean), _ptr(invstd), _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(ws),
                                                   ws.numel(), self.stream), "rg_bn_finalize_partials")
        else:
            mean, invstd = self.bn_stats_finalize(z, bn.eps, bn.momentum, rm, rv, nbt)
        y = self._f32(N, I, 2 * Ho, 2 * Wo)
        check(self.lib.rg_last_up_pre(_ptr(z), _ptr(cw.w), _ptr(bias), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(bn.gamma),
                                      _ptr(bn.beta), float(slope), N, Ho, Wo, O, I, int(tanh), self.dt, self.stream),
              "rg_last_up_pre")
        return y

    def last_up_bn2(self, z, partials, bn, slope: float, cw: ConvW, bias, tanh: bool, update_running=True):
        """last_up_bn for a double batch whose halves are two separate forward calls (own statistics, running statistics
        updated by the first half first): z [2n, ...], partials = conv_up's class-major column sums or None."""
        N2, Ho, Wo, O = z.shape
        I = cw.w.shape[1]
        n = N2 // 2
        if self.stat_reduce is not None or N2 % 2 or not self.lib.rg_last_up_pre_supported(Wo, O, I, self.dt):
            return None
        M2, C = self._mc(z)
        M = M2 // 2
        rm, rv, nbt = (bn.running_mean, bn.running_var, bn.nbt) if update_running else (None, None, None)
        if partials is not None:
            mean, invstd = self._f32(2, C), self._f32(2, C)
            ws = self._ws(2 * 32 * 2 * C * 4)
            check(self.lib.rg_bn_finalize_partials_g2(_ptr(partials), partials.shape[0] // 2, 4, M, C, float(bn.eps),
                                                      float(bn.momentum), _ptr(mean), _ptr(invstd), _ptr(rm), _ptr(rv),
                                                      _ptr(nbt), _ptr(ws), ws.numel(), self.stream),
                  "rg_bn_finalize_partials_g2")
        else:
            st = [self.bn_stats_finalize(z[h * n:(h + 1) * n], bn.eps, bn.momentum, rm, rv, nbt) for h in range(2)]
            mean, invstd = torch.stack([st[0][0], st[1][0]]), torch.stack([st[0][1], st[1][1]])
        y = self._f32(N2, I, 2 * Ho, 2 * Wo)
        for h in range(2):
            check(self.lib.rg_last_up_pre(_ptr(z[h * n:(h + 1) * n]), _ptr(cw.w), _ptr(bias), _ptr(y[h * n:(h + 1) * n]),
                                          _ptr(mean[h]), _ptr(invstd[h]), _ptr(bn.gamma), _ptr(bn.beta), float(slope), n, Ho, Wo,
                                          O, I, int(tanh), self.dt, self.stream), "rg_last_up_pre")
        return y

    def last_up(self, x, cw: ConvW, bias, tanh: bool):
        N, Ho, Wo, O = x.shape
        I = cw.w.shape[1]
        assert cw.w.shape[0] == O and x.is_contiguous()
        y = self._f32(N, I, 2 * Ho, 2 * Wo)
        check(self.lib.rg_last_up(_ptr(x), _ptr(cw.w), _ptr(bias), _ptr(y), N, Ho, Wo, O, I, int(tanh), self.dt,
                                  self.stream), "rg_last_up")
        return y

    def last_up_post(self, x, cw: ConvW, tanh_img=None):
        """last_up (no bias, no activation) with the first consumer's pass fused into the store phase (rg_last_up_post):
        y *= 1 - tanh_img^2 when tanh_img is given, and per-workgroup partial sums [blocks][4] (channel 0..2 sums, sum of
        squares) for parts_chan_sum / gp_coef_parts.  Returns (y, parts), or None when the shape has no such kernel."""
        N, Ho, Wo, O = x.shape
        I = cw.w.shape[1]
        if not self.fuse_input_post:
            return None
        nb = self.lib.rg_last_up_post_blocks(N, Ho, Wo, O, I, self.dt)
        if nb <= 0:
            return None
        assert cw.w.shape[0] == O and x.is_contiguous()
        assert tanh_img is None or (tanh_img.shape == (N, I, 2 * Ho, 2 * Wo) and tanh_img.is_contiguous()
                                    and tanh_img.dtype == torch.float32)
        y = self._f32(N, I, 2 * Ho, 2 * Wo)
        parts = self._f32(nb, 4)
        check(self.lib.rg_last_up_post(_ptr(x), _ptr(cw.w), _ptr(y), N, Ho, Wo, O, I, self.dt, _ptr(tanh_img), _ptr(parts),
                                       self.stream), "rg_last_up_post")
        return y, parts

    def parts_chan_sum(self, parts, out, accumulate: bool):
        check(self.lib.rg_last_up_part_chan_sum(_ptr(parts), parts.shape[0], _ptr(out), int(accumulate), self.stream),
              "rg_last_up_part_chan_sum")

    def gp_coef_parts(self, parts, lambd: float):
        loss, coef = self._f32(1), self._f32(1)
        check(self.lib.rg_gp_coef_parts_scaled(_ptr(parts), parts.shape[0], None, _ptr(loss), _ptr(coef), float(lambd),
                                               float(self.gp_seed_scale), float(self.gp_tangent_scale), self.stream),
              "rg_gp_coef_parts_scaled")
        return loss, coef

    def skinny_wgrad(self, low, high_nchw, dw, accumulate: bool, dbias=None, dbias_accumulate=False):
        """Weight gradient of an image-side layer.  dbias (optional): the layer's bias gradient = the column sums of `low`, formed
        by the same kernel where it can (rg_skinny_wgrad_bias); returns True when dbias was written (or left as deferred
        partials), False when the caller still has to call col_sum."""
        N, Ho, Wo, O = low.shape
        I = high_nchw.shape[1]
        assert high_nchw.dtype == torch.float32 and high_nchw.is_contiguous() and low.is_contiguous()
        if not self.skinny_bias:
            dbias = None
        nb = self.lib.rg_skinny_wgrad_workspace_bytes(N, Ho, Wo, O, I)
        cw = self._skinny_defer.get(dw.data_ptr()) if self._skinny_defer else None
        if cw is not None and cw.pending_slabs is None and accumulate:
            cw = None          # the first contribution had no slab form (a small image) and wrote dw: this one adds to it
        if cw is not None and self.dt != RG_F32 and self.stat_reduce is None and not self._in_side:
            # the optimizer step follows at once (the train_op runner registered the layer): the per-workgroup partial gradients
            # stay in a buffer of the layer's own -- a second contribution (accumulate) behind the first one's -- and Adam sums them
            prev = cw.pending_slabs
            if (prev is None) == bool(accumulate):
                raise RuntimeError("rna_gan_amd: deferred image-side weight gradient: contributions out of order")
            have = 0 if prev is None else prev[1]
            if cw._slab_ws is None or cw._slab_ws.numel() < 2 * nb + 4096:
                if prev is not None:
                    raise RuntimeError("rna_gan_amd: deferred image-side weight gradient: the slab buffer grew between two contributions")
                # persistent: graphs hold the addresses -- outgrown buffers are retired, not freed (see _wgrad_slabs)
                self._ws_retired.extend(b for b in (cw._slab_ws, cw._bias_ws) if b is not None)
                cw._slab_ws = torch.empty(2 * nb + 4096, dtype=torch.uint8, device=self.device)
                cw._bias_ws = torch.empty(2 * 1024 * O, dtype=torch.float32, device=self.device)
            want_bias = (dbias is not None and cw.bias is not None and cw.dbias is not None and
                         dbias.data_ptr() == cw.dbias.data_ptr() and O == 64)
            bprev = cw.pending_bias
            if want_bias and (bprev is None) == bool(dbias_accumulate):
                want_bias = False                        # (a bias contribution out of order: the caller's col_sum handles it)
            bhave = 0 if bprev is None else bprev[1]
            off = have * O * 48 * 4
            ns, bdone = ctypes.c_int(0), ctypes.c_int(0)
            check(self.lib.rg_skinny_wgrad_slabs(_ptr(low), _ptr(high_nchw), N, Ho, Wo, O, I, self.dt, cw._slab_ws.data_ptr() + off,
                                                 cw._slab_ws.numel() - off, ctypes.addressof(ns),
                                                 cw._bias_ws.data_ptr() + bhave * O * 4 if want_bias else 0,
                                                 ctypes.addressof(bdone), self.stream), "rg_skinny_wgrad_slabs")
            if ns.value > 0:
                cw.pending_slabs = (cw._slab_ws, have + int(ns.value), RG_F32)
                if want_bias and bdone.value:
                    if bhave + ns.value > 2 * 1024:
                        raise RuntimeError("rna_gan_amd: deferred bias-gradient partials exceed their buffer")
                    cw.pending_bias = (cw._bias_ws, bhave + int(ns.value))
                    return True
                return False
            if prev is not None:
                raise RuntimeError("rna_gan_amd: deferred image-side weight gradient: the second contribution has no slab form")
        ws = self._ws(nb)
        bdone = ctypes.c_int(0)
        check(self.lib.rg_skinny_wgrad_bias(_ptr(low), _ptr(high_nchw), _ptr(dw), _ptr(dbias), N, Ho, Wo, O, I, self.dt,
                                            int(accumulate), int(dbias_accumulate), _ptr(ws), ws.numel(), ctypes.addressof(bdone),
                                            self.stream), "rg_skinny_wgrad_bias")
        return bool(bdone.value)

    # ------------------------------------------------------------------ G.0 / head
    def g0_pack(self, cw: ConvW):
        """bf16 GEMM operand image [(tap, c)][E] of the generator's first weight (rebuilt when the master changed)."""
        E, C = cw.w.shape[0], cw.w.shape[1]
        if True:
            if cw.packs is None or cw.packs_version != cw.version:
                if cw.packs is None:
                    cw.packs = (torch.empty((16 * C, E), dtype=self.h16, device=self.device),)
                if cw.shadow is not None and cw.shadow_version == cw.version and self.pack_from_shadow:
                    check(self.lib.rg_pack_g0_weight_from_bf16(_ptr(cw.shadow), _ptr(cw.packs[0]), E, C, self.stream),
                          "rg_pack_g0_weight_from_bf16")
                else:
                    check(self.lib.rg_pack_g0_weight(_ptr(cw.w), _ptr(cw.packs[0]), E, C, self.H16, self.stream),
                          "rg_pack_g0_weight")
                cw.packs_version = cw.version
                if cw.shadow is not None:
                    cw.shadow_version = cw.version

    def g0_fwd(self, z, cw: ConvW):
        N, E = z.shape
        C = cw.w.shape[1]
        assert z.dtype == torch.float32 and z.is_contiguous() and cw.w.shape[0] == E
        wp = None
        if self.dt != RG_F32:
            self.g0_pack(cw)
            wp = cw.packs[0]
        y = self._act(N, 4, 4, C)
        ws = self._ws(self.lib.rg_g0_workspace_bytes(N, E, C, self.dt, self.algo))
        check(self.lib.rg_g0_fwd(_ptr(z), _ptr(cw.w), _ptr(wp), _ptr(y), N, E, C, self.dt, self.algo, _ptr(ws),
                                 ws.numel(), self.stream), "rg_g0_fwd")
        return y

    def g0_wgrad_deferred(self, z, gy, cw: ConvW, accumulate: bool):
        """Generator layer 0's weight gradient when the optimizer step follows immediately (cw.fuse_step, set by the train_op
        runner): nothing is computed here -- the operands are left on the handle and rna_gan_amd.optim.Adam forms the
        gradient inside its fused step for this tensor (rg_g0_wgrad_adam).  False: compute dw now (g0_wgrad)."""
        if not cw.fuse_step or accumulate or self.stat_reduce is not None:
            return False
        N, E = z.shape
        C = gy.shape[3]
        if not self.lib.rg_g0_wgrad_adam_supported(N, E, C, self.dt) or not z.is_contiguous() or not gy.is_contiguous():
            if cw.factor_stage is not None:
                raise RuntimeError("g0_wgrad_deferred: the data-parallel runner announced gathered factors for a shape the "
                                   "fused kernel does not take")
            return False
        if cw.factor_stage is not None:
            # data parallel: the operands go into this rank's slices of the gathered buffers (persistent memory: the
            # collective reads them after this graph's pool may have been handed to the next graph); losses.flush hands the
            # GATHERED factors to the optimizer
            zs, gs = cw.factor_stage
            zs.copy_(z)
            gs.copy_(gy.reshape(gs.shape))
            cw.pending_wgrad = "staged"
            return True
        cw.pending_wgrad = (z, gy, self.dt)
        return True

    def g0_wgrad(self, z, gy, dw, accumulate: bool):
        N, E = z.shape
        C = gy.shape[3]
        ws = self._ws(self.lib.rg_g0_workspace_bytes(N, E, C, self.dt, self.algo))
        check(self.lib.rg_g0_wgrad(_ptr(z), _ptr(gy), _ptr(dw), N, E, C, self.dt, int(accumulate), self.algo,
                                   _ptr(ws), ws.numel(), self.stream), "rg_g0_wgrad")

    def g0_bwd_data(self, gz0, cw: ConvW):
        """d/dz of the generator's first layer, gin[n][e] = sum_{tap,c} gz0[n][tap][c] * w[e][c][tap].  Off the
        reference's path (its noise never requires grad): one fp32 functor GEMM (rg_linear_affine_act) on the master
        weight viewed as [E][C*16]; the (tap, c) -> (c, tap) re-order of the small gz0 is host-side plumbing."""
        N, C = gz0.shape[0], gz0.shape[3]
        E = cw.w.shape[0]
        assert tuple(cw.w.shape) == (E, C, 4, 4) and cw.w.is_contiguous()
        x = gz0.float().reshape(N, 16, C).permute(0, 2, 1).contiguous().reshape(N, C * 16)
        return self.linear_affine_act(x, cw.w.reshape(E, C * 16), None, None, 1.0)

    def head_fwd(self, a, cw: ConvW, slope: float):
        N, C = a.shape[0], a.shape[3]
        assert a.shape[1] == 4 and a.shape[2] == 4 and a.is_contiguous()
        h, out = self._f32(N), self._f32(N)
        check(self.lib.rg_head_fwd(_ptr(a), _ptr(cw.w), _ptr(h), _ptr(out), N, C, float(slope), self.dt,
                                   self.stream), "rg_head_fwd")
        return h, out

    def head_grad(self, h, coef: float, slope: float):
        gh = torch.empty_like(h)
        check(self.lib.rg_head_grad(_ptr(h), _ptr(gh), h.numel(), float(coef), float(slope), self.stream),
              "rg_head_grad")
        return gh

    def head_bwd_data(self, gh, cw: ConvW):
        N, C = gh.numel(), cw.w.shape[1]
        ga = self._act(N, 4, 4, C)
        check(self.lib.rg_head_bwd_data(_ptr(gh), _ptr(cw.w), _ptr(ga), N, C, self.dt, self.stream),
              "rg_head_bwd_data")
        return ga

    def head_wgrad(self, gh, a, dw, accumulate: bool):
        N, C = a.shape[0], a.shape[3]
        check(self.lib.rg_head_wgrad(_ptr(gh), _ptr(a), _ptr(dw), N, C, self.dt, int(accumulate), self.stream),
              "rg_head_wgrad")

    # ------------------------------------------------------------------ batch norm
    def _mc(self, z):
        C = z.shape[-1]
        return z.numel() // C, C

    def bn_stats(self, z):
        M, C = self._mc(z)
        s, ss = self._f32(C), self._f32(C)
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 2))
        check(self.lib.rg_bn_stats(_ptr(z), _ptr(s), _ptr(ss), M, C, self.dt, _ptr(ws), ws.numel(), self.stream),
              "rg_bn_stats")
        return s, ss

    def bn_finalize(self, s, ss, count: int, eps: float, momentum: float,
                    running_mean=None, running_var=None, nbt=None):
        C = s.numel()
        mean, invstd = self._f32(C), self._f32(C)
        check(self.lib.rg_bn_finalize(_ptr(s), _ptr(ss), int(count), C, float(eps), float(momentum), _ptr(mean),
                                      _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(nbt), self.stream),
              "rg_bn_finalize")
        return mean, invstd

    def bn_stats_finalize(self, z, eps: float, momentum: float, running_mean=None, running_var=None, nbt=None):
        """Batch statistics of z -> (mean, invstd) (+ running-statistics update) in one reduction pass."""
        M, C = self._mc(z)
        mean, invstd = self._f32(C), self._f32(C)
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 2))
        check(self.lib.rg_bn_stats_finalize(_ptr(z), M, C, float(eps), float(momentum), _ptr(mean), _ptr(invstd),
                                            _ptr(running_mean), _ptr(running_var), _ptr(nbt), self.dt, _ptr(ws),
                                            ws.numel(), self.stream), "rg_bn_stats_finalize")
        return mean, invstd

    def sign_bits_for(self, N, H, W, C0=64, C1=128):
        """An (uninitialised) packed sign-bit tensor for a [N, H, W, C0] activation when the data-gradient conv above it
        (C1 -> C0 channels) takes its LeakyReLU mask in that form (see first_down), else None.  rg_first_down_bits packs one
        uint64 per pixel: 64 channels exactly."""
        if (self.dt != RG_F32 and C0 == 64 and H % 2 == 0 and W % 2 == 0 and
                self.lib.rg_conv_up_maskbits_supported(N, H // 2, W // 2, C1, C0, self.dt, self.algo)):
            return torch.empty((N, H, W), dtype=torch.int64, device=self.device)
        return None

    def bn_forward(self, z, gamma, beta, slope: float, eps: float, momentum: float, running_mean=None,
                   running_var=None, nbt=None, partials=None, out=None):
        """Train-mode BatchNorm + LeakyReLU: (a, mean, invstd).  partials: the column sums the producing conv's
        epilogue wrote (conv_down/conv_up want_stats) -- then no statistics pass over z is needed."""
        M, C = self._mc(z)
        sl = getattr(z, "_rg_slabs", None)
        if sl is not None:                        # z is still split-K slabs: reduce + statistics + apply in one launch
            return self._bn_forward_slabs(z, sl, 1, gamma, beta, slope, eps, momentum, running_mean, running_var, nbt, out)
        if self.stat_reduce is not None:          # global statistics: local sums -> all-reduce -> finalize -> apply
            s, ss = self.bn_stats(z)
            self.stat_reduce(s); self.stat_reduce(ss)
            mean, invstd = self.bn_finalize(s, ss, M * self.stat_world, eps, momentum, running_mean, running_var, nbt)
            a = self.bn_act(z, mean, invstd, gamma, beta, slope)
            if out is not None:
                out.copy_(a)
                a = out
            return a, mean, invstd
        mean, invstd = self._f32(C), self._f32(C)
        a = out if out is not None else torch.empty_like(z)
        assert a.shape == z.shape and a.is_contiguous()
        if partials is not None:
            ws = self._ws(32 * 2 * C * 4)
            check(self.lib.rg_bn_forward_partials(_ptr(partials), partials.shape[0], _ptr(z), M, C, float(eps),
                                                  float(momentum), _ptr(gamma), _ptr(beta), float(slope), _ptr(mean),
                                                  _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                                  _ptr(a), self.dt, _ptr(ws), ws.numel(), self.stream),
                  "rg_bn_forward_partials")
            return a, mean, invstd
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 2))
        check(self.lib.rg_bn_forward(_ptr(z), M, C, float(eps), float(momentum), _ptr(gamma), _ptr(beta), float(slope),
                                     _ptr(mean), _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                     _ptr(a), self.dt, _ptr(ws), ws.numel(), self.stream), "rg_bn_forward")
        return a, mean, invstd

    def _bn_forward_slabs(self, z, sl: SlabRef, groups, gamma, beta, slope, eps, momentum, running_mean, running_var, nbt,
                          out=None):
        M2, C = self._mc(z)
        assert sl.groups == groups and M2 % groups == 0
        M = M2 // groups
        mean, invstd = (self._f32(C), self._f32(C)) if groups == 1 else (self._f32(groups, C), self._f32(groups, C))
        a = out if out is not None else torch.empty_like(z)
        assert a.shape == z.shape and a.is_contiguous()
        scratch, sync = self._sb_bufs(M, C, groups)
        self._timed("bn_split_fused", 0.0, lambda: check(
            self.lib.rg_bn_forward_slabs(_ptr(sl.ws), sl.nsplit, sl.stride, sl.dtype, _ptr(z), _ptr(a), M, C, groups, float(eps),
                                         float(momentum), _ptr(gamma), _ptr(beta), float(slope), _ptr(mean), _ptr(invstd),
                                         _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(scratch), scratch.numel(),
                                         _ptr(sync), self.stream), "rg_bn_forward_slabs"))
        del z._rg_slabs                       # z is an ordinary tensor from here on
        self._slabs_pending = None
        return a, mean, invstd

    def bn_forward2(self, z, gamma, beta, slope: float, eps: float, momentum: float, running_mean=None,
                    running_var=None, nbt=None, partials=None, nblk=1):
        """bn_forward on the two batch halves of z ([2n, ...]) in one set of launches: (a, mean[2][C], invstd[2][C]), running
        statistics updated by the first half, then the second -- exactly two bn_forward calls.  partials: the conv
        epilogue's column sums laid out [nblk][2 halves][rows] (nblk = 1: conv_down, 4: conv_up's class-major rows), or None."""
        M2, C = self._mc(z)
        M = M2 // 2
        assert M2 % 2 == 0 and self.stat_reduce is None
        sl = getattr(z, "_rg_slabs", None)
        if sl is not None:
            return self._bn_forward_slabs(z, sl, 2, gamma, beta, slope, eps, momentum, running_mean, running_var, nbt)
        mean, invstd = self._f32(2, C), self._f32(2, C)
        a = torch.empty_like(z)
        ws = self._ws(2 * self.lib.rg_colreduce_workspace_bytes(M, C, 2) + 2 * 32 * 2 * C * 4)
        G = 0 if partials is None else partials.shape[0] // 2
        check(self.lib.rg_bn_forward_g2(_ptr(partials), G, int(nblk), _ptr(z), M, C, float(eps), float(momentum), _ptr(gamma),
                                        _ptr(beta), float(slope), _ptr(mean), _ptr(invstd), _ptr(running_mean),
                                        _ptr(running_var), _ptr(nbt), _ptr(a), self.dt, _ptr(ws), ws.numel(), self.stream),
              "rg_bn_forward_g2")
        return a, mean, invstd

    def bn_act_bwd2(self, z, ga, mean, invstd, gamma, beta, slope: float, dgamma=None, dbeta=None, accumulate: bool = False):
        """bn_act_bwd on the two batch halves (mean / invstd [2][C] from bn_forward2): gz, with dgamma / dbeta summed over
        both halves."""
        M2, C = self._mc(z)
        M = M2 // 2
        assert M2 % 2 == 0 and self.stat_reduce is None and mean.shape == (2, C)
        sl = getattr(ga, "_rg_slabs", None)
        if sl is not None:
            return self._bn_act_bwd_slabs(z, ga, sl, 2, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate, False)[0]
        bp = getattr(ga, "_rg_bwd_partials", None)
        if bp is not None and bp[2] == 2:
            return self._bn_act_bwd_partials(z, ga, bp, 2, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate)[0]
        gz = torch.empty_like(z)
        s_gy, s_gyxh = self._f32(2, C), self._f32(2, C)
        ws = self._ws(2 * self.lib.rg_colreduce_workspace_bytes(M, C, 2))
        check(self.lib.rg_bn_act_bwd_g2(_ptr(z), _ptr(ga), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(gz),
                                        _ptr(s_gy), _ptr(s_gyxh), _ptr(dgamma), _ptr(dbeta), int(accumulate), M, C,
                                        float(slope), self.dt, _ptr(ws), ws.numel(), self.stream), "rg_bn_act_bwd_g2")
        return gz

    def bn_act(self, z, mean, invstd, gamma, beta, slope: float):
        M, C = self._mc(z)
        a = torch.empty_like(z)
        check(self.lib.rg_bn_act(_ptr(z), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(a), M, C,
                                 float(slope), self.dt, self.stream), "rg_bn_act")
        return a

    def _bn_act_bwd_slabs(self, z, ga, sl: SlabRef, groups, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate,
                          keep_ga, out=None):
        M2, C = self._mc(z)
        assert sl.groups == groups and ga.shape == z.shape and M2 % groups == 0
        M = M2 // groups
        gz = out if out is not None else torch.empty_like(z)
        s_gy, s_gyxh = (self._f32(C), self._f32(C)) if groups == 1 else (self._f32(groups, C), self._f32(groups, C))
        scratch, sync = self._sb_bufs(M, C, groups)
        self._timed("bn_split_fused", 0.0, lambda: check(
            self.lib.rg_bn_act_bwd_slabs(_ptr(sl.ws), sl.nsplit, sl.stride, sl.dtype, _ptr(z), _ptr(ga) if keep_ga else 0, _ptr(gz), M,
                                         C, groups, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), float(slope),
                                         _ptr(s_gy), _ptr(s_gyxh), _ptr(dgamma), _ptr(dbeta), int(accumulate), _ptr(scratch),
                                         scratch.numel(), _ptr(sync), self.stream), "rg_bn_act_bwd_slabs"))
        del ga._rg_slabs                     # (ga itself is written only with keep_ga)
        self._slabs_pending = None
        return gz, s_gy, s_gyxh

    def _bn_act_bwd_partials(self, z, ga, bp, groups, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate, out=None):
        part, nblk, _ = bp
        M2, C = self._mc(z)
        M = M2 // groups
        gz = out if out is not None else torch.empty_like(z)
        s_gy, s_gyxh = (self._f32(C), self._f32(C)) if groups == 1 else (self._f32(groups, C), self._f32(groups, C))
        ws = self._ws(groups * 32 * 2 * C * 4)
        check(self.lib.rg_bn_act_bwd_partials(_ptr(part), part.shape[0] // groups, nblk, _ptr(z), _ptr(ga), _ptr(mean),
                                              _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(gz), _ptr(s_gy), _ptr(s_gyxh),
                                              _ptr(dgamma), _ptr(dbeta), int(accumulate), M, C, groups, float(slope), self.dt,
                                              _ptr(ws), ws.numel(), self.stream), "rg_bn_act_bwd_partials")
        del ga._rg_bwd_partials
        return gz, s_gy, s_gyxh

    def bn_act_bwd(self, z, ga, mean, invstd, gamma, beta, slope: float, dgamma=None, dbeta=None,
                   accumulate: bool = False, out=None, keep_ga=True):
        """keep_ga: only meaningful when ga is still split-K slabs (conv_* with defer): also write the reduced ga tensor
        (the penalty's first backward keeps it for the double-backward pass)."""
        M, C = self._mc(z)
        sl = getattr(ga, "_rg_slabs", None)
        if sl is not None:
            return self._bn_act_bwd_slabs(z, ga, sl, 1, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate, keep_ga,
                                          out)
        bp = getattr(ga, "_rg_bwd_partials", None)
        if bp is not None and bp[2] == 1 and self.stat_reduce is None:
            return self._bn_act_bwd_partials(z, ga, bp, 1, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate, out)
        gz = out if out is not None else torch.empty_like(z)
        assert gz.shape == z.shape and gz.is_contiguous()
        s_gy, s_gyxh = self._f32(C), self._f32(C)
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 2))
        if self.stat_reduce is not None:
            check(self.lib.rg_bn_bwd_sums(_ptr(z), _ptr(ga), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(s_gy),
                                          _ptr(s_gyxh), M, C, float(slope), self.dt, _ptr(ws), ws.numel(), self.stream),
                  "rg_bn_bwd_sums")
            if dgamma is not None:      # parameter gradients: this rank's contribution (C-length vectors)
                if accumulate:
                    dgamma.add_(s_gyxh); dbeta.add_(s_gy)
                else:
                    dgamma.copy_(s_gyxh); dbeta.copy_(s_gy)
            self.stat_reduce(s_gy); self.stat_reduce(s_gyxh)
            check(self.lib.rg_bn_bwd_apply(_ptr(z), _ptr(ga), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(s_gy),
                                           _ptr(s_gyxh), _ptr(gz), M, C, M * self.stat_world, float(slope), self.dt,
                                           self.stream), "rg_bn_bwd_apply")
            return gz, s_gy, s_gyxh
        check(self.lib.rg_bn_act_bwd(_ptr(z), _ptr(ga), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(gz),
                                     _ptr(s_gy), _ptr(s_gyxh), _ptr(dgamma), _ptr(dbeta), int(accumulate), M, C,
                                     float(slope), self.dt, _ptr(ws), ws.numel(), self.stream), "rg_bn_act_bwd")
        return gz, s_gy, s_gyxh

    def bn_tangent(self, z, zt, mean, invstd, gamma, beta, slope: float):
        M, C = self._mc(z)
        at = torch.empty_like(z)
        s_zt, s_xhzt = self._f32(C), self._f32(C)
        sl = getattr(zt, "_rg_slabs", None)
        if sl is not None:                    # zt is still split-K slabs: reduce + tangent sums + apply in one launch
            assert sl.groups == 1 and zt.shape == z.shape
            scratch, sync = self._sb_bufs(M, C, 1)
            self._timed("bn_split_fused", 0.0, lambda: check(
                self.lib.rg_bn_tangent_slabs(_ptr(sl.ws), sl.nsplit, sl.stride, sl.dtype, _ptr(z), _ptr(zt), _ptr(at), M, C, _ptr(mean),
                                             _ptr(invstd), _ptr(gamma), _ptr(beta), float(slope), _ptr(s_zt), _ptr(s_xhzt),
                                             _ptr(scratch), scratch.numel(), _ptr(sync), self.stream), "rg_bn_tangent_slabs"))
            del zt._rg_slabs
            self._slabs_pending = None
            return at, s_zt, s_xhzt
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 2))
        if self.stat_reduce is not None:
            check(self.lib.rg_bn_tangent_sums(_ptr(z), _ptr(zt), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                                              _ptr(s_zt), _ptr(s_xhzt), M, C, float(slope), self.dt, _ptr(ws), ws.numel(),
                                              self.stream), "rg_bn_tangent_sums")
            self.stat_reduce(s_zt); self.stat_reduce(s_xhzt)
            check(self.lib.rg_bn_tangent_apply(_ptr(z), _ptr(zt), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                                               _ptr(s_zt), _ptr(s_xhzt), _ptr(at), M, C, M * self.stat_world,
                                               float(slope), self.dt, self.stream), "rg_bn_tangent_apply")
            return at, s_zt, s_xhzt
        check(self.lib.rg_bn_tangent(_ptr(z), _ptr(zt), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(at),
                                     _ptr(s_zt), _ptr(s_xhzt), M, C, float(slope), self.dt, _ptr(ws), ws.numel(),
                                     self.stream), "rg_bn_tangent")
        return at, s_zt, s_xhzt

    def bn_double_bwd(self, z, qa, zt, ga1, mean, invstd, gamma, beta, slope: float, s_gy, s_gyxh, s_zt, s_xhzt,
                      dgamma, dbeta, accumulate: bool):
        M, C = self._mc(z)
        pz = torch.empty_like(z)
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 3))
        if self.stat_reduce is not None:       # s_gy .. s_xhzt are global sums already (bn_act_bwd / bn_tangent)
            raw_local = self._f32(3, C)
            if qa is None:
                raw_local.zero_()
            check(self.lib.rg_bn_dbl_sums(_ptr(z), _ptr(qa), _ptr(zt), _ptr(ga1), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                          _ptr(beta), _ptr(raw_local), M, C, float(slope), self.dt, _ptr(ws), ws.numel(),
                                          self.stream), "rg_bn_dbl_sums")
            raw_global = raw_local.clone()
            self.stat_reduce(raw_global)
            check(self.lib.rg_bn_dbl_apply(_ptr(z), _ptr(qa), _ptr(zt), _ptr(ga1), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                           _ptr(beta), _ptr(s_gy), _ptr(s_gyxh), _ptr(s_zt), _ptr(s_xhzt), _ptr(raw_global),
                                           _ptr(raw_local), _ptr(pz), _ptr(dgamma), _ptr(dbeta), int(accumulate), M, C,
                                           M * self.stat_world, float(slope), self.dt, _ptr(ws), ws.numel(),
                                           self.stream), "rg_bn_dbl_apply")
            return pz
        check(self.lib.rg_bn_double_bwd(_ptr(z), _ptr(qa), _ptr(zt), _ptr(ga1), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                        _ptr(beta), _ptr(s_gy), _ptr(s_gyxh), _ptr(s_zt), _ptr(s_xhzt), _ptr(pz),
                                        _ptr(dgamma), _ptr(dbeta), int(accumulate), M, C, float(slope), self.dt,
                                        _ptr(ws), ws.numel(), self.stream), "rg_bn_double_bwd")
        return pz

    # ------------------------------------------------------------------ pointwise / reductions
    def lrelu_bwd(self, g, a, slope: float):
        out = torch.empty_like(a)
        check(self.lib.rg_lrelu_bwd(_ptr(g), _ptr(a), _ptr(out), a.numel(), float(slope), self.dt, self.stream),
              "rg_lrelu_bwd")
        return out

    def col_sum(self, g, out, accumulate: bool):
        M, C = self._mc(g)
        ws = self._ws(self.lib.rg_colreduce_workspace_bytes(M, C, 1))
        check(self.lib.rg_col_sum(_ptr(g), _ptr(out), M, C, self.dt, int(accumulate), _ptr(ws), ws.numel(),
                                  self.stream), "rg_col_sum")

    def tanh_bwd(self, gy_nchw, y_nchw):
        gz = torch.empty_like(y_nchw)
        check(self.lib.rg_tanh_bwd(_ptr(gy_nchw), _ptr(y_nchw), _ptr(gz), y_nchw.numel(), self.stream), "rg_tanh_bwd")
        return gz

    def nchw_chan_sum(self, g_nchw, out, accumulate: bool):
        N, C, H, W = g_nchw.shape
        ws = self._ws(C * 256 * 4)
        check(self.lib.rg_nchw_chan_sum(_ptr(g_nchw), _ptr(out), N, C, H * W, int(accumulate), _ptr(ws), ws.numel(),
                                        self.stream), "rg_nchw_chan_sum")

    def interp(self, real, fake, eps):
        """eps: python float, or a 1-element device tensor (graph-replayable form)."""
        assert real.dtype == torch.float32 and real.is_contiguous() and fake.is_contiguous()
        out = torch.empty_like(real)
        if torch.is_tensor(eps):
            check(self.lib.rg_interp_dev(_ptr(real), _ptr(fake), _ptr(out), real.numel(), _ptr(eps), self.stream),
                  "rg_interp_dev")
        else:
            check(self.lib.rg_interp(_ptr(real), _ptr(fake), _ptr(out), real.numel(), float(eps), self.stream),
                  "rg_interp")
        return out

    def sqnorm(self, x):
        out = self._f32(1)
        ws = self._ws(self.lib.rg_reduce_workspace_bytes(x.numel()))
        check(self.lib.rg_sqnorm(_ptr(x), _ptr(out), x.numel(), _ptr(ws), ws.numel(), self.stream), "rg_sqnorm")
        return out

    def gp_coef(self, sq, lambd: float):
        loss, coef = self._f32(1), self._f32(1)
        check(self.lib.rg_gp_coef_scaled(_ptr(sq), _ptr(loss), _ptr(coef), float(lambd), float(self.gp_seed_scale),
                                         float(self.gp_tangent_scale), self.stream), "rg_gp_coef_scaled")
        return loss, coef

    def scale_by(self, x, coef_dev):
        out = torch.empty_like(x)
        check(self.lib.rg_scale_by(_ptr(x), _ptr(coef_dev), _ptr(out), x.numel(), self.stream), "rg_scale_by")
        return out

    def mean_diff(self, a, b=None, sign: float = 1.0):
        out = self._f32(1)
        check(self.lib.rg_mean_diff(_ptr(a), _ptr(b), _ptr(out), a.numel(), float(sign), self.stream), "rg_mean_diff")
        return out

    def stat_allreduce(self, t):
        """SUM over the ranks when statistics are synchronised (the penalty's squared norm), identity otherwise."""
        if self.stat_reduce is not None:
            self.stat_reduce(t)
        return t

    def latent_prep(self, u, z):
        N, E = u.shape
        out = torch.empty_like(u)
        if self.stat_reduce is not None:
            s, ss = self._f32(E), self._f32(E)
            check(self.lib.rg_latent_stats(_ptr(u), _ptr(z), _ptr(s), _ptr(ss), N, E, self.stream), "rg_latent_stats")
            self.stat_reduce(s); self.stat_reduce(ss)
            check(self.lib.rg_latent_apply(_ptr(u), _ptr(z), _ptr(s), _ptr(ss), _ptr(out), N, E, N * self.stat_world,
                                           self.stream), "rg_latent_apply")
            return out
        check(self.lib.rg_latent_prep(_ptr(u), _ptr(z), _ptr(out), N, E, self.stream), "rg_latent_prep")
        return out

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, p, g, m, v, step: int, lr: float, b1: float, b2: float, eps: float):
        check(self.lib.rg_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(step), float(lr), float(b1),
                                    float(b2), float(eps), self.stream), "rg_adam_step")

    def clamp_(self, p, lo: float, hi: float):
        check(self.lib.rg_clamp(_ptr(p), p.numel(), float(lo), float(hi), self.stream), "rg_clamp")

    # ------------------------------------------------------------------ dense (betaVAE encoder)
    def linear_affine_act(self, x, w, scale, shift, slope: float, wp=None):
        M, K = x.shape
        Nout = w.shape[0]
        assert x.dtype == torch.float32 and x.is_contiguous()
        if w.dim() != 2 or w.shape[1] != K:       # the kernels index w[j * K + k]: a mismatch reads out of bounds
            raise RuntimeError("linear_affine_act: x is (%d, %d) but the weight is %s (nn.Linear layout [out][in] expected)"
                               % (M, K, tuple(w.shape)))
        y = self._f32(M, Nout)
        algo = self.algo if wp is not None else _abi.ALGO_GENERIC
        ws = self._ws(self.lib.rg_linear_workspace_bytes(M, K, Nout, algo))
        check(self.lib.rg_linear_affine_act(_ptr(x), K, _ptr(w), _ptr(wp), _ptr(scale), _ptr(shift), _ptr(y), Nout, M,
                                            K, Nout, float(slope), algo, _ptr(ws), ws.numel(), self.stream),
              "rg_linear_affine_act")
        return y

    def pack_linear(self, w):
        Nout, K = w.shape
        Kp = (K + 63) // 64 * 64
        Np = (Nout + 127) // 128 * 128
        wp = torch.empty((Np, Kp), dtype=self.h16, device=self.device)
        check(self.lib.rg_pack_linear_weight(_ptr(w), _ptr(wp), Nout, K, Np, Kp, self.stream),
              "rg_pack_linear_weight")
        return wp

    def selftest(self):
        d = torch.zeros(2, dtype=torch.int32, device=self.device)
        check(self.lib.rg_selftest_layouts(_ptr(d), self.stream), "rg_selftest_layouts")
        return [int(v) for v in d.cpu()]
