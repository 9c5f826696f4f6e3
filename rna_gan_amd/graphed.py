"""HIP-graph execution of a train_op.

One train_op is ~180 kernel launches of 5-150 us each; issued from Python they cost more host time
than device time.  A train_op's launch sequence is static for fixed shapes, so after two eager
executions it is captured once (torch.cuda.CUDAGraph = hipGraph on ROCm; our kernels are enqueued on
the capturing stream like any other) and replayed afterwards.  Everything that varies per step is
read from device memory inside the graph: inputs are copied into static buffers before the replay,
the Adam step counter / bias corrections and the penalty's eps are device scalars.
"""
from __future__ import annotations

import math
import os

import torch

ENABLED = os.environ.get("RNAGAN_GRAPHS", "1") != "0"
WARMUP_CALLS = 2
_captured = []
_POOL = None


def _pool():
    """One memory pool for all step graphs: they never run concurrently, so they share activations.
    Consequence (torch's rule for shared pools): a graph's replay may overwrite what an EARLIER-REPLAYED graph left in the
    pool -- its intermediates were free blocks when the other graph was captured.  Tensors handed from one graph to the next
    are read before anything else replays; small outputs the CALLER reads later (the loss scalars) are therefore copied,
    inside the captured graph, into buffers that live outside the pool (StepGraph._stable)."""
    global _POOL
    if os.environ.get("RNAGAN_GRAPH_POOL", "1") == "0":
        return None                              # diagnostic: a private pool per graph
    if _POOL is None:
        _POOL = torch.cuda.graph_pool_handle()
    return _POOL


def mark_static(t):
    """Declare ``t`` a persistent input buffer: the caller passes this same tensor object to every call of a train_op and
    updates its contents in place.  Step graphs then read it directly instead of copying it into a buffer of their own."""
    t._rg_static = True
    return t


class StepGraph:
    """fn(*static_inputs) -> 1-element loss tensor, replayed from a captured graph.

    ``modules``: HIP modules used by fn.  Weight re-packing is decided by Python-side version counters
    that are frozen inside a graph, so the caller keys graphs by the staleness pattern of the modules
    (losses._Runner) and this class replays the counters' evolution after each replay: every module
    used ends up freshly packed, then the ``stepped`` modules (whose optimizer ran) become stale;
    ``optimizers``: rna_gan_amd.optim.Adam instances stepped inside fn (host step mirror)."""

    def __init__(self, fn, example_inputs, modules, optimizers, stepped=()):
        self.fn = fn
        # An input tensor marked ``t._rg_static = True`` is a buffer the caller reuses for every call (contents updated in
        # place, e.g. the trainer's normalised image batch): it is captured as is and never copied (mark_static()).
        self.static_in = [t if getattr(t, "_rg_static", False) else t.detach().clone() for t in example_inputs]
        self.modules = modules
        self.optimizers = optimizers
        self.stepped = list(stepped)
        self.graph = None
        self.static_out = None
        self.calls = 0
        self.failed = False
        self._out_spec = None          # (shape, dtype, device) of every tensor the last eager run returned
        self._rebuilt = {}             # fp32-storage modules: {id(module): [ConvW whose operand images the captured function rebuilds]}

    STABLE_NUMEL = 4096                # outputs up to this size get a home outside the graph pool

    @staticmethod
    def _flat(out):
        return list(out) if isinstance(out, (tuple, list)) else [out]

    def _note_outputs(self, out):
        self._out_spec = [(tuple(t.shape), t.dtype, t.device) if torch.is_tensor(t) else None for t in self._flat(out)]
        return out

    def _stable(self, out, bufs):
        """Inside the capture: copy every small output into its persistent buffer (allocated before the capture, i.e. NOT
        in the shared graph pool) and return the buffers in its place: a loss scalar stays readable after other graphs of
        the pool have replayed (bench.py reads an iteration's three losses after all three train_ops)."""
        res = []
        for t, b in zip(self._flat(out), bufs):
            if b is not None:
                b.copy_(t)
                res.append(b)
            else:
                res.append(t)
        return type(out)(res) if isinstance(out, (tuple, list)) else res[0]

    def _run(self):
        return self.fn(*self.static_in)

    def __call__(self, *inputs, allow_capture=True):
        """allow_capture=False: run eagerly this time and do not capture yet (the function reads tensors whose
        addresses are not stable yet, e.g. the outputs of another StepGraph that is still in its eager phase)."""
        for i, (s, t) in enumerate(zip(self.static_in, inputs)):
            if s.shape != t.shape:
                raise RuntimeError("StepGraph: input shape changed")
            if s is not t:
                if self.graph is None and getattr(t, "_rg_static", False) and t.dtype == s.dtype:
                    self.static_in[i] = t          # became a persistent buffer meanwhile (another graph's output): adopt it
                else:
                    s.copy_(t, non_blocking=True)
        self.calls += 1
        if self.failed or self.calls <= WARMUP_CALLS or (self.graph is None and not allow_capture):
            return self._note_outputs(self._run())
        if self.graph is None:
            try:
                torch.cuda.synchronize()
                bufs = [torch.empty(sp[0], dtype=sp[1], device=sp[2])
                        if sp is not None and math.prod(sp[0]) <= self.STABLE_NUMEL else None
                        for sp in (self._out_spec or [])]
                # fp32-storage modules: the operand images (bf16 planes) this function rebuilds, by their version stamps
                before = {id(m): m.pack_versions() for m in self.modules if getattr(m, "precision", None) == "fp32"}
                g = torch.cuda.CUDAGraph()
                # thread_local: other threads (RCCL's watchdog polls events) may touch the HIP runtime while
                # this thread captures; the default global mode turns that into a capture error / hang
                with torch.cuda.graph(g, pool=_pool(), capture_error_mode="thread_local"):
                    out = self._run()
                    self.static_out = self._stable(out, bufs) if len(bufs) == len(self._flat(out)) else out
                self.graph = g
                self._rebuilt = {mid: [cw for cw, pv in snap.items() if cw.packs_version != pv] for mid, snap in before.items()}
                if os.environ.get("RNAGAN_GRAPH_DEBUG"):
                    print("rna_gan_amd: captured graph #%d for %s" % (len(_captured), getattr(self.fn, "__name__", "?")),
                          flush=True)
                _captured.append(self)
            except Exception as e:  # capture unsupported (e.g. a collective that cannot be captured)
                self.failed = True
                print("rna_gan_amd: HIP graph capture failed (%s); continuing without graphs" % str(e)[:200])
                torch.cuda.synchronize()
                return self._run()
        self.graph.replay()
        for t in (self.static_out if isinstance(self.static_out, (tuple, list)) else (self.static_out,)):
            if torch.is_tensor(t):
                t._rg_static = True                # a captured graph's output IS a persistent buffer (see mark_static)
        for o in self.optimizers:
            o.note_replayed()
        for m in self.modules:
            m.mark_packs_fresh(only=self._rebuilt.get(id(m)))
        for m in self.stepped:
            m.weights_changed(by_optimizer=True)      # the replayed fused Adam refreshed the bf16 shadows too
        return self.static_out
