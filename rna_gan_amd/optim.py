"""Fused Adam on the flat parameter buffers (one HIP kernel per optimizer step).

Drop-in for ``torch.optim.Adam(params, lr, betas)`` as configured at
src/histopathology_gan.py:252,257 (stepped at src/wgan_loss.py:127,261,388): same update rule,
same ``state_dict()`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so optimizer
state in reference checkpoints loads and vice versa.

The step counter and the bias-correction constants live in DEVICE memory (rg_adam_hyper_dev +
rg_adam_step_dev), so an optimizer step contains no host-computed kernel argument and can be
replayed from a captured HIP graph (rna_gan_amd.graphed).
"""
from __future__ import annotations

import torch

from .models import _flat_view

from . import _abi
from ._abi import check


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if amsgrad:
            raise NotImplementedError("rna_gan_amd.optim.Adam: amsgrad is not on the RNA-GAN path")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                         foreach=False, fused=False)
        self._module = None
        self._m = None
        self._v = None
        self._flat_id = None
        self._step_dev = None      # int32[1] on the device
        self._hyper = None         # float32[12] on the device (9 used: rg_adam_hyper_dev2)
        self._host_steps = 0       # number of steps enqueued/replayed so far (mirror of *_step_dev)
        self.buf_gen = 0           # bumped whenever the moment / step buffers are re-allocated (part of the graph keys)
        self.grad_wire = None      # data parallel, bf16 wire: the all-reduced bf16 gradient buffer to step from

    def bind(self, module, fuse_linear_wgrad=False):
        """Tell the optimizer which HIP module owns its parameters (done by the Trainer).  The fused step updates the
        module's WHOLE flat buffer with param_groups[0]'s hyper-parameters, so the optimizer must hold exactly one
        group with every (trainable) parameter of the module.
        fuse_linear_wgrad (betaVAE training, bf16 kernels): the module's backward leaves the operands of every nn.Linear
        weight gradient (batch-only contraction) behind instead of forming it, and step() forms it inside the Adam pass of
        that weight (rg_linear_wgrad_adam).  Those weights' .grad is then NOT written: for loops of the reference's shape
        loss.backward(); optimizer.step() (src/betaVAE.py:225-226) with nothing reading .grad in between."""
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        theirs = list(module.parameters())
        if len(self.param_groups) != 1 or mine != {id(p) for p in theirs}:
            raise ValueError("rna_gan_amd.optim.Adam: one param group holding exactly module.parameters() is required "
                             "(the fused kernel steps the module's whole flat buffer)")
        if not all(p.requires_grad for p in theirs):
            raise ValueError("rna_gan_amd.optim.Adam: frozen parameters (requires_grad=False) are not supported")
        self._module = module
        if fuse_linear_wgrad:
            module._fuse_linear_wgrad = True
            if hasattr(module, "_trt"):
                module._trt = None                 # the training runtime reads the flag when it is built
        return self

    def _ensure(self):
        if self._module is None:
            raise RuntimeError("rna_gan_amd.optim.Adam must be bound to its module (Adam(...).bind(module))")
        flat = self._module.flat
        if self._flat_id is not flat:
            old = {p: self.state.get(p) for p in flat.params}
            self._m = torch.zeros_like(flat.data)
            self._v = torch.zeros_like(flat.data)
            step0 = 0
            for p, (off, n) in zip(flat.params, flat.offsets):
                st = old.get(p) or {}
                # same storage order as the parameter (tap-major conv weights stay strided views)
                m = _flat_view(self._m, off, n, p)
                v = _flat_view(self._v, off, n, p)
                if "exp_avg" in st:
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                step0 = max(step0, int(float(st.get("step", 0))))
                self.state[p] = {"step": torch.tensor(float(step0)), "exp_avg": m, "exp_avg_sq": v}
            self._host_steps = step0
            self._step_dev = torch.tensor([step0], dtype=torch.int32, device=flat.data.device)
            self._hyper = torch.zeros(12, dtype=torch.float32, device=flat.data.device)
            self._flat_id = flat
            self.buf_gen += 1
        return flat

    def _sync_step_state(self):
        for p in (self._flat_id.params if self._flat_id is not None else []):
            self.state[p]["step"] = torch.tensor(float(self._host_steps))

    def state_dict(self):
        from . import dist as D_
        D_.flush()                 # a data-parallel train_op may have left this optimizer's step in flight
        self._sync_step_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        from . import dist as D_
        D_.flush()
        super().load_state_dict(state_dict)
        flat = self._flat_id
        if flat is None or self._module is None or self._module._rt_flat is not flat:
            self._flat_id = None  # not homed yet (or the module was re-homed): done at next use by _ensure()
            return
        # already homed (Trainer.load_model after training has started): copy the loaded moments INTO the existing flat
        # buffers -- captured graphs hold their addresses -- and point the per-parameter state back at the views
        step0 = 0
        with torch.no_grad():
            for p, (off, n) in zip(flat.params, flat.offsets):
                st = self.state.get(p) or {}
                m, v = _flat_view(self._m, off, n, p), _flat_view(self._v, off, n, p)
                if "exp_avg" in st:
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                else:
                    m.zero_(); v.zero_()
                step0 = max(step0, int(float(st.get("step", 0))))
                self.state[p] = {"step": torch.tensor(float(step0)), "exp_avg": m, "exp_avg_sq": v}
            self._step_dev.fill_(step0)
        self._host_steps = step0

    def zero_grad(self, set_to_none: bool = False):
        # gradients are written (not accumulated) by the first backward of every step; the views are kept.
        # set_to_none=True (train_betaVAE's call, src/betaVAE.py:221) asks for "no stale gradient", which the
        # overwrite semantics already give: no 4 B/parameter memset
        if self._module is not None and not set_to_none:
            from . import dist as D_
            D_.flush()             # a pending data-parallel all-reduce still reads (fp32 wire: writes) this buffer
            self._module.flat.grad.zero_()

    def note_replayed(self):
        """A captured graph containing one step of this optimizer was replayed."""
        self._host_steps += 1

    @torch.no_grad()
    def step(self, closure=None):
        flat = self._ensure()
        g = self.param_groups[0]
        # the module's backend decides the build of the library (bf16 / fp16 storage: the shadow image, the slabs and the wire
        # are in ITS 16-bit type) and the loss scale its backward passes put on every gradient of this step
        mops = getattr(self._module, "_rt_ops", None)
        lib = mops.lib if mops is not None else _abi.load()
        ginv = 1.0 / float(getattr(mops, "loss_scale", 1.0)) if mops is not None else 1.0
        stream = torch.cuda.current_stream(flat.data.device).cuda_stream
        check(lib.rg_adam_hyper_dev2(self._step_dev.data_ptr(), float(g["lr"]), float(g["betas"][0]),
                                     float(g["betas"][1]), float(g["eps"]), float(g.get("weight_decay", 0.0)), ginv,
                                     self._hyper.data_ptr(), stream),
              "rg_adam_hyper_dev2")
        shadow = flat.shadow           # bf16 image of the parameters (bf16 precision only), written by the same launch
        lo = 0
        g0 = getattr(getattr(self._module, "_rt_net", None), "g0", None)
        pend = None if g0 is None else g0.pending_wgrad
        if pend is not None:
            # generator layer 0 (60 % of the generator's parameters): its weight gradient was NOT written -- form it and apply
            # the step in one streaming kernel (26 B per parameter instead of 4 + 30), then step the rest of the buffer
            g0.pending_wgrad = None
            z, gy, dt = pend
            off = (g0.w.data_ptr() - flat.data.data_ptr()) // 4
            n0 = g0.w.numel()
            if off != 0:
                raise RuntimeError("rna_gan_amd.optim.Adam: the deferred G.0 weight gradient expects that tensor at the head "
                                   "of the flat buffer")
            # (data parallel: z / gy are the factors gathered from all ranks, K = world x batch; the rest of the buffer steps
            # from the all-reduced wire below)
            E, C = g0.w.shape[0], g0.w.shape[1]
            check(lib.rg_g0_wgrad_adam(z.data_ptr(), gy.data_ptr(), flat.data.data_ptr(), self._m.data_ptr(),
                                       self._v.data_ptr(), self._hyper.data_ptr(), 0 if shadow is None else shadow.data_ptr(),
                                       z.shape[0], E, C, dt, stream), "rg_g0_wgrad_adam")
            lo = n0
        # nn.Linear weights whose gradient operands the backward left behind (bind(fuse_linear_wgrad=True)): gradient + Adam
        # in one pass per weight; everything between those segments (biases, BatchNorm parameters) steps from .grad as usual
        segs = []
        pend_lin = getattr(self._module, "_rg_pending_linear", None)
        if pend_lin:
            self._module._rg_pending_linear = []
            if shadow is not None or self.grad_wire is not None:
                raise RuntimeError("rna_gan_amd.optim.Adam: fused linear weight gradients expect an fp32-only, single-process step")
            for w, gT, xT, nsamp, pack in pend_lin:
                off = (w.data_ptr() - flat.data.data_ptr()) // 4
                O_, I_ = w.shape
                if not w.is_contiguous() or off < 0 or off + O_ * I_ > flat.data.numel():
                    raise RuntimeError("rna_gan_amd.optim.Adam: a pending linear weight is not a dense view of the flat buffer")
                # pack: {"image": bf16 [Np][Kp] zero-initialised, "version": ...} -- the runtime's operand image of this weight; the
                # kernel refreshes it from the updated values, and it is marked current for the tensor version seen here
                img = None if pack is None else pack["image"]
                check(lib.rg_linear_wgrad_adam(gT.data_ptr(), xT.data_ptr(), gT.shape[1], nsamp, flat.data.data_ptr() + 4 * off,
                                               self._m.data_ptr() + 4 * off, self._v.data_ptr() + 4 * off,
                                               self._hyper.data_ptr(), O_, I_, 0 if img is None else img.data_ptr(),
                                               0 if img is None else img.shape[1], stream), "rg_linear_wgrad_adam")
                if pack is not None:
                    pack["version"] = w._version
                segs.append((off, off + O_ * I_))
            segs.sort()
        # 4 x 4 conv layers whose split-K weight-gradient slabs the backward left unreduced (ConvW.pending_slabs, set by
        # ops_hip._wgrad_slabs for a pass the train_op runner marked with defer_slabs): ONE launch steps the rest of the buffer,
        # summing those layers' slabs in place of their (never written) reduced gradient
        slab_segs = []
        net = getattr(self._module, "_rt_net", None)
        for cw in (net.convs() if net is not None and hasattr(net, "convs") else []):
            ps = getattr(cw, "pending_slabs", None)
            if ps is not None:
                cw.pending_slabs = None
                slab_segs.append(((cw.w.data_ptr() - flat.data.data_ptr()) // 4, cw.w.numel(), ps[0], ps[1], ps[2]))
            pb = getattr(cw, "pending_bias", None)
            if pb is not None:
                # the image-side layer's bias-gradient partials of the same pass ([count][64] fp32): one more slab segment
                cw.pending_bias = None
                slab_segs.append(((cw.bias.data_ptr() - flat.data.data_ptr()) // 4, cw.bias.numel(), pb[0], pb[1], 0))
            pw = getattr(cw, "pending_wgrad", None)
            if isinstance(pw, tuple) and isinstance(pw[0], str) and pw[0] == "conv":
                # a layer whose weight-gradient plan has no split-K (ops_hip._wgrad_slabs left the operands): gradient tile and
                # Adam step in ONE launch; the streaming launch below skips the tensor (segment with nsplit = -1)
                cw.pending_wgrad = None
                _, low0, high0, low1, high1, (N_, Ho_, Wo_, O_, I_), dt, algo, flops = pw
                off = (cw.w.data_ptr() - flat.data.data_ptr()) // 4
                if shadow is None or off % 4 or off < lo:
                    raise RuntimeError("rna_gan_amd.optim.Adam: a deferred single-launch weight gradient expects a bf16 step and "
                                       "a 16-byte aligned tensor inside the flat buffer")
                ops = getattr(self._module, "_rt_ops", None)
                call = lambda: check(lib.rg_conv_wgrad_adam(
                    low0.data_ptr(), high0.data_ptr(), 0 if low1 is None else low1.data_ptr(),
                    0 if high1 is None else high1.data_ptr(), flat.data.data_ptr() + 4 * off, self._m.data_ptr() + 4 * off,
                    self._v.data_ptr() + 4 * off, self._hyper.data_ptr(), shadow.data_ptr() + 2 * off, N_, Ho_, Wo_, O_, I_, dt,
                    algo, stream), "rg_conv_wgrad_adam")
                if ops is not None and hasattr(ops, "_timed"):
                    # bench.py's per-family timing: a family of its own -- the launch's interval includes the Adam epilogue
                    ops._timed("conv_wgrad_adam", flops, call, cw=cw)
                else:
                    call()
                slab_segs.append((off, cw.w.numel(), None, -1, 0))
        if slab_segs:
            if segs or self.grad_wire is not None:
                raise RuntimeError("rna_gan_amd.optim.Adam: deferred split-K slabs expect a single-process step without fused "
                                   "linear weight gradients")
            import ctypes as C
            slab_segs.sort(key=lambda t: t[0])
            total = flat.data.numel()
            table, pos = [], lo
            for off, n, buf, ns, sdt in slab_segs:
                if off < pos or off % 4 or n % 4 or off + n > total:
                    raise RuntimeError("rna_gan_amd.optim.Adam: a deferred weight gradient does not sit 16-byte aligned inside "
                                       "the part of the flat buffer this launch steps")
                if off > pos:
                    table.append((pos - lo, off - pos, 0, 0, 0))
                table.append((off - lo, n, 0 if buf is None else buf.data_ptr(), ns, sdt))
                pos = off + n
            if total > pos:
                table.append((pos - lo, total - pos, 0, 0, 0))
            k = len(table)
            offs = (C.c_ulonglong * k)(*[t[0] for t in table])
            lens = (C.c_ulonglong * k)(*[t[1] for t in table])
            slabs = (C.c_void_p * k)(*[t[2] or None for t in table])
            nsp = (C.c_int * k)(*[t[3] for t in table])
            sdts = (C.c_int * k)(*[t[4] for t in table])
            check(lib.rg_adam_step_slabs(flat.data.data_ptr() + 4 * lo, flat.grad.data_ptr() + 4 * lo,
                                         self._m.data_ptr() + 4 * lo, self._v.data_ptr() + 4 * lo, total - lo,
                                         self._hyper.data_ptr(), 0 if shadow is None else shadow.data_ptr() + 2 * lo, k,
                                         C.addressof(offs), C.addressof(lens), C.addressof(slabs), C.addressof(nsp),
                                         C.addressof(sdts), stream),
                  "rg_adam_step_slabs")
            segs = [(lo, total)]                  # nothing left for the plain launches below
        pos = lo
        for a, b in segs + [(flat.data.numel(), flat.data.numel())]:
            if a > pos:
                check(lib.rg_adam_step_dev(flat.data.data_ptr() + 4 * pos, flat.grad.data_ptr() + 4 * pos,
                                           self._m.data_ptr() + 4 * pos, self._v.data_ptr() + 4 * pos, a - pos,
                                           self._hyper.data_ptr(), 0 if shadow is None else shadow.data_ptr() + 2 * pos,
                                           0 if self.grad_wire is None else self.grad_wire.data_ptr() + 2 * pos, stream),
                      "rg_adam_step_dev")
            pos = max(pos, b)
        if not torch.cuda.is_current_stream_capturing():
            self._host_steps += 1
        self._module.weights_changed(by_optimizer=True)
        return None
