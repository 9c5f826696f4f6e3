"""DCGAN generator / discriminator with the torchgan constructor signatures and state_dict keys
(reference: src/histopathology_gan.py:175-192 instantiates torchgan.models.DCGANGenerator /
DCGANDiscriminator; the generator recipe is mirrored by src/dcgan.py:27-44,52,57-75,82; the
discriminator recipe is the third-party one recalled in SURVEY.md Appendix A).

The nn.Conv2d / nn.ConvTranspose2d / nn.BatchNorm2d children are PARAMETER CONTAINERS only: they
give the modules the exact key names, shapes and dtypes of the reference checkpoints.  ``forward``
never calls them; it runs the hand-written HIP kernels through rna_gan_amd.engine.  There is no
eager fallback: on a machine without the HIP library / a GPU, forward raises.
"""
from __future__ import annotations

from math import ceil, log2
from typing import List

import torch
import torch.nn as nn

from . import dist as D_
from . import engine as E


def _num_repeats(size: int, what: str) -> int:
    if size < 16 or ceil(log2(size)) != log2(size):
        raise Exception("%s must be at least 16*16 and an exact power of 2" % what)
    return size.bit_length() - 4


def _flat_view(buf, off, n, p):
    if E.is_tap_major(p.data):
        O, I = p.shape[0], p.shape[1]
        return buf[off:off + n].view(O, 4, 4, I).permute(0, 3, 1, 2)
    return buf[off:off + n].view(p.shape)


class FlatParams:
    """All parameters of a module re-homed into ONE flat fp32 buffer (and their gradients into
    another), so that the optimizer step is a single fused kernel and the data-parallel gradient
    all-reduce runs on a few large contiguous buckets.  Parameters stay ordinary nn.Parameters
    (views), so state_dict / load_state_dict / .parameters() behave as usual."""

    _GEN = [0]

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters()]
        dev = params[0].device
        # generation of this re-homing: captured HIP graphs hold the raw addresses of data / grad / shadow, so the
        # graph cache keys carry it (losses._Runner) and a rebuilt FlatParams never replays an older graph
        FlatParams._GEN[0] += 1
        self.gen = FlatParams._GEN[0]
        self.numel = sum(p.numel() for p in params)
        pad = (-self.numel) % 4
        dt = params[0].dtype          # fp32 in the product; tests of the DP glue run the engine in fp64
        self.data = torch.zeros(self.numel + pad, dtype=dt, device=dev)
        self.grad = torch.zeros(self.numel + pad, dtype=dt, device=dev)
        self.esize = self.data.element_size()
        self.shadow = None            # optional bf16 image of `data` kept current by the fused Adam (ensure_shadow)
        self.offsets = []
        off = 0
        for p in params:
            n = p.numel()
            # storage order is kept: a tap-major conv weight stays tap-major inside the flat buffer
            dst = _flat_view(self.data, off, n, p)
            gview = _flat_view(self.grad, off, n, p)
            dst.copy_(p.data)
            p.data = dst
            p.grad = gview
            self.offsets.append((off, n))
            off += n
        self.params = params

    def ensure_shadow(self, dtype=torch.bfloat16):
        """dtype: the 16-bit type of the library build that writes it (bf16, or fp16 for precision "fp16")."""
        if self.shadow is None or self.shadow.dtype != dtype:
            self.shadow = torch.zeros(self.data.numel(), dtype=dtype, device=self.data.device)
        return self.shadow

    def owns(self, module: nn.Module) -> bool:
        """True if the module's parameters still live in this flat buffer.  Gradient views that were
        dropped (zero_grad(set_to_none=True)) or replaced are re-attached."""
        ps = list(module.parameters())
        if len(ps) != len(self.params):
            return False
        base = self.data.data_ptr()
        for p, (off, n) in zip(ps, self.offsets):
            if p.data_ptr() != base + self.esize * off:
                return False
        gbase = self.grad.data_ptr()
        for p, (off, n) in zip(ps, self.offsets):
            if p.grad is None or p.grad.data_ptr() != gbase + self.esize * off:
                p.grad = _flat_view(self.grad, off, n, p)
        return True


class _HipModule(nn.Module):
    """Shared runtime plumbing: lazily binds the module to a HipOps backend + engine view."""

    def __init__(self):
        super().__init__()
        self._rt_ops = None
        self._rt_net = None
        self._rt_flat = None
        # "bf16" (MFMA kernels), "fp32" (fp32 storage, parity mode) or "fp16" (the MFMA kernels built for IEEE fp16 storage,
        # loss-scaled backward: BASELINE configs[3])
        self.precision = "bf16"

    def set_precision(self, precision: str):
        if precision not in ("bf16", "fp32", "fp16"):
            raise ValueError("precision must be 'bf16', 'fp32' or 'fp16'")
        if precision != self.precision:
            self.precision = precision
            self._rt_ops = None
            self._rt_net = None
        return self

    def _build_net(self):
        raise NotImplementedError

    def runtime(self):
        """(ops, net) for the module's current device; (re)built when parameters were re-homed."""
        p0 = next(self.parameters())
        if p0.device.type != "cuda":
            raise RuntimeError("%s runs on the HIP kernels only: move it to a ROCm GPU (.to('cuda')); "
                               "there is no CPU fallback" % type(self).__name__)
        if self._rt_flat is None or not self._rt_flat.owns(self):
            E.tap_major_(self)
            self._rt_flat = FlatParams(self)
            self._rt_net = None
        if self._rt_ops is None or self._rt_net is None:
            from .ops_hip import HipOps
            dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[self.precision]
            self._rt_ops = D_.attach_sync(HipOps(dt, p0.device))
            self._rt_net = self._build_net()
            for cw in self._rt_net.convs():
                cw.owner = "D" if hasattr(self, "disc") else "G"
            if self.precision in ("bf16", "fp16") and p0.dtype == torch.float32:
                self._attach_shadows()
        return self._rt_ops, self._rt_net

    def _attach_shadows(self):
        """bf16 precision: the tap-major conv weights' GEMM operand (wdn) is a slice of the flat bf16 shadow that
        the fused Adam writes together with the fp32 masters."""
        flat = self._rt_flat
        shadow = flat.ensure_shadow(self._rt_ops.h16)
        base = flat.data.data_ptr()
        g0 = getattr(self._rt_net, "g0", None)
        for cw in self._rt_net.convs():
            if cw.layout != "OHWI" and cw is not g0:
                continue                      # only layers with a bf16 GEMM-operand image built from the master
            off = (cw.w.data_ptr() - base) // flat.esize
            n = cw.w.numel()
            cw.shadow = shadow[off:off + n].view(cw.O, 16, cw.I) if cw.layout == "OHWI" else shadow[off:off + n]
            cw.shadow_version = -1

    @property
    def flat(self) -> "FlatParams":
        self.runtime()
        return self._rt_flat

    def weights_changed(self, by_optimizer=False):
        """Call after the parameters were modified in place (optimizer step, clamp, load).  by_optimizer: the
        fused Adam did it, which also refreshed the bf16 shadow in the same launch."""
        if self._rt_net is not None:
            self._rt_net.bump()
            if by_optimizer and self._rt_flat is not None and self._rt_flat.shadow is not None:
                for cw in self._rt_net.convs():
                    if cw.shadow is not None:
                        cw.shadow_version = cw.version

    def packs_stale(self) -> int:
        """0: every packed (bf16 GEMM-operand) weight image is current; 1: some are older than their master but
        the bf16 shadows are current (only the transposed images need rebuilding); 2: shadows are stale too.
        The launch sequence of a step depends on this, so it is part of the graph cache key."""
        if self._rt_net is None:
            return 2
        st = 0
        for cw in self._rt_net.convs():
            if cw.packs is None and self.precision == "fp32":
                continue          # fp32 storage: only the layers that run on bf16 planes (ops_hip._plane_packs) keep operand images
            if cw.packs_version != cw.version:
                st = max(st, 2 if (cw.shadow is not None and cw.shadow_version != cw.version) else 1)
        return st

    def mark_packs_fresh(self, only=None):
        """A captured graph that rebuilt the packs was replayed: bring the Python-side counters in line.
        only (fp32 storage): the layers whose operand images THAT graph rebuilds (StepGraph records them at capture).  With
        16-bit storage every train_op uses -- and rebuilds -- every layer's images; with fp32 storage a layer runs on bf16
        planes only at the shapes that have that kernel, e.g. the double batch of the D-loss step but not the single batch of the
        other two train_ops, so a replayed graph says nothing about the layers it did not rebuild."""
        if self._rt_net is not None:
            for cw in (self._rt_net.convs() if only is None else only):
                cw.packs_version = cw.version
                if cw.shadow is not None:
                    cw.shadow_version = cw.version

    def pack_versions(self):
        """{conv handle: packs_version} (StepGraph: which images a captured function rebuilt)."""
        return {} if self._rt_net is None else {cw: cw.packs_version for cw in self._rt_net.convs()}

    def load_state_dict(self, *a, **k):
        D_.flush()
        r = super().load_state_dict(*a, **k)
        self.weights_changed()
        return r

    def state_dict(self, *a, **k):
        D_.flush()          # a data-parallel train_op may have left this module's optimizer step in flight
        return super().state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        # .to(same device) / .cuda() on a module that already lives there (gan_utils.generate_images does it on every
        # call) leaves every parameter in place: keep the flat buffers, the engine view and with them every captured
        # graph.  Only a real move / cast re-homes.
        flat = self._rt_flat
        if flat is not None and next(self.parameters()).device == flat.data.device and \
                next(self.parameters()).dtype == flat.data.dtype and flat.owns(self):
            return r
        self._rt_ops = None
        self._rt_net = None
        self._rt_flat = None
        return r


class Generator(_HipModule):
    """torchgan.models.Generator base (attributes used at src/wgan_loss.py:91,100; src/gan_utils.py:226)."""

    def __init__(self, encoding_dims, label_type="none"):
        super().__init__()
        self.encoding_dims = encoding_dims
        self.label_type = label_type

    def _weight_initializer(self):
        for m in self.modules():
            if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def sampler(self, sample_size, device):
        return [torch.randn(sample_size, self.encoding_dims, device=device)]


class Discriminator(_HipModule):
    """torchgan.models.Discriminator base."""

    def __init__(self, input_dims, label_type="none"):
        super().__init__()
        self.input_dims = input_dims
        self.label_type = label_type

    _weight_initializer = Generator._weight_initializer


class _GenForwardFn(torch.autograd.Function):
    """``generator(noise)`` under torch autograd (custom losses that call loss.backward() themselves): forward = the HIP
    engine's train-mode forward, backward = the engine's backward, which ADDS the parameter gradients straight into the
    flat gradient buffer the parameters' .grad views live in (first order only; the gradient penalty's second-order
    pass has its own path in the loss plugins)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        ops, net = module.runtime()
        img, gctx = E._gen_fwd(ops, net, x, update_running=True)
        ctx.module, ctx.gctx = module, gctx
        return img

    @staticmethod
    def backward(ctx, gimg):
        ops, net = ctx.module.runtime()
        gin = E._gen_bwd(ops, net, ctx.gctx, gimg.contiguous().float(), accumulate=True,
                         need_input_grad=ctx.needs_input_grad[1])
        return (None, gin) + (None,) * len(ctx.module._rt_flat.params)


class _DiscForwardFn(torch.autograd.Function):
    """``discriminator(x)`` under torch autograd: per-sample cotangents in, parameter gradients added into the flat
    gradient buffer, d/dx returned (so discriminator(generator(z)) chains)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        ops, net = module.runtime()
        out, dctx = E.disc_forward(ops, net, x, update_running=True)
        ctx.module, ctx.dctx, ctx.need_x = module, dctx, x.requires_grad
        # a consumer that returns no gradient for this output (losses._GradientPenaltyFn: the penalty's parameter gradients
        # come from its own second-order pass) must not trigger a whole discriminator backward on a zero cotangent
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return (None, None) + (None,) * len(ctx.module._rt_flat.params)
        ops, net = ctx.module.runtime()
        wgrad = any(p.requires_grad for p in ctx.module._rt_flat.params)
        gx = E.disc_backward(ops, net, ctx.dctx, gout.contiguous(), wgrad=wgrad, accumulate=True,
                             need_input_grad=ctx.need_x)
        return (None, gx) + (None,) * len(ctx.module._rt_flat.params)


def _wants_autograd(module, x):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


class DCGANGenerator(Generator):
    """ConvT(E,d,4,1,0)+BN+nl ; R x [ConvT(d,d/2,4,2,1)+BN+nl] ; ConvT(d,ch,4,2,1,bias)+last_nl."""

    def __init__(self, encoding_dims=100, out_size=32, out_channels=3, step_channels=64, batchnorm=True,
                 nonlinearity=None, last_nonlinearity=None, label_type="none"):
        super().__init__(encoding_dims, label_type)
        if not batchnorm:
            raise NotImplementedError("rna_gan_amd implements the batchnorm=True recipe used by RNA-GAN")
        reps = _num_repeats(out_size, "Target Image Size")
        self.ch = out_channels
        self.n = step_channels
        nl = nn.LeakyReLU(0.2) if nonlinearity is None else nonlinearity
        last_nl = nn.Tanh() if last_nonlinearity is None else last_nonlinearity
        d = int(self.n * (2 ** reps))
        model: List[nn.Module] = [nn.Sequential(nn.ConvTranspose2d(self.encoding_dims, d, 4, 1, 0, bias=False),
                                                nn.BatchNorm2d(d), nl)]
        for _ in range(reps):
            model.append(nn.Sequential(nn.ConvTranspose2d(d, d // 2, 4, 2, 1, bias=False), nn.BatchNorm2d(d // 2), nl))
            d = d // 2
        model.append(nn.Sequential(nn.ConvTranspose2d(d, self.ch, 4, 2, 1, bias=True), last_nl))
        self.model = nn.Sequential(*model)
        self._weight_initializer()

    def _build_net(self):
        return E.build_gen_net(self)

    def forward(self, x, feature_matching=False):
        """Generated images (N, ch, S, S) fp32.  Train mode: batch statistics + running-stat update (what the reference's
        generator(noise) calls do); with gradients enabled the call is recorded for torch autograd (first order:
        _GenForwardFn), under torch.no_grad() it is a plain forward."""
        D_.flush()
        ops, net = self.runtime()
        x = x.view(-1, x.size(1)).contiguous().float()
        if self.training and _wants_autograd(self, x):
            return _GenForwardFn.apply(self, x, *self._rt_flat.params)
        if self.training:
            img, _ = E.gen_forward(ops, net, x, update_running=True, keep=False)
        elif getattr(self, "inference_fp8", False) and self.precision == "bf16":
            img, _ = E.gen_forward_eval_fp8(ops, net, x)
        else:
            img = E.gen_forward_eval(ops, net, x)
        return img

    def set_inference_fp8(self, on: bool = True):
        """Eval-mode forward with fp8 (e4m3) weights / activations where the layer shapes allow (BASELINE configs[4]);
        training and train-mode forwards are unaffected."""
        self.inference_fp8 = bool(on)
        return self


class DCGANUpGenerator(Generator):
    """Resize-convolution generator of src/dcgan.py:8-99 (same constructor): ConvT(E,d,4,1,0)+BN+nl ;
    R x [Upsample(x2, bilinear)+ReflectionPad2d(1)+Conv2d(d,d/2,3)+BN+nl] ; Upsample+ReflectionPad2d(1)+Conv2d(d,ch,3)
    with NO final activation (``last_nonlinearity`` is accepted and unused, as in the reference, :32,:76-84).
    state_dict keys as the reference's: model.0.{0,1}.*, model.{i}.{2,3}.*, model.{R+1}.2.*."""

    def __init__(self, encoding_dims=100, out_size=32, out_channels=3, step_channels=64, batchnorm=True,
                 nonlinearity=None, last_nonlinearity=None, label_type="none"):
        super().__init__(encoding_dims, label_type)
        if not batchnorm:
            raise NotImplementedError("rna_gan_amd implements the batchnorm=True (resize-convolution) recipe")
        reps = _num_repeats(out_size, "Target Image Size")
        self.ch = out_channels
        self.n = step_channels
        nl = nn.LeakyReLU(0.2) if nonlinearity is None else nonlinearity
        d = int(self.n * (2 ** reps))
        model: List[nn.Module] = [nn.Sequential(nn.ConvTranspose2d(self.encoding_dims, d, 4, 1, 0, bias=False),
                                                nn.BatchNorm2d(d), nl)]
        for _ in range(reps):
            model.append(nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"), nn.ReflectionPad2d(1),
                                       nn.Conv2d(d, d // 2, kernel_size=3, stride=1, padding=0),
                                       nn.BatchNorm2d(d // 2), nl))
            d = d // 2
        model.append(nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"), nn.ReflectionPad2d(1),
                                   nn.Conv2d(d, self.ch, kernel_size=3, stride=1, padding=0)))
        self.model = nn.Sequential(*model)
        self._weight_initializer()

    def _build_net(self):
        return E.build_upgen_net(self)

    def forward(self, x, feature_matching=False):
        D_.flush()
        ops, net = self.runtime()
        x = x.view(-1, x.size(1)).contiguous().float()
        if self.training and _wants_autograd(self, x):
            return _GenForwardFn.apply(self, x, *self._rt_flat.params)
        if self.training:
            img, _ = E.upgen_forward(ops, net, x, update_running=True, keep=False)
        else:
            img = E.upgen_forward_eval(ops, net, x)
        return img


class DCGANDiscriminator(Discriminator):
    """Conv(c,d,4,2,1,bias)+nl ; R x [Conv(d,2d,4,2,1)+BN+nl] ; disc = Conv(d,1,4,1,0)+last_nl -> (N,)."""

    def __init__(self, in_size=32, in_channels=3, step_channels=64, batchnorm=True, nonlinearity=None,
                 last_nonlinearity=None, label_type="none"):
        super().__init__(in_channels, label_type)
        if not batchnorm:
            raise NotImplementedError("rna_gan_amd implements the batchnorm=True recipe used by RNA-GAN")
        reps = _num_repeats(in_size, "Input Image Size")
        self.n = step_channels
        nl = nn.LeakyReLU(0.2) if nonlinearity is None else nonlinearity
        last_nl = nn.LeakyReLU(0.2) if last_nonlinearity is None else last_nonlinearity
        d = self.n
        model: List[nn.Module] = [nn.Sequential(nn.Conv2d(self.input_dims, d, 4, 2, 1, bias=True), nl)]
        for _ in range(reps):
            model.append(nn.Sequential(nn.Conv2d(d, d * 2, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 2), nl))
            d *= 2
        self.model = nn.Sequential(*model)
        self.disc = nn.Sequential(nn.Conv2d(d, 1, 4, 1, 0, bias=False), last_nl)
        self._weight_initializer()

    def _build_net(self):
        return E.build_disc_net(self)

    def forward(self, x, feature_matching=False):
        D_.flush()
        ops, net = self.runtime()
        if feature_matching:
            # torchgan: the activation in front of the ``disc`` head, (N, C, 4, 4).  Train mode = batch statistics (and
            # running-statistics update) like every other forward; eval mode = running statistics.
            if self.training:
                _, ctx = E.disc_forward(ops, net, x.contiguous().float(), update_running=True)
                a = ctx.a[-1]
            else:
                a = E.disc_features_eval(ops, net, x.contiguous().float())
            return a.float().permute(0, 3, 1, 2).contiguous()
        x = x.contiguous().float()
        if not self.training:
            # eval mode (running statistics): never used by the reference, provided for drop-in completeness; forward only
            return E.disc_forward_eval(ops, net, x)
        if _wants_autograd(self, x):
            return _DiscForwardFn.apply(self, x, *self._rt_flat.params)
        out, _ = E.disc_forward(ops, net, x, update_running=True)
        return out
