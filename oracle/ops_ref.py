"""Torch (CPU) twins of every op the HIP library exports (TEST INFRASTRUCTURE ONLY).

``RefOps`` implements the same Python-level op interface as ``rna_gan_amd.ops_hip.HipOps``
with plain torch fp32 math.  Uses:
  * tests compare each HIP op against its twin on the same inputs;
  * tests run ``rna_gan_amd.engine`` (the explicit forward / backward / gradient-penalty
    sequencing) on top of RefOps on the CPU and compare it with the autograd oracle
    (oracle/ref_cpu.py) -- this validates the algorithm without a GPU.
It is never imported by the product path.

Conventions (identical to the HIP ops):
  * activations are NHWC tensors (N,H,W,C) of ``act_dtype`` (float32 or bfloat16); math is fp32,
    results are rounded to ``act_dtype`` on store;
  * image-side boundary tensors are NCHW float32;
  * a conv weight is ``w[O][I][4][4]`` fp32 where O = channels on the LOW-resolution side and
    I = channels on the HIGH-resolution side.  That is Conv2d's (out,in,kh,kw) for the
    discriminator and ConvTranspose2d's (in,out,kh,kw) for the generator, so ``conv_down`` /
    ``conv_up`` serve both networks and each other's backward;
  * per-channel statistics are fp32.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


from rna_gan_amd.engine import ConvW  # noqa: E402  (weight handle shared by both backends)


def _nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(dt)


def _lrelu_mask(v, slope):
    return torch.where(v > 0, torch.ones_like(v), torch.full_like(v, slope))


class RefOps:
    name = "ref"

    def __init__(self, act_dtype=torch.float32, device="cpu"):
        self.act_dtype = act_dtype
        self.device = torch.device(device)
        # math dtype: fp32 like the kernels; tests of the ALGORITHM use float64 to separate
        # algebra errors from rounding
        self.f = torch.float64 if act_dtype == torch.float64 else torch.float32
        # synchronised (global-batch) statistics in a data-parallel run: an in-place SUM all-reduce for small fp tensors
        # and the number of ranks (rna_gan_amd.dist.sync_stats); None = rank-local statistics
        self.stat_reduce = None
        self.stat_world = 1

    def _nchw(self, x):
        return x.to(self.f).permute(0, 3, 1, 2)

    from contextlib import contextmanager as _cm

    @_cm
    def side(self, *tensors):
        yield

    def join(self):
        pass

    def _wq(self, w):
        # the bf16 path computes with weights rounded to bf16
        return w.to(self.f) if self.act_dtype != torch.bfloat16 else w.to(self.act_dtype).to(self.f)

    # ------------------------------------------------------------------ conv family
    # (the handle's master may be stored tap-major; cw.oihw() / cw.store_grad_oihw() give the PyTorch view)
    def conv_down(self, x, cw: ConvW, want_stats=False, defer=0, bn_bwd=None):      # defer: a launch-fusion hint of the HIP backend
        y = _nhwc(F.conv2d(self._nchw(x), self._wq(cw.oihw()), None, stride=2, padding=1), self.act_dtype)
        return (y, None) if want_stats else y          # the twin has no fused statistics: bn_forward computes them

    def conv_up(self, x, cw: ConvW, mask_act=None, slope=1.0, want_stats=False, defer=0, bn_bwd=None):
        y = F.conv_transpose2d(self._nchw(x), self._wq(cw.oihw()), None, stride=2, padding=1)
        if mask_act is not None:       # fused LeakyReLU backward: applied to the fp32 result, rounded once
            y = y * _lrelu_mask(self._nchw(mask_act), slope)
        y = _nhwc(y, self.act_dtype)
        return (y, None) if want_stats else y

    def conv_wgrad(self, low, high, cw: ConvW, accumulate: bool):
        g = torch.nn.grad.conv2d_weight(self._nchw(high), (cw.O, cw.I, 4, 4), self._nchw(low), stride=2, padding=1)
        cw.store_grad_oihw(g, accumulate)

    def conv_wgrad2(self, low0, high0, low1, high1, cw: ConvW, accumulate: bool):
        self.conv_wgrad(low0, high0, cw, accumulate)
        self.conv_wgrad(low1, high1, cw, True)

    # ------------------------------------------------------------------ resize-conv block (DCGANUpGenerator)
    def _uppad(self, x_nhwc):
        up = F.interpolate(self._nchw(x_nhwc), scale_factor=2, mode="bilinear", align_corners=False)
        pad = F.pad(up, (1, 1, 1, 1), mode="reflect")
        # the interpolated image is a GEMM operand: rounded to the activation dtype on the bf16 path
        return pad if self.act_dtype != torch.bfloat16 else pad.to(torch.bfloat16).to(self.f)

    def upconv3(self, x, cw: ConvW, bias, out_nchw=False):
        y = F.conv2d(self._uppad(x), self._wq(cw.w), None if bias is None else bias.to(self.f))
        return y.contiguous() if out_nchw else _nhwc(y, self.act_dtype)

    def upconv3_bwd_data(self, gy, cw: ConvW, gy_nchw=False):
        g = gy.to(self.f) if gy_nchw else self._nchw(gy)
        gpad = F.conv_transpose2d(g, self._wq(cw.w))                      # gradient wrt the padded upsampled image
        N, Cin, Hp, Wp = gpad.shape
        x0 = torch.zeros(N, Cin, (Hp - 2) // 2, (Wp - 2) // 2, dtype=self.f, requires_grad=True)
        with torch.enable_grad():
            pad = F.pad(F.interpolate(x0, scale_factor=2, mode="bilinear", align_corners=False), (1, 1, 1, 1),
                        mode="reflect")
            gx, = torch.autograd.grad(pad, x0, gpad)                      # adjoint of a fixed linear map
        return _nhwc(gx, self.act_dtype)

    def upconv3_wgrad(self, gy, x, cw: ConvW, accumulate: bool, gy_nchw=False):
        g = gy.to(self.f) if gy_nchw else self._nchw(gy)
        d = torch.nn.grad.conv2d_weight(self._uppad(x), cw.w.shape, g)
        if accumulate:
            cw.dw.add_(d)
        else:
            cw.dw.copy_(d)

    def first_down_tangent(self, v_nchw, cw: ConvW, a0, slope: float):
        return self.lrelu_bwd(self.first_down(v_nchw, cw, None, 1.0), a0, slope)

    def sign_bits_for(self, N, H, W, C0=64, C1=128):
        return None

    def first_down(self, x_nchw, cw: ConvW, bias, slope: float, out=None):
        if out is not None:
            out[0].copy_(self.first_down(x_nchw, cw, bias, slope))
            return out[0]
        # bf16 path: the image-side layers run on the matrix cores too, so image and weights are rounded to
        # bf16 operands (fp32 accumulation); the fp32 path uses the masters as they are
        xin = x_nchw.to(self.f) if self.act_dtype != torch.bfloat16 else x_nchw.to(torch.bfloat16).to(self.f)
        y = F.conv2d(xin, self._wq(cw.w), bias, stride=2, padding=1)
        if slope != 1.0:
            y = F.leaky_relu(y, slope)
        return _nhwc(y, self.act_dtype)

    def last_up_bn(self, z, partials, bn, slope, cw, bias, tanh, update_running=True):
        return None

    def last_up_bn2(self, z, partials, bn, slope, cw, bias, tanh, update_running=True):
        return None

    def last_up(self, x, cw: ConvW, bias, tanh: bool):
        y = F.conv_transpose2d(self._nchw(x), self._wq(cw.w), bias, stride=2, padding=1)
        return torch.tanh(y) if tanh else y.contiguous()

    def last_up_post(self, x, cw: ConvW, tanh_img=None):
        """Twin of HipOps.last_up_post: one "workgroup" row of partial sums (channel sums, sum of squares)."""
        y = self.last_up(x, cw, None, False)
        if tanh_img is not None:
            y = y * (1 - tanh_img * tanh_img)
        parts = torch.cat([y.sum(dim=(0, 2, 3)), (y * y).sum().reshape(1)]).reshape(1, 4).to(self.f)
        return y.contiguous(), parts

    def parts_chan_sum(self, parts, out, accumulate: bool):
        s = parts[:, :3].sum(0)
        if accumulate:
            out.add_(s)
        else:
            out.copy_(s)

    def gp_coef_parts(self, parts, lambd: float):
        return self.gp_coef(parts[:, 3].sum().reshape(1), lambd)

    def skinny_wgrad(self, low, high_nchw, dw, accumulate: bool, dbias=None, dbias_accumulate=False):
        g = torch.nn.grad.conv2d_weight(high_nchw.to(self.f), dw.shape, self._nchw(low), stride=2, padding=1)
        if accumulate:
            dw.add_(g)
        else:
            dw.copy_(g)
        return False          # the bias gradient is not formed here: the engine calls col_sum

    # ------------------------------------------------------------------ G.0 / head
    def g0_fwd(self, z, cw: ConvW):
        # ConvTranspose2d(E->C, k4, s1, p0) on a 1x1 input: y[n,kh,kw,c] = sum_e z[n,e] w[e,c,kh,kw]
        y = torch.einsum("ne,ecij->nijc", z.to(self.f), self._wq(cw.w))
        return y.contiguous().to(self.act_dtype)

    def g0_wgrad_deferred(self, z, gy, cw, accumulate: bool):
        return False          # (HIP backend: the weight gradient of G.0 can be formed inside the fused optimizer step)

    def g0_wgrad(self, z, gy, dw, accumulate: bool):
        g = torch.einsum("ne,nijc->ecij", z.to(self.f), gy.to(self.f))
        if accumulate:
            dw.add_(g)
        else:
            dw.copy_(g)

    def g0_bwd_data(self, gz0, cw: ConvW):
        # d/dz of the first generator layer: gin[n,e] = sum gz0[n,kh,kw,c] w[e,c,kh,kw]  (fp32 functor GEMM in the product)
        return torch.einsum("nijc,ecij->ne", gz0.to(self.f), cw.w.to(self.f)).contiguous()

    def head_fwd(self, a, cw: ConvW, slope: float):
        # Conv2d(C->1, k4, s1, p0) on a 4x4 map: h[n] = sum a[n,kh,kw,c] w[0,c,kh,kw]
        h = torch.einsum("nijc,cij->n", a.to(self.f), self._wq(cw.w)[0])
        return h, F.leaky_relu(h, slope)

    def head_grad(self, h, coef: float, slope: float):
        return coef * _lrelu_mask(h, slope)

    def head_bwd_data(self, gh, cw: ConvW):
        ga = torch.einsum("n,cij->nijc", gh.to(self.f), self._wq(cw.w)[0])
        return ga.contiguous().to(self.act_dtype)

    def head_wgrad(self, gh, a, dw, accumulate: bool):
        g = torch.einsum("n,nijc->cij", gh.to(self.f), a.to(self.f)).unsqueeze(0)
        if accumulate:
            dw.add_(g)
        else:
            dw.copy_(g)

    # ------------------------------------------------------------------ batch norm (train mode)
    def bn_stats(self, z):
        zf = z.to(self.f).reshape(-1, z.shape[-1])
        return zf.sum(0), (zf * zf).sum(0)

    def _red(self, *ts):
        if self.stat_reduce is not None:
            for t in ts:
                self.stat_reduce(t)

    def stat_allreduce(self, t):
        """SUM over the ranks when statistics are synchronised (the penalty's squared norm), identity otherwise."""
        self._red(t)
        return t

    def bn_forward(self, z, gamma, beta, slope: float, eps: float, momentum: float, running_mean=None,
                   running_var=None, nbt=None, partials=None, out=None):
        if out is not None:
            a, mean, invstd = self.bn_forward(z, gamma, beta, slope, eps, momentum, running_mean, running_var, nbt, partials)
            out.copy_(a)
            return out, mean, invstd
        if self.stat_reduce is not None:
            s, ss = self.bn_stats(z)
            self._red(s, ss)
            count = (z.numel() // z.shape[-1]) * self.stat_world
            mean, invstd = self.bn_finalize(s, ss, count, eps, momentum, running_mean, running_var, nbt)
            return self.bn_act(z, mean, invstd, gamma, beta, slope), mean, invstd
        mean, invstd = self.bn_stats_finalize(z, eps, momentum, running_mean, running_var, nbt)
        return self.bn_act(z, mean, invstd, gamma, beta, slope), mean, invstd

    def bn_forward2(self, z, gamma, beta, slope: float, eps: float, momentum: float, running_mean=None,
                    running_var=None, nbt=None, partials=None, nblk=1):
        n = z.shape[0] // 2
        r = [self.bn_forward(z[h * n:(h + 1) * n], gamma, beta, slope, eps, momentum, running_mean, running_var, nbt)
             for h in range(2)]
        return torch.cat([r[0][0], r[1][0]]), torch.stack([r[0][1], r[1][1]]), torch.stack([r[0][2], r[1][2]])

    def bn_act_bwd2(self, z, ga, mean, invstd, gamma, beta, slope: float, dgamma=None, dbeta=None, accumulate: bool = False):
        n = z.shape[0] // 2
        r = [self.bn_act_bwd(z[h * n:(h + 1) * n], ga[h * n:(h + 1) * n], mean[h], invstd[h], gamma, beta, slope, dgamma, dbeta,
                             accumulate or h == 1)[0] for h in range(2)]
        return torch.cat(r)

    def bn_stats_finalize(self, z, eps: float, momentum: float, running_mean=None, running_var=None, nbt=None):
        s, ss = self.bn_stats(z)
        return self.bn_finalize(s, ss, z.numel() // z.shape[-1], eps, momentum, running_mean, running_var, nbt)

    def bn_finalize(self, s, ss, count: int, eps: float, momentum: float,
                    running_mean=None, running_var=None, nbt=None):
        mean = s / count
        var = torch.clamp(ss / count - mean * mean, min=0.0)
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            unb = var * (count / max(count - 1, 1))
            running_mean.mul_(1 - momentum).add_(momentum * mean)
            running_var.mul_(1 - momentum).add_(momentum * unb)
            nbt.add_(1)
        return mean, invstd

    def bn_act(self, z, mean, invstd, gamma, beta, slope: float):
        y = (z.to(self.f) - mean) * (invstd * gamma) + beta
        return F.leaky_relu(y, slope).to(self.act_dtype)

    def bn_act_bwd(self, z, ga, mean, invstd, gamma, beta, slope: float,
                   dgamma=None, dbeta=None, accumulate: bool = False, out=None, keep_ga=True):
        """Backward of a = lrelu(bn(z)).  Returns (gz, s_gy, s_gyxh) with gy = ga * lrelu'(y)."""
        if out is not None:
            gz, s_gy, s_gyxh = self.bn_act_bwd(z, ga, mean, invstd, gamma, beta, slope, dgamma, dbeta, accumulate)
            out.copy_(gz)
            return out, s_gy, s_gyxh
        xh = (z.to(self.f) - mean) * invstd
        y = xh * gamma + beta
        gy = ga.to(self.f) * _lrelu_mask(y, slope)
        C = z.shape[-1]
        m = (z.numel() // C) * self.stat_world
        s_gy = gy.reshape(-1, C).sum(0)
        s_gyxh = (gy * xh).reshape(-1, C).sum(0)
        if dgamma is not None:          # parameter gradients: this rank's contribution (summed by the grad all-reduce)
            if accumulate:
                dgamma.add_(s_gyxh); dbeta.add_(s_gy)
            else:
                dgamma.copy_(s_gyxh); dbeta.copy_(s_gy)
        self._red(s_gy, s_gyxh)         # the data gradient needs the batch means: global sums when synchronised
        gz = (gamma * invstd) * (gy - s_gy / m - xh * (s_gyxh / m))
        return gz.to(self.act_dtype), s_gy, s_gyxh

    def bn_tangent(self, z, zt, mean, invstd, gamma, beta, slope: float):
        """Forward-mode tangent of a = lrelu(bn(z)) in direction zt (batch statistics vary).
        Returns (at, s_zt, s_xhzt)."""
        xh = (z.to(self.f) - mean) * invstd
        y = xh * gamma + beta
        C = z.shape[-1]
        m = (z.numel() // C) * self.stat_world
        ztf = zt.to(self.f)
        s_zt = ztf.reshape(-1, C).sum(0)
        s_xhzt = (xh * ztf).reshape(-1, C).sum(0)
        self._red(s_zt, s_xhzt)
        yt = (gamma * invstd) * (ztf - s_zt / m - xh * (s_xhzt / m))
        return (yt * _lrelu_mask(y, slope)).to(self.act_dtype), s_zt, s_xhzt

    def bn_double_bwd(self, z, qa, zt, ga1, mean, invstd, gamma, beta, slope: float,
                      s_gy, s_gyxh, s_zt, s_xhzt, dgamma, dbeta, accumulate: bool):
        """Reverse of the (primal, tangent) pair through a = lrelu(bn(z)), at = d/de a(z + e zt).

        Cotangents: ``qa`` on a (None = zero), ``ga1`` on at (the first-backward gradient w.r.t. a;
        with gy = ga1*lrelu'(y) its sums s_gy, s_gyxh were produced by ``bn_act_bwd``); s_zt,
        s_xhzt come from ``bn_tangent``.  Returns pz = cotangent on z (the cotangent on zt is the
        first backward's gz and is not recomputed).  Accumulates dgamma, dbeta.  DESIGN.md sec. GP.
        """
        C = z.shape[-1]
        m_local = z.numel() // C
        m = m_local * self.stat_world           # s_gy, s_gyxh, s_zt, s_xhzt are global sums when synchronised
        xh = (z.to(self.f) - mean) * invstd
        y = xh * gamma + beta
        mask = _lrelu_mask(y, slope)
        gy = ga1.to(self.f) * mask
        ztf = zt.to(self.f)
        s_gyzt = (gy * ztf).reshape(-1, C).sum(0)
        self._red(s_gyzt)
        b = s_gyxh / m
        c = s_xhzt / m
        A = s_gyzt / m - (s_gy / m) * (s_zt / m)
        k2 = gamma * invstd * invstd
        pz = -k2 * (xh * (A - 3 * b * c) + c * (gy - s_gy / m) + b * (ztf - s_zt / m))
        dg = (m_local * invstd) * (A - b * c)       # this rank's share of m/sigma (A - bc)
        db = torch.zeros_like(dg)
        if qa is not None:
            qy = qa.to(self.f) * mask
            s_qy = qy.reshape(-1, C).sum(0)
            s_qyxh = (qy * xh).reshape(-1, C).sum(0)
            dg = dg + s_qyxh                        # local sums for the parameter gradients
            db = s_qy.clone()
            self._red(s_qy, s_qyxh)
            pz = pz + (gamma * invstd) * (qy - s_qy / m - xh * (s_qyxh / m))
        if accumulate:
            dgamma.add_(dg); dbeta.add_(db)
        else:
            dgamma.copy_(dg); dbeta.copy_(db)
        return pz.to(self.act_dtype)

    # ------------------------------------------------------------------ pointwise / reductions
    def lrelu_bwd(self, g, a, slope: float):
        """g * lrelu'(.) with the mask taken from the sign of the OUTPUT a (lrelu keeps sign)."""
        return (g.to(self.f) * _lrelu_mask(a.to(self.f), slope)).to(self.act_dtype)

    def col_sum(self, g, out, accumulate: bool):
        s = g.to(self.f).reshape(-1, g.shape[-1]).sum(0)
        if accumulate:
            out.add_(s)
        else:
            out.copy_(s)

    def tanh_bwd(self, gy_nchw, y_nchw):
        return gy_nchw * (1 - y_nchw * y_nchw)

    def nchw_chan_sum(self, g_nchw, out, accumulate: bool):
        s = g_nchw.sum(dim=(0, 2, 3))
        if accumulate:
            out.add_(s)
        else:
            out.copy_(s)

    def interp(self, real, fake, eps):
        return eps * real + (1 - eps) * fake

    def sqnorm(self, x):
        return (x.double() ** 2).sum().to(self.f).reshape(1)

    def gp_coef(self, sq, lambd: float):
        """From ||g||^2 (device scalar): loss = (||g||-1)^2, coef = lambd*2(||g||-1)/||g||."""
        nrm = torch.sqrt(sq)
        return (nrm - 1) ** 2, lambd * 2 * (nrm - 1) / nrm

    def scale_by(self, x, coef_dev):
        return x * coef_dev

    def fill_const(self, n: int, value: float):
        return torch.full((n,), value, dtype=self.f, device=self.device)

    def mean_diff(self, a, b=None, sign: float = 1.0):
        """sign * mean(a - b) (b optional) as a 1-element device tensor."""
        v = a.to(self.f) if b is None else a.to(self.f) - b.to(self.f)
        return (sign * v.mean()).reshape(1)

    def latent_prep(self, u, z):
        n = u + z
        if self.stat_reduce is not None:
            s, ss = n.sum(0), (n * n).sum(0)
            self._red(s, ss)
            nt = n.shape[0] * self.stat_world
            mu = s / nt
            return (n - mu) / torch.sqrt((ss - nt * mu * mu) / (nt - 1))
        return (n - n.mean(0)) / n.std(0)

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, p, g, m, v, step: int, lr: float, b1: float, b2: float, eps: float):
        """torch.optim.Adam (no amsgrad, no weight decay) on flat fp32 buffers; ``step`` is 1-based."""
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)

    def clamp_(self, p, lo: float, hi: float):
        p.clamp_(lo, hi)

    # ------------------------------------------------------------------ dense (betaVAE encoder)
    def linear_affine_act(self, x, w, scale, shift, slope: float, wp=None):
        """act((x @ w.T) * scale + shift); Linear+BatchNorm1d(eval) folded, slope=1 -> no act."""
        y = (x.to(self.f) @ self._wq(w).t()) * scale + shift
        if slope != 1.0:
            y = F.leaky_relu(y, slope)
        return y
