"""betaVAE training / decoding on the HIP kernels.

Mirrors src/betaVAE.py: ``betaVAE.forward`` (:108-114), ``betaVAEloss`` (:145-163), ``train_betaVAE`` (:166-284),
``evaluate_betaVAE`` (:286-330).  The model is autograd-free inside: ``forward`` is ONE ``torch.autograd.Function``
whose backward runs the hand-written backward pass and writes every parameter gradient straight into the flat
gradient buffer behind the parameters' ``.grad`` views; ``betaVAEloss`` is a second Function (fused loss + its three
gradients).  The reference loop ``loss['total_loss'].backward(); optimizer.step()`` therefore runs unchanged.

A Linear layer's three GEMMs (forward, data gradient, weight gradient) are all "NT" GEMMs on re-laid operands
(include/rnagan_hip.h, betaVAE TRAINING section); BatchNorm1d (train mode) + LeakyReLU reuse the row kernels of the
GAN path on [N][C] rows.  precision "bf16": bf16 MFMA operands / fp32 accumulate, everything between the GEMMs fp32;
"fp32": the functor GEMM (parity mode).
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from . import _abi
from ._abi import check, RG_BF16


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _ceil64(v: int) -> int:
    return (v + 63) // 64 * 64


class _Saved:
    pass


class VaeRuntime:
    """Kernel-level forward / backward of one betaVAE on its device."""

    def __init__(self, model):
        from .ops_hip import HipOps
        dev = model.z_mu.weight.device
        if dev.type != "cuda":
            raise RuntimeError("betaVAE training runs on the HIP kernels only (move the module to a ROCm GPU)")
        self.model = model
        self.ops = HipOps(torch.float32, dev)          # fp32 rows between the GEMMs
        self.lib, self.dev = self.ops.lib, dev
        self.bf16 = model.precision == "bf16"
        # rna_gan_amd.optim.Adam.bind(model, fuse_linear_wgrad=True): the Linear weight gradients are formed inside the
        # optimizer step from the operands linear_bwd leaves behind (bf16 kernels only)
        self.fuse_wgrad = self.bf16 and bool(getattr(model, "_fuse_linear_wgrad", False))
        self._packs = {}
        self.enc = [(b[0], b[1], float(b[2].negative_slope)) for b in list(model.encoder.encoder.children())[1:]]
        dec = list(model.decoder.children())
        self.dec = [(b[0], b[1], float(b[2].negative_slope)) for b in dec[:-1]]
        self.out_lin = dec[-1][0]
        self.p_drop = float(list(model.encoder.encoder.children())[0][0].p)

    # ------------------------------------------------------------------ operands of the NT GEMM
    def ld(self, F: int) -> int:
        """row stride of the [N][F] input/output rows (zero padded to the bf16 GEMM's K granule)"""
        return _ceil64(F) if self.bf16 else F

    def opA(self, A):
        """fp32 [M][K] (dense) -> A operand"""
        if not self.bf16:
            return A
        M, K = A.shape
        a = torch.empty((M, _ceil64(K)), dtype=torch.bfloat16, device=self.dev)
        check(self.lib.rg_cast_pad(_ptr(A), _ptr(a), M, K, a.shape[1], RG_BF16, self.ops.stream), "rg_cast_pad")
        return a

    def opB(self, W):
        """fp32 [Nout][K] (dense, Linear.weight layout) -> B operand.  With fuse_wgrad the fused Adam pass keeps a bf16
        operand image per weight current (rg_linear_wgrad_adam's wpack); it is trusted while the tensor's version counter is
        the one the optimizer saw (any torch-side write -- load_state_dict, another optimizer -- bumps it: repacked then)."""
        if not self.bf16:
            return W
        pk = self._packs.get(id(W)) if self.fuse_wgrad else None
        if pk is not None and pk["version"] == W._version:
            return pk["image"]
        img = self.ops.pack_linear(W)
        if self.fuse_wgrad:
            if pk is None:
                self._packs[id(W)] = {"image": img, "version": None}       # padding already zero; adopted by the next step
            else:
                pk["image"].copy_(img)
                pk["version"] = None
                img = pk["image"]
        return img

    def opT(self, S):
        """fp32 [R][C] (dense) -> S^T as an operand: [C][R] (bf16: R zero padded to a multiple of 64)"""
        R, C = S.shape
        if self.bf16:
            t = torch.empty((C, _ceil64(R)), dtype=torch.bfloat16, device=self.dev)
            check(self.lib.rg_transpose_pack_bf16(_ptr(S), _ptr(t), R, C, t.shape[1], C, self.ops.stream),
                  "rg_transpose_pack_bf16")
        else:
            t = torch.empty((C, R), dtype=torch.float32, device=self.dev)
            check(self.lib.rg_transpose_f32(_ptr(S), _ptr(t), R, C, self.ops.stream), "rg_transpose_f32")
        return t

    def mm(self, a, b, M, Nout, y, ldy, shift=None, scale=None, slope=1.0):
        """y[M][:Nout] (row stride ldy) = act((a[:M] . b[:Nout]^T) * scale + shift)"""
        K = a.shape[1]
        assert b.shape[1] == K and a.shape[0] >= M and b.shape[0] >= Nout and ldy >= Nout
        if self.bf16:
            ws = self.ops._ws(self.lib.rg_gemm_nt_bf16_workspace_bytes(M, K, Nout))
            check(self.lib.rg_gemm_nt_bf16(_ptr(a), _ptr(b), _ptr(scale), _ptr(shift), _ptr(y), ldy, M, K, Nout,
                                           float(slope), _ptr(ws), ws.numel(), self.ops.stream), "rg_gemm_nt_bf16")
        else:
            ws = self.ops._ws(256)
            check(self.lib.rg_linear_affine_act(_ptr(a), K, _ptr(b), 0, _ptr(scale), _ptr(shift), _ptr(y), ldy, M, K, Nout,
                                                float(slope), _abi.ALGO_GENERIC, _ptr(ws), ws.numel(), self.ops.stream),
                  "rg_linear_affine_act")
        return y

    # ------------------------------------------------------------------ layers
    def linear(self, h, lin, ldy=None, scale=None, shift=None, slope=1.0):
        N = h.shape[0]
        out_f = lin.weight.shape[0]
        ldy = ldy or out_f
        y = (torch.zeros if ldy != out_f else torch.empty)((N, ldy), dtype=torch.float32, device=self.dev)
        return self.mm(self.opA(h), self.opB(lin.weight), N, out_f, y, ldy, shift=lin.bias if shift is None else shift,
                       scale=scale, slope=slope)

    def linear_bwd(self, lin, hin, gy, need_dx: bool):
        """hin [N][K>=in] (the layer's input rows), gy [N][C>=out]: writes lin.weight.grad / lin.bias.grad, returns dx"""
        out_f, in_f = lin.weight.shape
        N = gy.shape[0]
        if self.fuse_wgrad:
            # dW = gy^T . hin is formed inside the Adam pass of this weight (rg_linear_wgrad_adam): its transposed bf16
            # operands -- the ones the GEMM below would take -- are left on the model for optimizer.step()
            self.model._rg_pending_linear.append((lin.weight, self.opT(gy), self.opT(hin), N, self._packs.get(id(lin.weight))))
        else:
            self.mm(self.opT(gy), self.opT(hin), out_f, in_f, lin.weight.grad, in_f)      # dW = gy^T . hin
        if gy.shape[1] == out_f:
            self.ops.col_sum(gy, lin.bias.grad, False)
        else:
            tmp = torch.empty(gy.shape[1], dtype=torch.float32, device=self.dev)
            self.ops.col_sum(gy, tmp, False)
            lin.bias.grad.copy_(tmp[:out_f])
        if not need_dx:
            return None
        gx = torch.empty((N, in_f), dtype=torch.float32, device=self.dev)
        return self.mm(self.opA(gy), self.opT(lin.weight), N, in_f, gx, in_f)             # dx = gy . W

    def pad_rows(self, x, ld, mask=None, scale=1.0):
        N, F = x.shape
        y = torch.empty((N, ld), dtype=torch.float32, device=self.dev)
        check(self.lib.rg_vae_dropout(_ptr(x), _ptr(mask), _ptr(y), N, F, ld, float(scale), self.ops.stream),
              "rg_vae_dropout")
        return y

    def _bn(self, z, bn, slope):
        return self.ops.bn_forward(z, bn.weight, bn.bias, slope, bn.eps, 0.1 if bn.momentum is None else bn.momentum,
                                   bn.running_mean, bn.running_var, bn.num_batches_tracked)

    def encode_train(self, x, mask, mean_only=False):
        """Train-mode betaVAE.encode (src/betaVAE.py:102-107): Dropout keep-mask, batch-statistics BatchNorm1d (running
        statistics updated), forward only -> (z_mean, z_log_var, x_encoded)."""
        m = self.model
        x = x.contiguous().float()
        h = self.pad_rows(x, self.ld(x.shape[1]), mask, 1.0 / (1.0 - self.p_drop))
        with torch.no_grad():
            for lin, bn, slope in self.enc:
                h, _, _ = self._bn(self.linear(h, lin), bn, slope)
            mu = self.linear(h, m.z_mu)
            lv = None if mean_only else self.linear(h, m.z_logvar)
        return mu, lv, h

    # ------------------------------------------------------------------ train-mode forward / backward
    def forward_train(self, x, mask, eps):
        m = self.model
        x = x.contiguous().float()
        N, F = x.shape
        ld = self.ld(F)
        sv = _Saved()
        h = self.pad_rows(x, ld, mask, 1.0 / (1.0 - self.p_drop))      # nn.Dropout (train): keep-mask * 1/(1-p)
        sv.enc = []
        for lin, bn, slope in self.enc:
            z = self.linear(h, lin)
            a, mean, invstd = self._bn(z, bn, slope)
            sv.enc.append((h, z, mean, invstd))
            h = a
        sv.h3 = h
        sv.mu = self.linear(h, m.z_mu)
        sv.lv = self.linear(h, m.z_logvar)
        sv.eps = eps.contiguous().float()
        zl = torch.empty_like(sv.mu)
        check(self.lib.rg_vae_reparam(_ptr(sv.mu), _ptr(sv.lv), _ptr(sv.eps), _ptr(zl), zl.numel(), self.ops.stream),
              "rg_vae_reparam")
        h = zl
        sv.dec = []
        for lin, bn, slope in self.dec:
            z = self.linear(h, lin)
            a, mean, invstd = self._bn(z, bn, slope)
            sv.dec.append((h, z, mean, invstd))
            h = a
        sv.hd = h
        sv.out = self.linear(h, self.out_lin, ldy=ld)                  # [N][ld]; pad columns are zero (tanh(0) = 0)
        check(self.lib.rg_tanh_inplace(_ptr(sv.out), sv.out.numel(), self.ops.stream), "rg_tanh_inplace")
        return sv

    def backward(self, sv, g_out, g_mu, g_lv):
        """g_out [N][ld] dense (d loss / d out, pad columns ignored), g_mu / g_lv [N][Z] or None."""
        m, ops = self.model, self.ops
        m._rg_pending_linear = []                 # (weight, gT, xT, N) per Linear layer when fuse_wgrad
        gz = torch.empty_like(sv.out)
        check(self.lib.rg_tanh_bwd(_ptr(g_out), _ptr(sv.out), _ptr(gz), gz.numel(), ops.stream), "rg_tanh_bwd")
        gh = self.linear_bwd(self.out_lin, sv.hd, gz, True)
        for (lin, bn, slope), (hin, z, mean, invstd) in zip(reversed(self.dec), reversed(sv.dec)):
            gzz, _, _ = ops.bn_act_bwd(z, gh, mean, invstd, bn.weight, bn.bias, slope, bn.weight.grad, bn.bias.grad, False)
            gh = self.linear_bwd(lin, hin, gzz, True)
        gmu, glv = torch.empty_like(sv.mu), torch.empty_like(sv.lv)
        check(self.lib.rg_vae_reparam_bwd(_ptr(gh), _ptr(sv.lv), _ptr(sv.eps), _ptr(g_mu), _ptr(g_lv), _ptr(gmu), _ptr(glv),
                                          gmu.numel(), ops.stream), "rg_vae_reparam_bwd")
        gh = self.linear_bwd(m.z_mu, sv.h3, gmu, True)
        gh2 = self.linear_bwd(m.z_logvar, sv.h3, glv, True)
        check(self.lib.rg_add_inplace(_ptr(gh), _ptr(gh2), gh.numel(), ops.stream), "rg_add_inplace")
        k = len(self.enc)
        for (lin, bn, slope), (hin, z, mean, invstd) in zip(reversed(self.enc), reversed(sv.enc)):
            k -= 1
            gzz, _, _ = ops.bn_act_bwd(z, gh, mean, invstd, bn.weight, bn.bias, slope, bn.weight.grad, bn.bias.grad, False)
            gh = self.linear_bwd(lin, hin, gzz, k > 0)

    # ------------------------------------------------------------------ eval-mode forward (BatchNorm folded, no dropout)
    def _folded(self, lin, bn):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return scale.contiguous(), ((lin.bias - bn.running_mean) * scale + bn.bias).contiguous()

    def decode_eval(self, z):
        h = z.contiguous().float()
        with torch.no_grad():
            for lin, bn, slope in self.dec:
                scale, shift = self._folded(lin, bn)
                h = self.linear(h, lin, scale=scale, shift=shift, slope=slope)
            F = self.out_lin.weight.shape[0]
            out = self.linear(h, self.out_lin, ldy=self.ld(F))
            check(self.lib.rg_tanh_inplace(_ptr(out), out.numel(), self.ops.stream), "rg_tanh_inplace")
        return out

    def loss(self, x, xr_pad, ld, mu, lv, beta, training):
        N, F = x.shape
        xp = x.contiguous().float() if ld == F else self.pad_rows(x.contiguous().float(), ld)
        losses = torch.empty(3, dtype=torch.float32, device=self.dev)
        gxr = torch.empty((N, ld), dtype=torch.float32, device=self.dev)
        gmu, glv = torch.empty_like(mu), torch.empty_like(lv)
        ws = self.ops._ws(self.lib.rg_vae_loss_workspace_bytes())
        check(self.lib.rg_vae_loss(_ptr(xp), _ptr(xr_pad), N, F, ld, _ptr(mu), _ptr(lv), mu.shape[1], float(beta),
                                   int(bool(training)), _ptr(losses), _ptr(gxr), _ptr(gmu), _ptr(glv), _ptr(ws), ws.numel(),
                                   self.ops.stream), "rg_vae_loss")
        return losses, gxr, gmu, glv


def _padded_base(t, ld):
    """t is the [:, :F] view of a dense [N][ld] buffer -> that buffer (else None)"""
    b = t._base
    if b is not None and b.dim() == 2 and b.shape == (t.shape[0], ld) and b.is_contiguous() and t.stride() == (ld, 1) \
            and t.data_ptr() == b.data_ptr():
        return b
    return None


def _dense_rows(rt, t, ld):
    """[N][F] tensor -> dense fp32 [N][ld] rows (pad columns zero or ignored by the consumers)"""
    N, F = t.shape
    if ld == F:
        return t.contiguous().float()
    b = _padded_base(t, ld)
    return b if b is not None else rt.pad_rows(t.contiguous().float(), ld)


class _VaeForwardFn(torch.autograd.Function):
    """(out, z_mean, z_log_var) = betaVAE(x) in train mode; `anchor` is a parameter, so that autograd calls backward."""

    @staticmethod
    def forward(ctx, x, anchor, model, mask, eps):
        rt = model.train_runtime()
        sv = rt.forward_train(x, mask, eps)
        ctx.rt, ctx.sv = rt, sv
        return sv.out[:, :x.shape[1]], sv.mu, sv.lv

    @staticmethod
    def backward(ctx, g_out, g_mu, g_lv):
        rt, sv = ctx.rt, ctx.sv
        N, ld = sv.out.shape
        if g_out is None:
            g_out = torch.zeros((N, ld), dtype=torch.float32, device=sv.out.device)
        else:
            g_out = _dense_rows(rt, g_out, ld)
        rt.backward(sv, g_out, None if g_mu is None else g_mu.contiguous(), None if g_lv is None else g_lv.contiguous())
        ctx.sv = None
        return None, None, None, None, None


class _VaeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_recons, z_mean, z_logvar, beta, training, rt):
        N, F = x.shape
        ld = rt.ld(F)
        xr = _dense_rows(rt, x_recons.detach(), ld)
        losses, gxr, gmu, glv = rt.loss(x, xr, ld, z_mean.detach().contiguous(), z_logvar.detach().contiguous(), beta,
                                        training)
        ctx.g = (gxr, gmu, glv, F)
        total, rec, kl = losses[0].clone(), losses[1].clone(), losses[2].clone()
        ctx.mark_non_differentiable(rec, kl)
        return total, rec, kl

    @staticmethod
    def backward(ctx, gt, _gr, _gk):
        gxr, gmu, glv, F = ctx.g
        return None, (gxr * gt)[:, :F], gmu * gt, glv * gt, None, None, None


def betaVAEloss(x, x_recons, z_mean, z_logvar, beta, kld_weight=0.005, training=True):
    """src/betaVAE.py:145-163: {'total_loss', 'reconstruction_loss', 'kl_loss'} (kld_weight is unused there too).
    One fused kernel pair computes the three scalars and d total / d (x_recons, z_mean, z_logvar)."""
    rt = getattr(x_recons, "_vae_rt", None) or _runtime_for(x_recons.device)
    total, rec, kl = _VaeLossFn.apply(x, x_recons, z_mean, z_logvar, float(beta), bool(training), rt)
    return {"total_loss": total, "reconstruction_loss": rec, "kl_loss": kl}


_RT_BY_DEVICE = {}


def _runtime_for(device):
    """the runtime of the model that last ran forward on this device (the loss needs its precision / row padding)"""
    rt = _RT_BY_DEVICE.get(str(device))
    if rt is None:
        raise RuntimeError("betaVAEloss: run the rna_gan_amd betaVAE forward on this device first")
    return rt


def vae_forward(model, x):
    """betaVAE.forward (src/betaVAE.py:108-114) on the HIP kernels, train or eval mode."""
    rt = model.train_runtime()
    _RT_BY_DEVICE[str(x.device)] = rt
    N = x.shape[0]
    eps = model.fixed_eps if model.fixed_eps is not None else torch.randn((N, model.z_dim), device=x.device)
    if model.training:
        if model.fixed_mask is not None:
            mask = model.fixed_mask
        else:
            mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - rt.p_drop)
        mask = mask.to(device=x.device, dtype=torch.uint8).contiguous()
        model.flat_params()                                   # parameters / gradients live in the flat buffers
        if torch.is_grad_enabled():
            out, mu, lv = _VaeForwardFn.apply(x, model.z_mu.bias, model, mask, eps.to(x.device))
        else:
            sv = rt.forward_train(x, mask, eps.to(x.device))
            out, mu, lv = sv.out[:, :x.shape[1]], sv.mu, sv.lv
        out._vae_rt = rt                                      # betaVAEloss picks the model's runtime up from here
        return out, mu, lv
    z_mean, z_log_var, _ = model.encode(x)
    z = torch.empty_like(z_mean)
    e = eps.to(x.device).contiguous().float()
    check(rt.lib.rg_vae_reparam(_ptr(z_mean), _ptr(z_log_var), _ptr(e), _ptr(z), z.numel(), rt.ops.stream),
          "rg_vae_reparam")
    out = rt.decode_eval(z)[:, :x.shape[1]]
    out._vae_rt = rt
    return out, z_mean, z_log_var


# --------------------------------------------------------------------------------------------------------------
# training / evaluation loops (host logic of src/betaVAE.py:166-330)
# --------------------------------------------------------------------------------------------------------------
_KEYS = ("total_loss", "reconstruction_loss", "kl_loss")


def _to_device(batch, device):
    x = batch["rna_data"]
    return x.to(device, non_blocking=True) if device is not None else (x.cuda() if torch.cuda.is_available() else x)


def train_betaVAE(model, optimizer, dataloader, save_dir="checkpoints/models/", device=None, log_interval=100,
                  summary_writer=None, num_epochs=100, scheduler=None, verbose=True):
    """Epoch loop with a 'train' and a 'val' phase; keeps the state_dict with the best validation total loss as
    ``model_dict_best.pt``, the last one as ``model_last.pt`` and reloads the best before returning (model, results).
    ``optimizer``: rna_gan_amd.Adam(...).bind(model) runs the fused HIP step; any torch optimizer works on the views."""
    os.makedirs(save_dir, exist_ok=True)
    if hasattr(optimizer, "note_replayed") and getattr(optimizer, "_module", None) is model and model.precision == "bf16":
        # this loop steps right after backward and reads no .grad in between: the Linear weight gradients are formed inside
        # the optimizer step (rg_linear_wgrad_adam)
        optimizer.bind(model, fuse_linear_wgrad=True)
    best = {"total_loss": float("inf")}
    best_epoch = 0
    history = {ph: {k: [] for k in _KEYS} for ph in ("train", "val")}
    steps = {"train": 0, "val": 0}
    dev = next(model.parameters()).device
    for epoch in range(num_epochs):
        if verbose:
            print("Epoch {}/{}".format(epoch, num_epochs - 1))
            print("-" * 10)
        for phase in ("train", "val"):
            model.train(phase == "train")
            running = {k: [] for k in _KEYS}
            logged = {k: 0.0 for k in _KEYS}
            for batch in dataloader[phase]:
                x = _to_device(batch, dev)
                optimizer.zero_grad(set_to_none=True)
                with torch.set_grad_enabled(phase == "train"):
                    out, z_mean, z_log_var = model(x)
                    losses = betaVAEloss(x, out, z_mean, z_log_var, model.beta, training=model.training)
                if phase == "train":
                    losses["total_loss"].backward()
                    optimizer.step()
                    if scheduler:
                        scheduler.step()
                steps[phase] += 1
                for k in _KEYS:
                    running[k].append(losses[k].detach().item())
                if summary_writer is not None and steps[phase] % log_interval == 0:
                    for k in _KEYS:
                        cur = float(np.mean(running[k]))
                        summary_writer.add_scalar("{}/{}".format(phase, k), cur - logged[k], steps[phase])
                        logged[k] = cur
            epoch_loss = {k: float(np.mean(running[k])) if running[k] else float("nan") for k in _KEYS}
            for k in _KEYS:
                history[phase][k].append(epoch_loss[k])
            if verbose:
                print("{} Total Loss: {:.4f} | Reconstruction Loss: {:.4f} | KL Loss: {:.4f}".format(
                    phase, epoch_loss["total_loss"], epoch_loss["reconstruction_loss"], epoch_loss["kl_loss"]))
            if phase == "val" and epoch_loss["total_loss"] < best["total_loss"]:
                best["total_loss"] = epoch_loss["total_loss"]
                torch.save(copy.deepcopy(model.state_dict()), os.path.join(save_dir, "model_dict_best.pt"))
                best_epoch = epoch
    torch.save(model.state_dict(), os.path.join(save_dir, "model_last.pt"))
    best_path = os.path.join(save_dir, "model_dict_best.pt")
    if os.path.exists(best_path):
        model.load_state_dict(torch.load(best_path))
    return model, {"best_epoch": best_epoch, "best_loss": best, "history": history}


def evaluate_betaVAE(model, dataloader, verbose=True):
    """(mean losses, predictions, inputs) over a dataloader in eval mode (src/betaVAE.py:286-330)."""
    model.eval()
    dev = next(model.parameters()).device
    running = {k: [] for k in _KEYS}
    predictions, real = [], []
    for batch in dataloader:
        x = _to_device(batch, dev)
        with torch.no_grad():
            out, z_mean, z_log_var = model(x)
            losses = betaVAEloss(x, out, z_mean, z_log_var, model.beta, training=False)
        predictions.append(out.detach().cpu().numpy().tolist())
        real.append(x.detach().cpu().numpy().tolist())
        for k in _KEYS:
            running[k].append(losses[k].detach().item())
    test_loss = {k: float(np.mean(running[k])) for k in _KEYS}
    if verbose:
        print("Total Loss: {:.4f} | Reconstruction Loss: {:.4f} | KL Loss: {:.4f}".format(
            test_loss["total_loss"], test_loss["reconstruction_loss"], test_loss["kl_loss"]))
    return test_loss, predictions, real
