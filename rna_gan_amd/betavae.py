"""betaVAE with the reference's module layout / state_dict keys (src/betaVAE.py:18-42,63-107).

The GAN hot path uses ``encode`` in EVAL mode (Dropout = identity, BatchNorm1d with running
statistics), i.e. 3 x [Linear + BN1d + LeakyReLU(0.01)] then ``z_mu`` / ``z_logvar``
(src/wgan_loss.py:67-69,96-97).  Each layer is ONE HIP GEMM whose epilogue applies the folded
BatchNorm scale/shift and the activation (rg_linear_affine_act); with precision "bf16" the weights
are streamed as a packed bf16 copy (this path is weight-streaming/HBM bound).
``forward`` (train and eval mode, decoder included) is the betaVAE training path of SURVEY 8f row f4:
rna_gan_amd/vae_train.py.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn


_TOKENS = [0]


def _next_token():
    _TOKENS[0] += 1
    return _TOKENS[0]


class RNAEncoder(nn.Module):
    def __init__(self, in_channels: int, hidden_dims: List[int]):
        super().__init__()
        self.in_channels = in_channels
        modules: List[nn.Module] = [nn.Sequential(nn.Dropout())]
        for h in hidden_dims:
            modules.append(nn.Sequential(nn.Linear(in_channels, h), nn.BatchNorm1d(h), nn.LeakyReLU()))
            in_channels = h
        self.encoder = nn.Sequential(*modules)


class betaVAE(nn.Module):
    def __init__(self, in_channels: int, z_dim: int, encoder_dims: List[int], hidden_dims_decoder: List[int],
                 beta: float = 2, encoder_checkpoint=None):
        super().__init__()
        self.encoder = RNAEncoder(in_channels, encoder_dims)
        if encoder_checkpoint:
            self.encoder.load_state_dict(torch.load(encoder_checkpoint))
        self.z_mu = nn.Linear(z_dim, z_dim)
        self.z_logvar = nn.Linear(z_dim, z_dim)
        self.beta = beta
        mods: List[nn.Module] = []
        cin = z_dim
        for h in hidden_dims_decoder:
            mods.append(nn.Sequential(nn.Linear(cin, h), nn.BatchNorm1d(h), nn.LeakyReLU()))
            cin = h
        mods.append(nn.Sequential(nn.Linear(cin, in_channels), nn.Tanh()))
        self.decoder = nn.Sequential(*mods)
        self.z_dim = z_dim
        self.precision = "bf16"
        self._plan = None
        # provenance of the current weights (losses._LatentCache shares one encode between modules holding the SAME weights):
        # a token handed over by load_state_dict / state_dict / a checkpoint file, valid while no parameter or buffer has
        # been written since (tensor version counters) -- identity, not a fingerprint of the values
        self._token = None
        self._token_versions = None
        self._ops = None
        self._flat = None
        self._trt = None
        # parity hooks: a fixed Dropout keep-mask ([N][in_channels] uint8) / reparametrisation noise ([N][z_dim])
        # instead of device-side sampling
        self.fixed_mask = None
        self.fixed_eps = None

    # ---------------------------------------------------------------- runtime
    def set_precision(self, precision: str):
        self.precision = precision
        self._plan = None
        self._trt = None
        return self

    def _apply(self, fn, *a, **k):
        # .to() / .cuda() / .float(): the same weights in another place or type (buffers become NEW tensor objects, so the
        # version snapshot is retaken); device and dtype are part of the signature
        valid = self._token is not None and self._token_versions == self._versions()
        r = super()._apply(fn, *a, **k)
        if valid:
            self._token_versions = self._versions()
        self._plan = None
        self._ops = None
        self._flat = None
        self._trt = None
        return r

    # training runtime: parameters / gradients re-homed into flat buffers (fused Adam), kernel-level fwd/bwd
    def flat_params(self):
        from .models import FlatParams
        if self._flat is None or not self._flat.owns(self):
            self._flat = FlatParams(self)
            self._plan = None
        return self._flat

    @property
    def flat(self):
        return self.flat_params()

    def weights_changed(self, by_optimizer: bool = False):
        self._plan = None
        self._token = None          # kernels wrote the flat buffers (no tensor version counter sees that)

    def train(self, mode: bool = True):
        # the eval-mode plan folds BatchNorm into per-layer scale/shift vectors: rebuild it after any training phase
        # (a plain torch optimizer updates the parameter views without telling this module)
        self._plan = None
        if mode:
            self._token = None      # a training phase may rewrite the weights through the flat buffers
        return super().train(mode)

    def train_runtime(self):
        if self._trt is None:
            from .vae_train import VaeRuntime
            self._trt = VaeRuntime(self)
        return self._trt

    def _versions(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers()))

    def adopt_weights_token(self, token):
        """Declare that the weights now held are the ones identified by ``token`` (a hashable value: the same token on two
        modules means 'loaded from the same source, untouched since')."""
        self._token = token
        self._token_versions = self._versions()

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        if not a and not k.get("prefix"):
            # a full state_dict carries the identity of the weights it was read from: a module that loads it holds the
            # same weights as this one (bench.py / tests hand one module's state_dict to the other two loss plugins)
            if self._token is None or self._token_versions != self._versions():
                self.adopt_weights_token(("module", id(self), _next_token()))
            sd._rg_weights_token = self._token
        return sd

    def load_state_dict(self, state_dict, *a, **k):
        r = super().load_state_dict(state_dict, *a, **k)
        self._plan = None
        token = getattr(state_dict, "_rg_weights_token", None)
        if token is not None and not r.missing_keys and not r.unexpected_keys:
            self.adopt_weights_token(token)
        else:
            self._token = None
        return r

    def load_checkpoint_file(self, path):
        """load_state_dict(torch.load(path)) with the file's identity as the weights token: the three loss plugins of
        --loss_type wganvae each load the same checkpoint (src/wgan_loss.py:67-69, :159-161, :289-291)."""
        import os
        st = os.stat(path)
        r = self.load_state_dict(torch.load(path, map_location="cpu"))
        self.adopt_weights_token(("file", os.path.realpath(path), st.st_mtime_ns, st.st_size))
        return r

    def __getstate__(self):          # pickled inside loss objects in checkpoints: drop runtime handles
        d = dict(self.__dict__)
        d["_plan"] = None
        d["_ops"] = None
        d["_flat"] = None
        d["_trt"] = None
        return d

    def _build_plan(self):
        from .ops_hip import HipOps
        dev = self.z_mu.weight.device
        if dev.type != "cuda":
            raise RuntimeError("betaVAE.encode runs on the HIP kernels only (move the module to a ROCm GPU)")
        # precision "fp16": the fp16 build of the library (fp16 operand images, v_mfma_*_f16); "fp32": unpacked fp32 weights
        self._ops = HipOps(torch.float16 if self.precision == "fp16" else torch.bfloat16, dev)
        packed = self.precision in ("bf16", "fp16")
        plan = []
        with torch.no_grad():
            for blk in list(self.encoder.encoder.children())[1:]:
                lin, bn, act = blk[0], blk[1], blk[2]
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)      # one-time fold, C-length vectors
                shift = (lin.bias - bn.running_mean) * scale + bn.bias
                plan.append((lin.weight.detach(), self._ops.pack_linear(lin.weight.detach()) if packed else None,
                             scale.contiguous(), shift.contiguous(), float(act.negative_slope)))
            for lin in (self.z_mu, self.z_logvar):
                plan.append((lin.weight.detach(), self._ops.pack_linear(lin.weight.detach()) if packed else None,
                             None, lin.bias.detach(), 1.0))
        self._plan = plan

    def signature(self):
        """Identity of the eval-mode encoder's weights as the loss plugins see it (losses._LatentCache: the reference builds
        three copies from one checkpoint; copies with equal signatures hold the same weights and produce equal latents).
        (precision, weights token) while the weights are untouched since the token was adopted; otherwise private to this
        module and its current tensor versions -- never equal to another module's.

        LIMIT: the token is validated by the tensors' VERSION counters.  A write that bypasses them --
        ``p.data.copy_(...)``, ``p.data = ...``, a collective on ``p.data``, a custom optimizer stepping ``.data`` -- leaves
        the token looking valid.  Whoever writes weights that way must call ``weights_changed()`` afterwards (the CLI's
        broadcast does, then ``adopt_weights_token``); ``load_state_dict``, ``.to()`` and in-place ops on the parameters
        themselves are tracked.  A value fingerprint would need a device read per batch here, which is what the token
        replaced."""
        versions = self._versions()
        w = self.z_mu.weight
        if self._token is not None and self._token_versions == versions:
            return (self.precision, self._token, str(w.dtype), str(w.device))
        return (self.precision, ("private", id(self)), versions)

    def encode(self, x, mean_only=False):
        """(z_mean, z_log_var, x_encoded) as src/betaVAE.py:102-107, eval mode.  mean_only: skip z_log_var (returned as
        None) -- the loss plugins only use z_mean (``z, _, _ = betavae.encode(rna)``, src/wgan_loss.py:96-97), the
        reference computes and discards the other head."""
        if self.training:
            # train mode (src/betaVAE.py:102-107 with the module in train(): Dropout active, BatchNorm1d on batch
            # statistics with running-statistics update): the encoder half of the training forward, without autograd
            # (gradients flow through forward(), the path train_betaVAE uses)
            rt = self.train_runtime()
            N = x.shape[0]
            if self.fixed_mask is not None:
                mask = self.fixed_mask
            else:
                mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - rt.p_drop)
            return rt.encode_train(x, mask.to(device=x.device, dtype=torch.uint8).contiguous(), mean_only)
        if self._plan is None:
            self._build_plan()
        ops = self._ops
        h = x.contiguous().float()
        for (w, wp, scale, shift, slope) in self._plan[:-2]:
            h = ops.linear_affine_act(h, w, scale, shift, slope, wp=wp)
        w, wp, scale, shift, slope = self._plan[-2]
        z_mean = ops.linear_affine_act(h, w, scale, shift, slope, wp=wp)
        if mean_only:
            return z_mean, None, h
        w, wp, scale, shift, slope = self._plan[-1]
        z_log_var = ops.linear_affine_act(h, w, scale, shift, slope, wp=wp)
        return z_mean, z_log_var, h

    def forward(self, x):
        """(out, z_mean, z_log_var) as src/betaVAE.py:108-114: train mode = Dropout + batch-statistics BatchNorm,
        differentiable through one autograd.Function (parameter gradients land in the flat gradient buffer)."""
        from .vae_train import vae_forward
        return vae_forward(self, x)

    def decode(self, z):
        if self.training:
            raise NotImplementedError("betaVAE.decode: eval mode only (the train-mode decoder runs inside forward)")
        return self.train_runtime().decode_eval(z)[:, :self.encoder.in_channels]

    def sample(self, num_samples: int, current_device, interpolation=None, alpha: float = 1.0):
        """src/betaVAE.py:116-140: decode N(0, I) latents (optionally shifted by alpha * interpolation)."""
        z = torch.randn(num_samples, self.z_dim).to(current_device)
        if interpolation is not None:
            z = z + torch.from_numpy(alpha * interpolation).float().to(current_device)
        return self.decode(z)
