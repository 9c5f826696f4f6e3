"""Inception-v3 pool features for the FID metric on the HIP kernels.

The reference's FID (src/fid.py:33-94) runs torchvision's ``inception_v3(pretrained=True)`` up to ``Mixed_7c`` (forward
hook, :39-45), averages the (N, 2048, 8, 8) map to (N, 2048) (:60-64) and feeds images in [0, 1] scaled to [-1, 1] (:55);
``pretrained=True`` also switches torchvision's ``transform_input`` on (its ImageNet re-normalisation of the three
channels).  Pretrained weights cannot be downloaded here; this module provides everything around them:

* ``InceptionV3`` -- an ``nn.Module`` with torchvision's attribute names and parameter shapes, so ``load_state_dict`` takes
  a torchvision ``inception_v3`` checkpoint as it is (``AuxLogits.*`` and ``fc.*`` are present for that reason; neither is
  evaluated: the reference reads Mixed_7c in eval mode);
* ``InceptionV3.features(x)`` -- the forward pass on the C ABI: NHWC fp32 activations, every BasicConv2d (Conv2d without
  bias + eval-mode BatchNorm2d(eps 1e-3) + ReLU) ONE GEMM with the folded BatchNorm affine and the ReLU in its epilogue
  (rg_linear_affine_act, slope 0) over the patch matrix of rg_im2col_nhwc (1x1 convolutions read the activation in
  place), the branches of a block writing straight into their channel slice of the block's output (no concatenation
  pass), pooling by rg_pool2d_nhwc, the input transform by rg_nchw_to_nhwc_affine, the final average by
  rg_spatial_mean_nhwc.  The architecture is held as a TABLE (``_BLOCKS``) that one small executor walks -- written
  independently of the module-per-block restatement in oracle/inception_ref.py it is tested against.

``fid.inception_feature_extractor(state_dict_or_path)`` wraps it as the ``feature_extractor`` of ``fid.calculate_fid``.
Parity with torchvision itself is UNPINNED (package and weights absent); with seeded random weights the HIP path agrees
with the oracle restatement to fp32 round-off (tests/test_inception_gpu.py), and the state_dict key / shape manifest is
checked on the CPU (tests/test_inception_cpu.py).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _abi
from ._abi import check

# (name, cout, (kh, kw), (sh, sw), (ph, pw)) of every BasicConv2d of a block, and how the block wires them:
#   ("conv", name, src)            src -> conv -> new buffer                    (src: "x" = block input or an earlier name)
#   ("out", [sources...])          channel-concatenated block output; a source is a conv name or ("avgpool"|"maxpool", ...)
# Convs listed under "out" write directly into their slice of the output buffer.
_A = lambda pf: {"convs": [("branch1x1", 64, (1, 1), (1, 1), (0, 0)), ("branch5x5_1", 48, (1, 1), (1, 1), (0, 0)),
                           ("branch5x5_2", 64, (5, 5), (1, 1), (2, 2)), ("branch3x3dbl_1", 64, (1, 1), (1, 1), (0, 0)),
                           ("branch3x3dbl_2", 96, (3, 3), (1, 1), (1, 1)), ("branch3x3dbl_3", 96, (3, 3), (1, 1), (1, 1)),
                           ("branch_pool", pf, (1, 1), (1, 1), (0, 0))],
                 "wire": {"branch1x1": "x", "branch5x5_1": "x", "branch5x5_2": "branch5x5_1", "branch3x3dbl_1": "x",
                          "branch3x3dbl_2": "branch3x3dbl_1", "branch3x3dbl_3": "branch3x3dbl_2", "branch_pool": "avgpool(x)"},
                 "out": ["branch1x1", "branch5x5_2", "branch3x3dbl_3", "branch_pool"]}
_B = {"convs": [("branch3x3", 384, (3, 3), (2, 2), (0, 0)), ("branch3x3dbl_1", 64, (1, 1), (1, 1), (0, 0)),
                ("branch3x3dbl_2", 96, (3, 3), (1, 1), (1, 1)), ("branch3x3dbl_3", 96, (3, 3), (2, 2), (0, 0))],
      "wire": {"branch3x3": "x", "branch3x3dbl_1": "x", "branch3x3dbl_2": "branch3x3dbl_1", "branch3x3dbl_3": "branch3x3dbl_2"},
      "out": ["branch3x3", "branch3x3dbl_3", "maxpool(x)"]}
_C = lambda c7: {"convs": [("branch1x1", 192, (1, 1), (1, 1), (0, 0)), ("branch7x7_1", c7, (1, 1), (1, 1), (0, 0)),
                           ("branch7x7_2", c7, (1, 7), (1, 1), (0, 3)), ("branch7x7_3", 192, (7, 1), (1, 1), (3, 0)),
                           ("branch7x7dbl_1", c7, (1, 1), (1, 1), (0, 0)), ("branch7x7dbl_2", c7, (7, 1), (1, 1), (3, 0)),
                           ("branch7x7dbl_3", c7, (1, 7), (1, 1), (0, 3)), ("branch7x7dbl_4", c7, (7, 1), (1, 1), (3, 0)),
                           ("branch7x7dbl_5", 192, (1, 7), (1, 1), (0, 3)), ("branch_pool", 192, (1, 1), (1, 1), (0, 0))],
                 "wire": {"branch1x1": "x", "branch7x7_1": "x", "branch7x7_2": "branch7x7_1", "branch7x7_3": "branch7x7_2",
                          "branch7x7dbl_1": "x", "branch7x7dbl_2": "branch7x7dbl_1", "branch7x7dbl_3": "branch7x7dbl_2",
                          "branch7x7dbl_4": "branch7x7dbl_3", "branch7x7dbl_5": "branch7x7dbl_4", "branch_pool": "avgpool(x)"},
                 "out": ["branch1x1", "branch7x7_3", "branch7x7dbl_5", "branch_pool"]}
_D = {"convs": [("branch3x3_1", 192, (1, 1), (1, 1), (0, 0)), ("branch3x3_2", 320, (3, 3), (2, 2), (0, 0)),
                ("branch7x7x3_1", 192, (1, 1), (1, 1), (0, 0)), ("branch7x7x3_2", 192, (1, 7), (1, 1), (0, 3)),
                ("branch7x7x3_3", 192, (7, 1), (1, 1), (3, 0)), ("branch7x7x3_4", 192, (3, 3), (2, 2), (0, 0))],
      "wire": {"branch3x3_1": "x", "branch3x3_2": "branch3x3_1", "branch7x7x3_1": "x", "branch7x7x3_2": "branch7x7x3_1",
               "branch7x7x3_3": "branch7x7x3_2", "branch7x7x3_4": "branch7x7x3_3"},
      "out": ["branch3x3_2", "branch7x7x3_4", "maxpool(x)"]}
_E = {"convs": [("branch1x1", 320, (1, 1), (1, 1), (0, 0)), ("branch3x3_1", 384, (1, 1), (1, 1), (0, 0)),
                ("branch3x3_2a", 384, (1, 3), (1, 1), (0, 1)), ("branch3x3_2b", 384, (3, 1), (1, 1), (1, 0)),
                ("branch3x3dbl_1", 448, (1, 1), (1, 1), (0, 0)), ("branch3x3dbl_2", 384, (3, 3), (1, 1), (1, 1)),
                ("branch3x3dbl_3a", 384, (1, 3), (1, 1), (0, 1)), ("branch3x3dbl_3b", 384, (3, 1), (1, 1), (1, 0)),
                ("branch_pool", 192, (1, 1), (1, 1), (0, 0))],
      "wire": {"branch1x1": "x", "branch3x3_1": "x", "branch3x3_2a": "branch3x3_1", "branch3x3_2b": "branch3x3_1",
               "branch3x3dbl_1": "x", "branch3x3dbl_2": "branch3x3dbl_1", "branch3x3dbl_3a": "branch3x3dbl_2",
               "branch3x3dbl_3b": "branch3x3dbl_2", "branch_pool": "avgpool(x)"},
      "out": ["branch1x1", "branch3x3_2a", "branch3x3_2b", "branch3x3dbl_3a", "branch3x3dbl_3b", "branch_pool"]}

# the trunk in execution order: ("conv", name, cout, k, s, p) | ("maxpool",) | ("block", name, spec)
_TRUNK = [("conv", "Conv2d_1a_3x3", 32, (3, 3), (2, 2), (0, 0)), ("conv", "Conv2d_2a_3x3", 32, (3, 3), (1, 1), (0, 0)),
          ("conv", "Conv2d_2b_3x3", 64, (3, 3), (1, 1), (1, 1)), ("maxpool",),
          ("conv", "Conv2d_3b_1x1", 80, (1, 1), (1, 1), (0, 0)), ("conv", "Conv2d_4a_3x3", 192, (3, 3), (1, 1), (0, 0)),
          ("maxpool",),
          ("block", "Mixed_5b", _A(32)), ("block", "Mixed_5c", _A(64)), ("block", "Mixed_5d", _A(64)),
          ("block", "Mixed_6a", _B),
          ("block", "Mixed_6b", _C(128)), ("block", "Mixed_6c", _C(160)), ("block", "Mixed_6d", _C(160)),
          ("block", "Mixed_6e", _C(192)),
          ("block", "Mixed_7a", _D), ("block", "Mixed_7b", _E), ("block", "Mixed_7c", _E)]


class _BasicConv2d(nn.Module):
    """Parameter holder with torchvision's names (conv.weight, bn.{weight, bias, running_mean, running_var, ...})."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=p, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)


class _Holder(nn.Module):
    pass


def _out_channels(spec, cin):
    couts = {c[0]: c[1] for c in spec["convs"]}
    return sum(couts[s] if s in couts else cin for s in spec["out"])


class InceptionV3(nn.Module):
    """torchvision ``Inception3(aux_logits=True, transform_input=True)`` restricted to what src/fid.py evaluates."""

    def __init__(self):
        super().__init__()
        cin = 3
        for item in _TRUNK:
            if item[0] == "conv":
                _, name, cout, k, s, p = item
                setattr(self, name, _BasicConv2d(cin, cout, k, s, p))
                cin = cout
            elif item[0] == "block":
                _, name, spec = item
                blk = _Holder()
                couts = {}
                for (cname, cout, k, s, p) in spec["convs"]:
                    src = spec["wire"][cname]
                    c_in = cin if src in ("x", "avgpool(x)") else couts[src]
                    setattr(blk, cname, _BasicConv2d(c_in, cout, k, s, p))
                    couts[cname] = cout
                setattr(self, name, blk)
                cin = _out_channels(spec, cin)
                if name == "Mixed_6e":          # torchvision registers AuxLogits here (module order = state_dict key order)
                    aux = _Holder()
                    aux.conv0 = _BasicConv2d(768, 128, (1, 1), (1, 1), (0, 0))
                    aux.conv1 = _BasicConv2d(128, 768, (5, 5), (1, 1), (0, 0))
                    aux.fc = nn.Linear(768, 1000)
                    self.AuxLogits = aux
        assert cin == 2048
        self.fc = nn.Linear(2048, 1000)
        self._folded: Dict[str, Tuple] = {}
        self.eval()

    # ---------------------------------------------------------------- runtime
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._folded = {}
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._folded = {}
        return r

    def _fold(self, key: str, bc: _BasicConv2d):
        """GEMM operands of a BasicConv2d: weight as [Cout][(kh, kw, Cin)] fp32 and the eval-mode BatchNorm as scale / shift."""
        f = self._folded.get(key)
        if f is None:
            with torch.no_grad():
                w = bc.conv.weight.detach().float()
                wr = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
                scale = (bc.bn.weight / torch.sqrt(bc.bn.running_var + bc.bn.eps)).float().contiguous()
                shift = (bc.bn.bias - bc.bn.running_mean * scale).float().contiguous()
            f = self._folded[key] = (wr, scale, shift)
        return f

    @torch.no_grad()
    def features(self, x01: torch.Tensor) -> torch.Tensor:
        """(N, 3, 299, 299) float in [0, 1] on the GPU -> (N, 2048) pool features (src/fid.py:47-65)."""
        if self.training:
            raise RuntimeError("InceptionV3.features: eval mode only (the reference evaluates the network in eval mode)")
        if x01.dim() != 4 or tuple(x01.shape[1:]) != (3, 299, 299):
            raise ValueError("Expected input shape to be: (N,3,299,299), but got {}".format(tuple(x01.shape)))
        dev = self.fc.weight.device
        if dev.type != "cuda":
            raise RuntimeError("InceptionV3.features runs on the HIP kernels only (move the module to a ROCm GPU)")
        lib = _abi.load()
        stream = torch.cuda.current_stream(dev).cuda_stream
        x01 = x01.to(dev).float().contiguous()
        N = x01.shape[0]
        # x * 2 - 1 (src/fid.py:55), then torchvision's transform_input: ch * (std_c / 0.5) + (mean_c - 0.5) / 0.5
        std = torch.tensor([0.229, 0.224, 0.225], device=dev) / 0.5
        mean = (torch.tensor([0.485, 0.456, 0.406], device=dev) - 0.5) / 0.5
        scale, shift = (2.0 * std).contiguous(), (mean - std).contiguous()
        x = torch.empty((N, 299, 299, 3), dtype=torch.float32, device=dev)
        check(lib.rg_nchw_to_nhwc_affine(x01.data_ptr(), x.data_ptr(), N, 3, 299, 299, scale.data_ptr(), shift.data_ptr(),
                                         stream), "rg_nchw_to_nhwc_affine")
        scratch = {"buf": None}

        def cols_buf(nfloat):
            if scratch["buf"] is None or scratch["buf"].numel() < nfloat:
                scratch["buf"] = torch.empty(nfloat, dtype=torch.float32, device=dev)
            return scratch["buf"]

        def conv(key, bc, src, c0, cin, dst, d0):
            """src: [N, H, W, Csrc] NHWC tensor, channels c0 .. c0+cin-1 are the input; writes channels d0.. of dst."""
            _, H, W, Csrc = src.shape
            kh, kw = bc.conv.kernel_size
            sh, sw = bc.conv.stride
            ph, pw = bc.conv.padding
            Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
            assert dst.shape[1] == Ho and dst.shape[2] == Wo
            wr, sc, shf = self._fold(key, bc)
            cout, K = wr.shape
            M = N * Ho * Wo
            a_ptr, lda = src.data_ptr() + 4 * c0, Csrc
            if (kh, kw, sh, sw, ph, pw) != (1, 1, 1, 1, 0, 0):
                cols = cols_buf(M * K)
                check(lib.rg_im2col_nhwc(a_ptr, Csrc, cols.data_ptr(), N, H, W, cin, kh, kw, sh, sw, ph, pw, stream),
                      "rg_im2col_nhwc")
                a_ptr, lda = cols.data_ptr(), K
            check(lib.rg_linear_affine_act(a_ptr, lda, wr.data_ptr(), 0, sc.data_ptr(), shf.data_ptr(),
                                           dst.data_ptr() + 4 * d0, dst.shape[3], M, K, cout, 0.0, _abi.ALGO_GENERIC, 0, 0,
                                           stream), "rg_linear_affine_act")

        def pool(src, k, s, p, mode, dst, d0):
            _, H, W, C = src.shape
            check(lib.rg_pool2d_nhwc(src.data_ptr(), C, dst.data_ptr() + 4 * d0, dst.shape[3], N, H, W, C, k, s, p, mode,
                                     stream), "rg_pool2d_nhwc")

        def new(h, w, c):
            return torch.empty((N, h, w, c), dtype=torch.float32, device=dev)

        for item in _TRUNK:
            _, H, W, C = x.shape
            if item[0] == "conv":
                _, name, cout, k, s, p = item
                bc = getattr(self, name)
                y = new((H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1, cout)
                conv(name, bc, x, 0, C, y, 0)
                x = y
            elif item[0] == "maxpool":
                y = new((H - 3) // 2 + 1, (W - 3) // 2 + 1, C)
                pool(x, 3, 2, 0, 0, y, 0)
                x = y
            else:
                _, name, spec = item
                blk = getattr(self, name)
                specs = {c[0]: c for c in spec["convs"]}
                strided = any(specs[s][3] != (1, 1) for s in spec["out"] if s in specs)
                Ho, Wo = ((H - 3) // 2 + 1, (W - 3) // 2 + 1) if strided else (H, W)
                out = new(Ho, Wo, _out_channels(spec, C))
                offs, o = {}, 0
                for s in spec["out"]:
                    offs[s] = o
                    o += specs[s][1] if s in specs else C
                bufs = {"x": x}
                for (cname, cout, k, s, p) in spec["convs"]:
                    srcname = spec["wire"][cname]
                    if srcname == "avgpool(x)":
                        if "avgpool(x)" not in bufs:
                            bufs["avgpool(x)"] = new(H, W, C)
                            pool(x, 3, 1, 1, 1, bufs["avgpool(x)"], 0)
                    src = bufs[srcname]
                    hi, wi = src.shape[1], src.shape[2]
                    ho, wo = (hi + 2 * p[0] - k[0]) // s[0] + 1, (wi + 2 * p[1] - k[1]) // s[1] + 1
                    if cname in offs:
                        conv(name + "." + cname, getattr(blk, cname), src, 0, src.shape[3], out, offs[cname])
                    else:
                        bufs[cname] = new(ho, wo, cout)
                        conv(name + "." + cname, getattr(blk, cname), src, 0, src.shape[3], bufs[cname], 0)
                if "maxpool(x)" in offs:
                    pool(x, 3, 2, 0, 0, out, offs["maxpool(x)"])
                x = out
        assert tuple(x.shape[1:]) == (8, 8, 2048)
        feats = torch.empty((N, 2048), dtype=torch.float32, device=dev)
        check(lib.rg_spatial_mean_nhwc(x.data_ptr(), feats.data_ptr(), N, 64, 2048, stream), "rg_spatial_mean_nhwc")
        return feats

    def forward(self, x01):
        return self.features(x01)


def manifest() -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every state_dict entry, in torchvision's order: what a checkpoint must provide."""
    with torch.device("meta"):
        m = InceptionV3()
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
