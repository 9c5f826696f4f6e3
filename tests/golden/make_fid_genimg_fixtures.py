#!/usr/bin/env python3
"""Golden fixtures F8 / F9 made by IMPORTING THE REFERENCE's src/fid.py and src/gan_utils.py (build container only).

Those two files import packages that are absent here (cv2, torchvision, lmdb, lz4framed, torchgan.trainer, the
reference's own read_data which needs the former).  None of them is touched by the functions executed below except
``transforms.Normalize`` (gan_utils.generate_images, src/gan_utils.py:236-241), so they are satisfied with EMPTY module
stubs in sys.modules, plus a 4-line Normalize that applies torchvision's documented definition
``(x - mean[c]) / std[c]`` per channel.  What runs from the reference:
  * fid.calculate_frechet_distance (src/fid.py:112-163) on seeded activation statistics, including a rank-deficient
    pair (singular covariance product) and the eps-regularised branch forced through non-finite sqrtm output;
  * gan_utils.generate_images (src/gan_utils.py:197-244), conditioned branch (betaVAE latent + uniform noise, chunks
    of 10, un-normalise, NHWC) and unconditioned branch, with a 3-attribute stand-in for the torchgan Trainer
    (generator / device) around the oracle's generator module (a 256-pixel generator: the function hard-codes
    view(-1, 3, 256, 256)) and the REFERENCE's betaVAE class.

    python tests/golden/make_fid_genimg_fixtures.py        ->  f8_frechet.npz, f9_generate_images.npz
"""
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "_torchgan_shim"))
sys.path.insert(0, REF)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Normalize(nn.Module):                      # torchvision.transforms.Normalize's definition, (x - mean) / std
    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = torch.tensor(mean), torch.tensor(std)

    def forward(self, x):
        return (x - self.mean.view(-1, 1, 1)) / self.std.view(-1, 1, 1)


for _n in ("cv2", "lmdb", "lz4framed", "read_data"):
    _stub(_n)
_tv = _stub("torchvision")
_tv.transforms = _stub("torchvision.transforms", Normalize=_Normalize, ConvertImageDtype=lambda *a, **k: nn.Identity())
_tv.models = _stub("torchvision.models", inception_v3=None)
import torchgan  # noqa: E402  (constructor-only shim)
torchgan.trainer = _stub("torchgan.trainer", Trainer=object, ParallelTrainer=object)

from oracle import ref_cpu as R  # noqa: E402
import betaVAE as ref_betavae  # noqa: E402   (reference)
import gan_utils as ref_gan_utils  # noqa: E402  (reference)
import fid as ref_fid  # noqa: E402          (reference)


from fid_cases import stats_cases  # noqa: E402   (tests/golden/fid_cases.py)


def f8():
    out = {}
    for name, (m1, s1, m2, s2) in stats_cases().items():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out[name] = np.float64(ref_fid.calculate_frechet_distance(m1, s1, m2, s2))
            out[name + ".eps_branch"] = np.float64(any("singular product" in str(x.message) for x in w))
    # force the eps branch: sqrtm returning non-finite values for the first call only
    real_sqrtm = ref_fid.linalg.sqrtm
    calls = []

    def flaky(mat, disp=True):
        calls.append(1)
        if len(calls) == 1:
            bad = np.full_like(np.asarray(mat, dtype=np.float64), np.nan)
            return (bad, 0.0) if not disp else bad
        return real_sqrtm(mat, disp=disp) if not disp else real_sqrtm(mat)
    m1, s1, m2, s2 = stats_cases()["generic"]
    ref_fid.linalg.sqrtm = flaky
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out["generic.forced_eps"] = np.float64(ref_fid.calculate_frechet_distance(m1, s1, m2, s2, eps=1e-6))
    finally:
        ref_fid.linalg.sqrtm = real_sqrtm
    np.savez_compressed(os.path.join(HERE, "f8_frechet.npz"), **out)
    print("wrote f8_frechet.npz", {k: float(v) for k, v in out.items()})


def f9():
    E_, F = 16, 40
    G = R.seeded_fill_(R.OracleDCGANGenerator(E_, 256, 3, 1, nonlinearity=nn.LeakyReLU(0.2), last_nonlinearity=nn.Tanh()), 81)
    G.train()                                                          # a freshly loaded trainer's mode
    bv = ref_betavae.betaVAE(F, E_, [32, 24, E_], [24, 32], beta=0.0005)
    R.seeded_fill_(bv, 82)
    bv.eval()
    tr = types.SimpleNamespace(generator=G, device=torch.device("cpu"))
    rna = R.synthetic_rna(1, F, seed=83, distinct=1)
    torch.manual_seed(5)
    cond = ref_gan_utils.generate_images(tr, gene_exp=rna, sample_size=13, betavae=bv)
    bufs_after_cond = {k: v.clone() for k, v in G.state_dict().items() if "running" in k}
    torch.manual_seed(6)
    unc = ref_gan_utils.generate_images(tr, sample_size=20)
    assert cond.shape == (13, 256, 256, 3) and unc.shape == (20, 256, 256, 3), (cond.shape, unc.shape)
    out = {"cond.sub": cond[:, ::16, ::16, :], "cond.sum": np.float64(cond.astype(np.float64).sum()),
           "cond.sumsq": np.float64((cond.astype(np.float64) ** 2).sum()), "cond.min": np.float64(cond.min()),
           "cond.max": np.float64(cond.max()),
           "unc.sub": unc[:, ::16, ::16, :], "unc.sum": np.float64(unc.astype(np.float64).sum()),
           "unc.sumsq": np.float64((unc.astype(np.float64) ** 2).sum())}
    for k, v in bufs_after_cond.items():
        out["bn_after_cond." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "f9_generate_images.npz"), **out)
    print("wrote f9_generate_images.npz", cond.shape, unc.shape)


if __name__ == "__main__":
    torch.set_num_threads(8)
    f8()
    f9()
