"""f2 on the GPU: the HIP Inception-v3 feature extractor (rna_gan_amd/inception.py: im2col + GEMM with folded BatchNorm +
ReLU epilogue, pooling, channel-slice outputs) against the oracle restatement (oracle/inception_ref.py, plain PyTorch on
the CPU) with seeded random weights, and through fid.calculate_fid."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.inception_ref import OracleInception3
from oracle import ref_cpu as R
from rna_gan_amd import fid as PF
from rna_gan_amd import inception as PI


def _seeded_oracle(seed):
    torch.manual_seed(seed)
    o = OracleInception3().eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in list(o.named_parameters()) + list(o.named_buffers()):
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            elif name.endswith("running_mean"):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif name.endswith("bn.weight"):
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            elif name.endswith("bias"):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            else:                                   # conv / linear weights: variance-preserving scale (ReLU network)
                fan_in = int(np.prod(t.shape[1:]))
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5)
    return o


def test_features_match_oracle():
    o = _seeded_oracle(11)
    p = PI.InceptionV3()
    p.load_state_dict(o.state_dict())
    p = p.cuda().eval()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 3, 299, 299, generator=g)
    with torch.no_grad():
        want = o.features(x)
    got = p.features(x.cuda()).cpu()
    assert got.shape == (3, 2048)
    err = float((got - want).abs().max() / (want.abs().max() + 1e-12))
    print("inception features: rel-max error vs oracle %.2e, feature scale %.3f" % (err, float(want.abs().mean())))
    assert err < 2e-4                     # fp32 GEMMs with different summation orders through 47 layers
    assert float(want.std()) > 1e-3       # (a live signal reaches the end: the comparison is not 0 == 0)


def test_calculate_fid_with_the_hip_extractor():
    o = _seeded_oracle(12)
    extract = PF.inception_feature_extractor(o.state_dict())
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, size=(6, 64, 64, 3), dtype=np.uint8)
    b = rng.integers(0, 200, size=(6, 64, 64, 3), dtype=np.uint8)
    d_ab = PF.calculate_fid(a, b, extract, batch_size=3)
    d_aa = PF.calculate_fid(a, a, extract, batch_size=3)
    # the same numbers from the oracle network through the oracle's distance
    def feats(imgs):
        x = PF.preprocess_images(imgs)
        with torch.no_grad():
            return o.features(x).double().numpy()
    fa, fb = feats(a), feats(b)
    want = R.frechet_distance(fa.mean(0), np.cov(fa, rowvar=False), fb.mean(0), np.cov(fb, rowvar=False))
    assert abs(d_aa) < 1e-3 * max(1.0, abs(d_ab))
    assert abs(d_ab - want) <= 2e-3 * abs(want) + 1e-6, (d_ab, want)
