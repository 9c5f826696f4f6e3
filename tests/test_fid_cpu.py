"""Frechet distance: product math vs the oracle restatement of src/fid.py:95-163 and vs closed forms."""
import numpy as np

from oracle import ref_cpu as R
from rna_gan_amd import fid


def test_frechet_distance_closed_forms():
    rng = np.random.default_rng(0)
    d = 6
    mu1, mu2 = rng.standard_normal(d), rng.standard_normal(d)
    c1, c2 = rng.uniform(0.5, 2.0, d), rng.uniform(0.5, 2.0, d)
    # diagonal (commuting) covariances: Tr(C1 + C2 - 2 sqrt(C1 C2)) = sum (sqrt(c1) - sqrt(c2))^2
    want = np.sum((mu1 - mu2) ** 2) + np.sum((np.sqrt(c1) - np.sqrt(c2)) ** 2)
    got = fid.frechet_distance(mu1, np.diag(c1), mu2, np.diag(c2))
    np.testing.assert_allclose(got, want, rtol=1e-9)
    assert abs(fid.frechet_distance(mu1, np.diag(c1), mu1, np.diag(c1))) < 1e-9


def test_frechet_distance_matches_oracle_on_sample_statistics():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((500, 12)) @ rng.standard_normal((12, 12))
    b = 0.3 + rng.standard_normal((400, 12)) @ rng.standard_normal((12, 12))
    m1, s1 = fid.activation_statistics(a)
    m2, s2 = fid.activation_statistics(b)
    np.testing.assert_allclose(m1, a.mean(0)); np.testing.assert_allclose(s1, np.cov(a, rowvar=False))
    np.testing.assert_allclose(fid.frechet_distance(m1, s1, m2, s2), R.frechet_distance(m1, s1, m2, s2), rtol=1e-10)
    # singular product (rank-deficient features): both go through the eps-regularised branch or agree anyway
    c = np.concatenate([a[:, :6], a[:, :6]], axis=1)
    mc, sc = fid.activation_statistics(c)
    np.testing.assert_allclose(fid.frechet_distance(mc, sc, m2, s2), R.frechet_distance(mc, sc, m2, s2), rtol=1e-6)


def test_frechet_distance_reference_fixture(golden_dir):
    """f8_frechet.npz holds what the REFERENCE's calculate_frechet_distance (src/fid.py:112-163, imported by
    tests/golden/make_fid_genimg_fixtures.py) returned for the seeded statistics of tests/golden/fid_cases.py,
    including a rank-deficient pair, fewer samples than features and the eps-regularised branch (forced): the oracle
    restatement and the product's frechet_distance must both reproduce it."""
    import os
    import sys
    import warnings
    sys.path.insert(0, golden_dir)
    from fid_cases import stats_cases
    from scipy import linalg
    fx = np.load(os.path.join(golden_dir, "f8_frechet.npz"))
    for name, (m1, s1, m2, s2) in stats_cases().items():
        want = float(fx[name])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for fn in (R.frechet_distance, fid.frechet_distance):
                got = fn(m1, s1, m2, s2)
                assert abs(got - want) <= 1e-7 * max(abs(want), 1.0) + (1e-6 if name == "same" else 0.0), (name, fn, got, want)
    # eps branch: sqrtm yields non-finite values on the first call -> offset eps * I on both covariances
    m1, s1, m2, s2 = stats_cases()["generic"]
    real = linalg.sqrtm
    for fn in (R.frechet_distance, fid.frechet_distance):
        calls = []

        def flaky(mat, disp=True):
            calls.append(1)
            if len(calls) == 1:
                bad = np.full_like(np.asarray(mat, dtype=np.float64), np.nan)
                return (bad, 0.0) if not disp else bad
            return real(mat, disp=disp) if not disp else real(mat)
        linalg.sqrtm = flaky
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = fn(m1, s1, m2, s2, eps=1e-6)
        finally:
            linalg.sqrtm = real
        assert len(calls) == 2 and abs(got - float(fx["generic.forced_eps"])) <= 1e-7 * abs(got), (fn, got)


def test_preprocessing_and_protocol():
    """src/fid.py:166-232,:312-330 around a pluggable feature extractor: 299x299 float CHW in [0,1] from uint8 / float
    NHWC, identity for 299-pixel inputs, and the 5-repetition mean +- std report."""
    import torch
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, size=(5, 64, 64, 3), dtype=np.uint8)
    x = fid.preprocess_images(u8)
    assert x.shape == (5, 3, 299, 299) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    same = rng.random((299, 299, 3)).astype(np.float32)
    np.testing.assert_allclose(fid.preprocess_image(same).permute(1, 2, 0).numpy(), same, atol=1e-6)
    # a constant image stays constant; bilinear interpolation preserves the mean of a linear ramp
    ramp = np.tile(np.linspace(0, 1, 32, dtype=np.float32)[None, :, None], (32, 1, 3))
    r = fid.preprocess_image(ramp)
    assert abs(float(r.mean()) - 0.5) < 1e-3 and float((r[:, :, 1:] - r[:, :, :-1]).min()) >= -1e-6
    feat = lambda t: torch.nn.functional.adaptive_avg_pool2d(t, (4, 4)).reshape(t.shape[0], -1).numpy()   # noqa: E731
    real = rng.random((40, 32, 32, 3)).astype(np.float32)
    k = [0]

    def gen():
        k[0] += 1
        return np.clip(real + 0.05 * k[0] * rng.standard_normal(real.shape).astype(np.float32), 0, 1)
    rep = fid.fid_protocol(gen, real, feat, iterations=5, batch_size=8)
    assert len(rep["fid_values"]) == 5 and k[0] == 5
    assert abs(rep["mean"] - np.mean(rep["fid_values"])) < 1e-12 and abs(rep["std"] - np.std(rep["fid_values"])) < 1e-12
    assert rep["fid_values"][4] > rep["fid_values"][0] > 0
