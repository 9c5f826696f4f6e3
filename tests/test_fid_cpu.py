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
