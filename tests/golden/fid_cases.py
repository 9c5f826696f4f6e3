"""Seeded activation-statistics cases shared by tests/golden/make_fid_genimg_fixtures.py (which stores what the
reference's calculate_frechet_distance returns for them in f8_frechet.npz) and the tests that replay them."""
import numpy as np


def stats_cases():
    rng = np.random.default_rng(2024)
    d = 24
    a = rng.standard_normal((300, d)) @ rng.standard_normal((d, d))
    b = 0.25 + rng.standard_normal((260, d)) @ rng.standard_normal((d, d)) * 0.8
    c = np.concatenate([a[:, :d // 2], a[:, :d // 2]], axis=1)                 # rank-deficient features
    few = rng.standard_normal((7, d))                                         # fewer samples than features
    cases = {"generic": (a, b), "same": (a, a.copy()), "rank_deficient": (c, b), "few_samples": (few, b[:9])}
    return {k: (np.mean(x, 0), np.cov(x, rowvar=False), np.mean(y, 0), np.cov(y, rowvar=False)) for k, (x, y) in cases.items()}
