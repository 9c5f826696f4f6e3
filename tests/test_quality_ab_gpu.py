"""Reduced-size run of tools/train_quality_ab.py (the training-quality A/B behind the north star's FID clause, BASELINE.md 4;
src/fid.py:98-163): the bf16 MFMA mode and the fp32 mode (the reference's arithmetic, pinned to the CPU oracle) train from the
same seeds on the same data order; a second fp32 run from another seed gives the run-to-run spread.  Frechet distances
(src/fid.py:112-163's formula) on the trunk features of the fp32 run's discriminator.

Stated bounds (the full-size table is in DESIGN 13):
  * FD(bf16, fp32 same seed) <= 3 x FD(fp32 other seed, fp32)  -- switching the arithmetic moves the trained generator no
    further than re-seeding the reference arithmetic does (trajectories diverge either way);
  * FD(bf16, held-out real) <= 1.5 x the larger of the two fp32 runs' distances to the held-out real tiles -- the bf16 mode does
    not train a worse generator."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

FACTOR_BETWEEN, FACTOR_REAL = 3.0, 1.5


@pytest.mark.parametrize("loss_type,enc", [("wgan", 128), ("wganvae", 2048)])
def test_bf16_training_quality_matches_fp32_within_seed_spread(loss_type, enc):
    import train_quality_ab as T
    rec = T.run_ab(loss_type, size=32, step=32, enc=enc, iters=200, batch=32, n_eval=768, rna_features=128, n_slides=32,
                   tiles_per_slide=32, inception=False, log=lambda *a: print(*a))
    row = rec["frechet"]["d_trunk(A)"]
    print(rec["loss_curves_windowed_means[g,d,gp]"])
    print(row)
    vals = [row[k] for k in ("FD(B_bf16, A_fp32)", "FD(C_fp32_seed2, A_fp32)", "FD(A, real)", "FD(B, real)", "FD(C, real)")]
    assert all(np.isfinite(v) and v >= -1e-6 for v in vals), row
    assert row["FD(B_bf16, A_fp32)"] <= FACTOR_BETWEEN * row["FD(C_fp32_seed2, A_fp32)"], row
    assert row["FD(B, real)"] <= FACTOR_REAL * max(row["FD(A, real)"], row["FD(C, real)"]), row
    # training did something: every run's generator is closer to the held-out real tiles than an untrained one would be is
    # not asserted (no fourth run); the curves must at least be finite everywhere
    for tag, w in rec["loss_curves_windowed_means[g,d,gp]"].items():
        assert np.isfinite(np.asarray(w)).all(), tag
