"""Seeded slide databases and slide table shared by tests/golden/make_data_fixtures.py and tests/test_data_cpu.py."""
import numpy as np
import pandas as pd


def make_slides():
    rng = np.random.default_rng(77)
    return {"GTEX-AAA-0001.svs": [rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8) for _ in range(9)],
            "GTEX-BBB-0002.svs": [rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8) for _ in range(3)],
            "GTEX-CCC-0003.svs": [rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8) for _ in range(6)]}


def make_table(slides):
    rng = np.random.default_rng(78)
    rna = np.round(rng.gamma(1.5, 20.0, size=(len(slides), 7)), 3)
    rna[0, 2] = 0.0                      # zeros stay 0 after the log
    rna[:, 5] = 4.0                      # a constant gene: StandardScaler divides by 1
    cols = {"wsi_file_name": slides, "tissue": ["lung"] * len(slides)}
    for j in range(rna.shape[1]):
        cols["rna_G%d" % j] = rna[:, j]
    return pd.DataFrame(cols)
