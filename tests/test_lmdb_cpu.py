"""f3: slide databases are LMDB files (src/preprocess/patch_gen_grid.py:92-133 writes them,
src/read_data.py:284-342 reads them).  rna_gan_amd.lmdb_ro reads such a file without the lmdb package; here against files
from tests/lmdb_writer.py (an independent writer of the published format -- no liblmdb exists in the build container, so
parity with liblmdb-written files is unpinned)."""
import os
import pickle
import random
import struct

import numpy as np
import pytest

from lmdb_writer import write_lmdb
from rna_gan_amd import data as PD
from rna_gan_amd.lmdb_ro import LmdbFormatError, LmdbReadOnly


def _rand_bytes(rng, n):
    return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()


@pytest.mark.parametrize("psize", [4096, 16384])
def test_get_len_iter_multi_level_tree(tmp_path, psize):
    rng = np.random.default_rng(7)
    items = {}
    for i in range(700):                      # small values: inline leaf nodes, several leaf pages, >= 1 branch level
        items[str(i).encode("ascii")] = _rand_bytes(rng, int(rng.integers(0, 120)))
    for i in range(700, 760):                 # large values: overflow page runs (a 256x256x3 tile record is ~100-200 KB)
        items[str(i).encode("ascii")] = _rand_bytes(rng, int(rng.integers(psize, 5 * psize)))
    items[b"__keys__"] = _rand_bytes(rng, 3000)
    items[b""] = b"empty-key"                 # shortest possible key sorts first
    path = str(tmp_path / "slide.db")
    info = write_lmdb(path, items, psize=psize)
    assert info["depth"] >= 2 and info["overflow_pages"] > 0
    db = LmdbReadOnly(path)
    assert len(db) == len(items) and db.stat()["psize"] == psize and db.stat()["depth"] == info["depth"]
    for k, v in items.items():
        assert db[k] == v, k
    assert list(db) == sorted(items)                              # key order = memcmp, then length
    assert dict(db.items()) == items
    for missing in (b"760", b"__keys", b"__keys__x", b"\xff", b"00"):
        with pytest.raises(KeyError):
            db[missing]
    assert b"12" in db and b"nope" not in db
    db.close()


def test_three_level_tree_and_newer_meta(tmp_path):
    items = {("%05d" % i).encode(): struct.pack("<I", i) * 3 for i in range(60000)}
    path = str(tmp_path / "big.db")
    info = write_lmdb(path, items, psize=4096)
    assert info["depth"] >= 3
    db = LmdbReadOnly(path)
    for i in random.Random(3).sample(range(60000), 500):
        assert db[("%05d" % i).encode()] == struct.pack("<I", i) * 3
    assert len(db) == 60000
    # meta page 0 describes an older, empty transaction: the reader must have taken meta 1
    write_lmdb(path, {b"a": b"1"}, stale_first_meta=False)
    assert LmdbReadOnly(path)[b"a"] == b"1"


def test_empty_and_malformed(tmp_path):
    p = str(tmp_path / "empty.db")
    write_lmdb(p, {})
    db = LmdbReadOnly(p)
    assert len(db) == 0 and list(db) == []
    with pytest.raises(KeyError):
        db[b"0"]
    bad = str(tmp_path / "bad.db")
    with open(bad, "wb") as f:
        f.write(b"\0" * 8192)
    with pytest.raises(LmdbFormatError):
        LmdbReadOnly(bad)
    trunc = str(tmp_path / "trunc.db")
    write_lmdb(trunc, {str(i).encode(): b"x" * 5000 for i in range(10)})
    with open(trunc, "r+b") as f:
        f.truncate(3 * 4096)
    with pytest.raises(LmdbFormatError):
        LmdbReadOnly(trunc)[b"9"]


def test_dataset_reads_lmdb_slide_databases(tmp_path):
    """The reference's layout end to end: <patch_data_path>/<wsi>/<wsi with .svs -> .db> is an LMDB FILE holding
    lz4framed(pickle((name, bytes, shape))) records under b"0".. and the key list under b"__keys__"
    (src/preprocess/patch_gen_grid.py:92-133); PatchRNADataset samples and decodes tiles from it (src/read_data.py:284-342)."""
    import pandas as pd
    rng = np.random.default_rng(11)
    root = tmp_path / "patches"
    rows, tiles_by_wsi = [], {}
    for s in range(3):
        wsi = "GTEX-%d.svs" % s
        os.makedirs(root / wsi)
        n = 5 + s
        tiles = rng.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8)
        items = {str(i).encode("ascii"): PD.encode_record("%s_patch_%d" % (wsi[:-4], i), tiles[i]) for i in range(n)}
        items[b"__keys__"] = PD.encode_keys(n)
        write_lmdb(str(root / wsi / wsi.replace(".svs", ".db")), items)
        tiles_by_wsi[wsi] = tiles
        rows.append({"wsi_file_name": wsi, "rna_a": float(s), "rna_b": 2.0 * s, "patch_data_path": str(root), "labels": 0})
    table = pd.DataFrame(rows)
    random.seed(5)
    ds = PD.PatchRNADataset(str(root), table, 32, max_patches_total=4)
    assert len(ds) == 12
    for idx in range(len(ds)):
        item = ds[idx]
        wsi, i = ds.filenames[idx], ds.images[idx]
        want = tiles_by_wsi[wsi][i][:, :, ::-1]                      # stored BGR -> RGB (src/read_data.py:339)
        assert np.array_equal(item["image"].permute(1, 2, 0).numpy(), want)
        assert float(item["rna_data"][0]) == float(wsi.split("-")[1][0])
    # a key that is missing from the database gives image None (the reference's decompress_and_deserialize returns None
    # on any failure and collate_fn drops the record), not an exception
    ds.keys[0] = b"999"
    assert ds[0]["image"] is None
