"""Input side on the CPU: LZ4 frame codec against hand-assembled frames of the published format, the tile
record codec, and the dataset sampling / decoding / RNA preparation against fixture F10 (tests/golden/f10_data.npz: the
REFERENCE's src/read_data.py datasets and the pandas + scikit-learn call sequence of src/histopathology_gan.py:131-151,
run by tests/golden/make_data_fixtures.py)."""
import os
import random
import struct
import sys

import numpy as np
import pytest
import torch

from rna_gan_amd import data as PD


def test_xxh32_known_answers():
    assert PD._xxh32(b"") == 0x02CC5D05
    assert PD._xxh32(b"abc") == 0x32D153FF
    assert PD._xxh32(b"Nobody inspects the spammish repetition") == 0xE2293B2F


def _frame(blocks, flg=0x60, bd=0x40, content_size=None, end=True):
    desc = bytes([flg, bd]) + (struct.pack("<Q", content_size) if content_size is not None else b"")
    out = struct.pack("<I", 0x184D2204) + desc + bytes([(PD._xxh32(desc) >> 8) & 0xFF])
    for stored, payload in blocks:
        out += struct.pack("<I", len(payload) | (0x80000000 if stored else 0)) + payload
    return out + (struct.pack("<I", 0) if end else b"")


def test_lz4_frame_decoder_on_hand_assembled_frames():
    # stored block
    assert PD.lz4f_decompress(_frame([(True, b"hello world")])) == b"hello world"
    # one compressed block: 4 literals "abcd", match offset 4 length 8 (overlap-free copy twice), then literals "XYZ12"
    #   token 0x44: 4 literals, match length 4 + 4 = 8 ; offset 0x0004 ; last sequence token 0x50: 5 literals
    blk = bytes([0x44]) + b"abcd" + bytes([0x04, 0x00]) + bytes([0x50]) + b"XYZ12"
    assert PD.lz4f_decompress(_frame([(False, blk)])) == b"abcd" + b"abcdabcd" + b"XYZ12"
    # overlapping match (run-length): 1 literal "z", match offset 1 length 4 + 15 + 3 = 22 -> 23 x "z", then 5 literals
    blk = bytes([0x1F]) + b"z" + bytes([0x01, 0x00, 0x03]) + bytes([0x50]) + b"12345"
    assert PD.lz4f_decompress(_frame([(False, blk)])) == b"z" * 23 + b"12345"
    # long literal run (15 + 255 + 5 = 275 literals, length bytes 255, 5)
    lit = bytes(range(256)) + bytes(19)
    blk = bytes([0xF0, 255, 5]) + lit
    assert PD.lz4f_decompress(_frame([(False, blk)])) == lit
    # linked blocks (FLG bit 5 clear): the second block copies from the first one's output
    b1 = bytes([0x80]) + b"ABCDEFGH"
    b2 = bytes([0x04]) + bytes([0x08, 0x00]) + bytes([0x50]) + b"vwxyz"       # 0 literals, match offset 8 length 8
    assert PD.lz4f_decompress(_frame([(False, b1), (False, b2)], flg=0x40)) == b"ABCDEFGH" + b"ABCDEFGH" + b"vwxyz"
    # content size field, skippable frame in front, two frames back to back
    f1 = _frame([(True, b"one")], flg=0x68, content_size=3)
    skip = struct.pack("<II", 0x184D2A50, 4) + b"\xde\xad\xbe\xef"
    assert PD.lz4f_decompress(skip + f1 + _frame([(True, b"two")])) == b"onetwo"
    with pytest.raises(ValueError):
        PD.lz4f_decompress(_frame([(True, b"one")], flg=0x68, content_size=4))
    with pytest.raises(ValueError):
        PD.lz4f_decompress(b"\x00\x01\x02\x03\x04\x05\x06\x07")


def test_lz4_round_trip_and_records():
    rng = np.random.default_rng(0)
    cases = [b"", b"a", bytes(rng.integers(0, 256, 100000, dtype=np.uint8)), b"abcabcabc" * 30000,
             bytes(rng.integers(0, 4, 200000, dtype=np.uint8)), bytes(70000)]
    for c in cases:
        f = PD.lz4f_compress(c)
        assert PD.lz4f_decompress(f) == c
    assert len(PD.lz4f_compress(b"abcabcabc" * 30000)) < 5000 and len(PD.lz4f_compress(bytes(70000))) < 1000
    tile = rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8)
    img = PD.decompress_and_deserialize(PD.encode_record("s_patch_0", tile))
    assert img.dtype == torch.uint8 and img.shape == (3, 16, 16)
    assert torch.equal(img, torch.from_numpy(tile[:, :, ::-1].copy()).permute(2, 0, 1))       # BGR -> RGB, HWC -> CHW
    assert PD.decompress_and_deserialize(b"garbage") is None
    import pickle
    assert pickle.loads(PD.lz4f_decompress(PD.encode_keys(3))) == [b"0", b"1", b"2"]


def test_datasets_and_rna_table_reference_fixture(golden_dir, tmp_path):
    sys.path.insert(0, golden_dir)
    from data_cases import make_slides, make_table
    fx = np.load(os.path.join(golden_dir, "f10_data.npz"))
    slides = make_slides()
    table, mean, scale = PD.log_standardize_rna(make_table(list(slides)))
    assert list(table.columns) == list(fx["rna.columns"])
    rna_cols = [c for c in table.columns if "rna_" in c]
    np.testing.assert_allclose(table[rna_cols].to_numpy(dtype=np.float64), fx["rna.values"], rtol=1e-12, atol=1e-12)
    assert scale[5] == 1.0 and float(table["rna_G5"].abs().max()) == 0.0                      # constant gene
    table["patch_data_path"] = ["/data/patches"] * table.shape[0]
    table["labels"] = [0, 1, 0][:table.shape[0]]
    stores = {}
    for wsi, tiles in slides.items():
        st = {str(i).encode(): PD.encode_record("%s_patch_%d" % (wsi, i), t) for i, t in enumerate(tiles)}
        st[b"__keys__"] = PD.encode_keys(len(tiles))
        stores[os.path.join("/data/patches", wsi, wsi.replace(".svs", ".db"))] = st
    for name, cls in (("rna", PD.PatchRNADataset), ("plain", PD.PatchDataset)):
        random.seed(1234)
        ds = cls("/data/patches", table, 16, transforms=lambda im: im.float() / 255.0, max_patches_total=5, stores=stores)
        assert len(ds) == int(fx[name + ".len"])
        assert ds.filenames == list(fx[name + ".filenames"])
        assert [k.decode() for k in ds.keys] == list(fx[name + ".keys"])
        for i in range(len(ds)):
            item = ds[i]
            img = item["image"] if isinstance(item, dict) else item[0]
            lab = item["labels"] if isinstance(item, dict) else item[1]
            np.testing.assert_allclose(float(img.double().sum()), fx[name + ".image_sums"][i], rtol=1e-12)
            np.testing.assert_array_equal(img[:, 0, :4].numpy(), fx[name + ".image_heads"][i])
            assert float(lab) == fx[name + ".labels"][i]
            if isinstance(item, dict):
                np.testing.assert_array_equal(item["rna_data"].numpy(), fx[name + ".rna"][i])
    # the same slide databases through the directory backend, uint8 out (device-side normalisation path)
    root = str(tmp_path / "patches")
    for wsi, tiles in slides.items():
        PD.write_tile_store(os.path.join(root, wsi, wsi.replace(".svs", "")), tiles, slide_id=wsi)
    table["patch_data_path"] = [root] * table.shape[0]
    random.seed(1234)
    ds = PD.PatchRNADataset(root, table, 16, transforms=None, max_patches_total=5)
    assert len(ds) == int(fx["rna.len"]) and [k.decode() for k in ds.keys] == list(fx["rna.keys"])
    it = ds[0]
    assert it["image"].dtype == torch.uint8
    np.testing.assert_allclose(float(PD.ToFloatNormalize(0.0, 1.0)(it["image"]).double().sum()), fx["rna.image_sums"][0], rtol=1e-6)


def test_mixed_tissue_tables(tmp_path):
    """BASELINE configs[3]'s data side (src/histopathology_gan.py:111-151): two tissue tables -> one table with the CSV index
    as tissue id and each row's own tile directory, RNA columns ln'd (zeros stay 0) and standardised over the UNION of the
    tissues, per-slide tile sampling from each tissue's store; the items carry the tissue id."""
    import pandas as pd
    rng = np.random.default_rng(5)
    genes = ["rna_G%d" % i for i in range(6)]
    paths, roots, slides = [], [], {}
    for t, (tissue, n) in enumerate((("lung", 3), ("brain", 2))):
        names = ["%s_%d.svs" % (tissue, i) for i in range(n)]
        x = rng.gamma(2.0, 3.0 + 4.0 * t, size=(n, len(genes)))
        x[0, 1] = 0.0                                      # an unexpressed gene: ln(0) -> 0
        df = pd.DataFrame(x, columns=genes)
        df.insert(0, "wsi_file_name", names)
        df["tissue"] = tissue
        p = str(tmp_path / ("%s.csv" % tissue))
        df.to_csv(p, index=False)
        root = str(tmp_path / ("patches_" + tissue))
        for wsi in names:
            tiles = [rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8) for _ in range(4)]
            PD.write_tile_store(os.path.join(root, wsi, wsi.replace(".svs", "")), tiles, slide_id=wsi)
            slides[wsi] = t
        paths.append(p); roots.append(root)
    table = PD.load_slide_tables(paths, roots)
    assert table.shape[0] == 5 and list(table["labels"]) == [0, 0, 0, 1, 1]
    assert list(table["patch_data_path"]) == [roots[0]] * 3 + [roots[1]] * 2
    raw = table[genes].to_numpy(dtype=np.float64)
    prepared, mean, scale = PD.log_standardize_rna(table)
    lx = np.where(raw == 0, 0.0, np.log(np.where(raw == 0, 1.0, raw)))
    np.testing.assert_allclose(mean, lx.mean(0), rtol=1e-12)
    np.testing.assert_allclose(prepared[genes].to_numpy(dtype=np.float64), (lx - lx.mean(0)) / lx.std(0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(prepared[genes].to_numpy(dtype=np.float64).mean(0), 0.0, atol=1e-12)   # over the union
    assert abs(float(prepared[genes].to_numpy(dtype=np.float64)[:3].mean())) > 1e-3                   # not per tissue
    random.seed(7)
    ds = PD.PatchRNADataset(roots, prepared, 16, transforms=PD.ToFloatNormalize(0.5, 0.5), max_patches_total=3)
    assert len(ds) == 15
    for i in range(len(ds)):
        it = ds[i]
        wsi = ds.filenames[i] if isinstance(ds.filenames[i], str) else ds.filenames[i].decode()
        key = [k for k in slides if wsi.startswith(k.replace(".svs", ""))]
        assert len(key) == 1 and float(it["labels"]) == float(slides[key[0]])
        assert it["image"].shape == (3, 16, 16) and it["rna_data"].shape == (6,)
        row = prepared[prepared["wsi_file_name"] == key[0]][genes].to_numpy(dtype=np.float32)[0]
        np.testing.assert_array_equal(it["rna_data"].numpy(), row)
    # a single table passed as a string (configs/gan_run_lung.json's form)
    one = PD.load_slide_tables(paths[0], roots[0])
    assert one.shape[0] == 3 and set(one["labels"]) == {0}


def test_rank_shards_have_equal_batch_counts():
    """Multi-rank CLI runs: every rank must see the same number of full batches -- each train_op issues a
    gradient all-reduce, so one extra batch on one rank hangs the job -- and the shards must partition one tile list."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cli_hgan", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "histopathology_gan.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    for n_items in (0, 7, 63, 64, 65, 1000, 1001, 1023):
        for world in (2, 4, 8):
            for bs in (1, 8, 64):
                shards = [cli.shard_indices(n_items, r, world, bs) for r in range(world)]
                assert len({len(s) for s in shards}) == 1 and len(shards[0]) % bs == 0
                flat = [i for s in shards for i in s]
                assert len(set(flat)) == len(flat) and all(0 <= i < n_items for i in flat)
                assert len(flat) >= n_items - world * bs - world + 1 or n_items < world * bs


def test_collate_keeps_the_batch_size_under_data_parallel():
    """the reference's collate_fn drops unreadable records (src/histopathology_gan.py:24-34).  One process: the
    batch shrinks, as in the reference.  Data parallel: every rank must see the same batch size (gathered G.0 factors, captured
    step graphs and the per-train_op collectives are sized by it), so the dropped records are replaced by repeating the
    readable ones; a batch with nothing readable is an error there."""
    import importlib.util
    import pytest
    import torch
    spec = importlib.util.spec_from_file_location("cli_hgan2", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "histopathology_gan.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    good = lambda v: {"image": torch.full((3, 4, 4), float(v)), "rna_data": torch.full((5,), float(v)), "labels": torch.tensor(0.0)}
    bad = {"image": None, "rna_data": torch.zeros(5), "labels": torch.tensor(0.0)}
    batch = [good(1), bad, good(2), bad, bad, good(3), good(4), bad]
    one = cli.make_collate_fn(True, 1)(list(batch))
    assert one["image"].shape[0] == 4 and one["rna_data"].shape == (4, 5)
    dp = cli.make_collate_fn(True, 8)(list(batch))
    assert dp["image"].shape[0] == 8 and dp["rna_data"].shape == (8, 5)
    assert sorted(dp["image"][:, 0, 0, 0].tolist()) == [1.0, 1.0, 2.0, 2.0, 3.0, 3.0, 4.0, 4.0]     # readable records, repeated
    assert torch.equal(dp["image"][:, 0, 0, 0], dp["rna_data"][:, 0])                               # rows stay paired
    with pytest.raises(RuntimeError, match="no readable"):
        cli.make_collate_fn(True, 2)([bad, bad])
    # tuple batches (the stock-loss datasets): (image, label)
    tup = [(torch.ones(3, 2, 2), torch.tensor(0.0)), (None, torch.tensor(0.0))]
    assert cli.make_collate_fn(False, 1)(list(tup))[0].shape[0] == 1
    assert cli.make_collate_fn(False, 2)(list(tup))[0].shape[0] == 2
