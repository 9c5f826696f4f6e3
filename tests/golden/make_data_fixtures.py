#!/usr/bin/env python3
"""Golden fixture F10 for the input side, build container only.

(a) Dataset logic: the REFERENCE's src/read_data.py is imported behind sys.modules stubs for the packages that are absent
    here -- ``lmdb`` (a dict-backed environment with the three calls the reader makes: open / begin / stat+get),
    ``lz4framed`` (compress / decompress = this build's LZ4 frame codec, so the stub only supplies the container
    format, not the reader's logic), ``cv2`` (cvtColor(BGR2RGB) = channel reversal, its documented effect),
    ``torchvision.io`` (unused symbol).  Its PatchRNADataset / PatchDataset then run unmodified on seeded slide
    databases: which tiles are sampled (random.sample order), the keys looked up, the decoded uint8 CHW tensors, the
    item dictionaries.  Stored: per item (slide, key, image checksum, first pixels), rna vector, label.
(b) RNA table: the call sequence of src/histopathology_gan.py:131-151 executed with the real pandas and scikit-learn
    (the code is inline in the script's main() and cannot be imported): log with zeros kept, column reorder,
    StandardScaler.fit_transform.  Stored: the transformed table.

    python tests/golden/make_data_fixtures.py   ->  f10_data.npz
"""
import os
import pickle
import random
import sys
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

from data_cases import make_slides, make_table  # noqa: E402  (tests/golden/data_cases.py: seeded inputs)
from rna_gan_amd import data as PD  # noqa: E402       (container codec only)

STORES = {}


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Txn:
    def __init__(self, store):
        self.store = store

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def stat(self):
        return {"entries": len(self.store)}

    def get(self, key):
        return self.store.get(key)


class _Env:
    def __init__(self, path, **kw):
        if path not in STORES:
            raise FileNotFoundError(path)
        self.store = STORES[path]

    def begin(self, write=False):
        return _Txn(self.store)


_stub("lmdb", open=lambda path, **kw: _Env(path, **kw))
_stub("lz4framed", compress=PD.lz4f_compress, decompress=PD.lz4f_decompress)
_stub("cv2", COLOR_BGR2RGB=4, cvtColor=lambda img, code: np.ascontiguousarray(img[:, :, ::-1]))
_tv = _stub("torchvision")
_tv.io = _stub("torchvision.io", read_image=None)

import read_data as ref_read_data  # noqa: E402   (reference)


def main():
    out = {}
    slides = make_slides()
    for wsi, tiles in slides.items():
        path = os.path.join("/data/patches", wsi, wsi.replace(".svs", ".db"))
        st = {u"{}".format(i).encode("ascii"): PD.encode_record("{0}_patch_{1}".format(wsi, i), t) for i, t in enumerate(tiles)}
        st[b"__keys__"] = PD.encode_keys(len(tiles))
        STORES[path] = st
    table = make_table(list(slides))
    # (b) RNA preparation: the reference's OWN TEXT (src/histopathology_gan.py:133-151 is inline code of its main(), not an
    # importable function), extracted at generation time and executed on the synthetic table -- nothing is re-typed here
    import linecache
    import textwrap
    from sklearn.preprocessing import StandardScaler
    ref_file = os.path.join(REF, "histopathology_gan.py")
    text = "".join(linecache.getline(ref_file, n) for n in range(133, 152))
    assert "def _get_log" in text and "scaler.fit_transform" in text and "train_df[rna_columns] = rna_values" in text, \
        "src/histopathology_gan.py:133-151 is not the RNA preparation block any more"
    ns = {"np": np, "StandardScaler": StandardScaler, "train_df": table.copy()}
    exec(compile(textwrap.dedent(text), ref_file + ":133-151", "exec"), ns)
    train_df = ns["train_df"]
    rna_columns = ns["rna_columns"]
    out["rna.columns"] = np.array(list(train_df.columns))
    out["rna.values"] = train_df[rna_columns].values.astype(np.float64)
    # (a) the reference datasets on the prepared table
    train_df["patch_data_path"] = ["/data/patches"] * train_df.shape[0]
    train_df["labels"] = [0, 1, 0][:train_df.shape[0]]
    for name, cls in (("rna", ref_read_data.PatchRNADataset), ("plain", ref_read_data.PatchDataset)):
        random.seed(1234)
        ds = cls("/data/patches", train_df, 16, transforms=lambda im: im.float() / 255.0, max_patches_total=5)
        out[name + ".len"] = np.int64(len(ds))
        out[name + ".filenames"] = np.array(ds.filenames)
        out[name + ".keys"] = np.array([k.decode() for k in ds.keys])
        sums, heads, rnas, labels = [], [], [], []
        for i in range(len(ds)):
            item = ds[i]
            img = item["image"] if isinstance(item, dict) else item[0]
            lab = item["labels"] if isinstance(item, dict) else item[1]
            assert img.shape == (3, 16, 16) and img.dtype == torch.float32
            sums.append(float(img.double().sum())); heads.append(img[:, 0, :4].numpy().copy())
            labels.append(float(lab))
            if isinstance(item, dict):
                rnas.append(item["rna_data"].numpy().copy())
        out[name + ".image_sums"] = np.array(sums)
        out[name + ".image_heads"] = np.stack(heads)
        out[name + ".labels"] = np.array(labels)
        if rnas:
            out[name + ".rna"] = np.stack(rnas)
    np.savez_compressed(os.path.join(HERE, "f10_data.npz"), **out)
    print("wrote f10_data.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
