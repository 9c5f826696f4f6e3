"""Input side of the training path: the reference's tile records, its per-slide tile sampling and its
RNA table preparation, restated without the packages that are absent here.

* **Tile store** (writer src/preprocess/patch_gen_grid.py:92-142, readers src/read_data.py:146-372): one key-value
  database per slide; value of key ``b"<i>"`` = ``lz4framed.compress(pickle.dumps((name, bytes, shape)))`` with
  ``bytes`` the uint8 HWC tile, and ``b"__keys__"`` = the compressed pickle of the key list.  The reference keeps the
  databases in LMDB files: ``open_tile_store`` reads them through the ``lmdb`` package when it is installed and through
  the built-in read-only walker of LMDB's on-disk B+tree otherwise (rna_gan_amd/lmdb_ro.py: meta pages, branch / leaf /
  overflow pages); any mapping (dict / shelve / a directory of files through ``DirStore``) is accepted as well -- the
  record format, the sampling and the decoding are the same for every backend.
* **LZ4 frame** (``lz4framed`` is absent): ``lz4f_decompress`` implements the LZ4 frame format v1.6 (magic 0x184D2204,
  FLG / BD / header checksum, independent or linked blocks, stored blocks, optional content size / block / content
  checksums -- checksums are skipped, not verified: xxHash32 is only needed to WRITE a header) and the LZ4 block format
  (token, literal / match lengths with 255-continuation, 2-byte little-endian offsets, overlapping copies);
  ``lz4f_compress`` writes valid frames (greedy hash-chain matcher, 64 KB independent blocks).  Parity with the
  lz4framed package itself is unpinned (not installable here); the decoder is tested on hand-assembled frames.
* **RNA table** (src/histopathology_gan.py:131-151): natural log with zeros kept at 0, columns reordered to
  [rna_*, others], StandardScaler over the rna_ columns (population standard deviation, constant columns scaled by 1).
"""
from __future__ import annotations

import os
import pickle
import random
import struct
from typing import Any, List, Mapping, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

# ------------------------------------------------------------------------------------------------------------------
# LZ4 block + frame format
# ------------------------------------------------------------------------------------------------------------------
LZ4F_MAGIC = 0x184D2204


def lz4_block_decompress(src: bytes, prefix: bytes = b"", max_out: Optional[int] = None) -> bytes:
    """One LZ4 block.  ``prefix``: previously decoded data a linked block may reference (up to 64 KB back)."""
    out = bytearray(prefix)
    base = len(out)
    i, n = 0, len(src)
    while i < n:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]
        if i + lit > n:
            raise ValueError("lz4: literal run past the end of the block")
        i += lit
        if i >= n:                      # the last sequence has literals only
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        if off == 0 or off > len(out):
            raise ValueError("lz4: invalid match offset")
        ml = (token & 15) + 4
        if (token & 15) == 15:
            while True:
                b = src[i]; i += 1
                ml += b
                if b != 255:
                    break
        start = len(out) - off
        if off >= ml:
            out += out[start:start + ml]
        else:                            # overlapping copy: the pattern of `off` bytes repeats
            pat = bytes(out[start:])
            reps = ml // off + 1
            out += (pat * reps)[:ml]
        if max_out is not None and len(out) - base > max_out:
            raise ValueError("lz4: block larger than the frame's block size")
    return bytes(out[base:])


def lz4f_decompress(data: bytes) -> bytes:
    """LZ4 frame -> bytes (what lz4framed.decompress returns).  Concatenated frames are concatenated."""
    out = bytearray()
    pos, n = 0, len(data)
    while pos < n:
        if n - pos < 7:
            raise ValueError("lz4f: truncated frame")
        magic, = struct.unpack_from("<I", data, pos); pos += 4
        if 0x184D2A50 <= magic <= 0x184D2A5F:                     # skippable frame
            size, = struct.unpack_from("<I", data, pos); pos += 4 + size
            continue
        if magic != LZ4F_MAGIC:
            raise ValueError("lz4f: bad magic 0x%08x" % magic)
        desc_start = pos
        flg, bd = data[pos], data[pos + 1]; pos += 2
        if (flg >> 6) != 1:
            raise ValueError("lz4f: unsupported version")
        independent, block_csum = bool(flg & 0x20), bool(flg & 0x10)
        has_size, content_csum, has_dict = bool(flg & 0x08), bool(flg & 0x04), bool(flg & 0x01)
        max_block = {4: 1 << 16, 5: 1 << 18, 6: 1 << 20, 7: 1 << 22}.get((bd >> 4) & 7)
        if max_block is None:
            raise ValueError("lz4f: bad block size code")
        content_size = None
        if has_size:
            content_size, = struct.unpack_from("<Q", data, pos); pos += 8
        if has_dict:
            pos += 4
        if pos >= n or data[pos] != (_xxh32(data[desc_start:pos]) >> 8) & 0xFF:   # liblz4 rejects the frame as well
            raise ValueError("lz4f: frame header checksum mismatch")
        pos += 1
        frame_start = len(out)
        while True:
            bsz, = struct.unpack_from("<I", data, pos); pos += 4
            if bsz == 0:                                          # EndMark
                break
            stored = bool(bsz & 0x80000000)
            bsz &= 0x7FFFFFFF
            blk = data[pos:pos + bsz]
            if len(blk) != bsz:
                raise ValueError("lz4f: truncated block")
            pos += bsz
            if block_csum:
                if n - pos < 4 or struct.unpack_from("<I", data, pos)[0] != _xxh32(blk):
                    raise ValueError("lz4f: block checksum mismatch")
                pos += 4
            if stored:
                out += blk
            else:
                prefix = b"" if independent else bytes(out[max(frame_start, len(out) - 65536):])
                out += lz4_block_decompress(blk, prefix, max_block)
        if content_csum:
            if n - pos < 4 or struct.unpack_from("<I", data, pos)[0] != _xxh32(bytes(out[frame_start:])):
                raise ValueError("lz4f: content checksum mismatch")
            pos += 4
        if content_size is not None and len(out) - frame_start != content_size:
            raise ValueError("lz4f: content size mismatch")
    return bytes(out)


def _xxh32(data: bytes, seed: int = 0) -> int:
    """xxHash32 (needed for the frame header checksum byte)."""
    P1, P2, P3, P4, P5 = 2654435761, 2246822519, 3266489917, 668265263, 374761393
    M = 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M            # noqa: E731
    n, i = len(data), 0
    if n >= 16:
        v = [(seed + P1 + P2) & M, (seed + P2) & M, seed & M, (seed - P1) & M]
        while i <= n - 16:
            for k in range(4):
                w, = struct.unpack_from("<I", data, i + 4 * k)
                v[k] = (rotl((v[k] + w * P2) & M, 13) * P1) & M
            i += 16
        h = (rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18)) & M
    else:
        h = (seed + P5) & M
    h = (h + n) & M
    while i <= n - 4:
        w, = struct.unpack_from("<I", data, i)
        h = (rotl((h + w * P3) & M, 17) * P4) & M
        i += 4
    while i < n:
        h = (rotl((h + data[i] * P5) & M, 11) * P1) & M
        i += 1
    h ^= h >> 15; h = (h * P2) & M
    h ^= h >> 13; h = (h * P3) & M
    h ^= h >> 16
    return h


def lz4_block_compress(src: bytes) -> bytes:
    """Greedy LZ4 block compressor (4-byte hash table, last occurrence).  Respects the format's end-of-block rules: the
    last 5 bytes are literals and no match starts within the last 12 bytes."""
    n = len(src)
    out = bytearray()
    table = {}
    anchor, i = 0, 0
    limit = n - 12

    def emit(lit_end, mlen, off):
        lit = lit_end - anchor
        token_l = 15 if lit >= 15 else lit
        token_m = 0 if mlen == 0 else (15 if mlen - 4 >= 15 else mlen - 4)
        out.append((token_l << 4) | token_m)
        if lit >= 15:
            r = lit - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(src[anchor:lit_end])
        if mlen:
            out.append(off & 255); out.append(off >> 8)
            if mlen - 4 >= 15:
                r = mlen - 4 - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)

    while i < limit:
        key = src[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and src[cand + m] == src[i + m]:
                m += 1
            emit(i, m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(n, 0, 0)
    return bytes(out)


def lz4f_compress(data: bytes, block_size: int = 1 << 16) -> bytes:
    """bytes -> LZ4 frame (independent blocks of one of the format's four sizes, 64 KB by default, no checksums besides
    the mandatory header byte, content size recorded): readable by any LZ4 frame decoder -- checked against liblz4's
    LZ4F_decompress, the call lz4framed.decompress makes (tests/test_lz4_system_cpu.py)."""
    code = {1 << 16: 4, 1 << 18: 5, 1 << 20: 6, 1 << 22: 7}.get(block_size)
    if code is None:
        raise ValueError("lz4f: block size must be 64 KB, 256 KB, 1 MB or 4 MB")
    flg = (1 << 6) | 0x20 | 0x08                                   # version 01, independent blocks, content size present
    bd = code << 4
    desc = bytes([flg, bd]) + struct.pack("<Q", len(data))
    out = bytearray(struct.pack("<I", LZ4F_MAGIC) + desc + bytes([(_xxh32(desc) >> 8) & 0xFF]))
    for off in range(0, len(data), block_size):
        chunk = data[off:off + block_size]
        comp = lz4_block_compress(chunk) if len(chunk) > 16 else None
        if comp is not None and len(comp) < len(chunk):
            out += struct.pack("<I", len(comp)) + comp
        else:
            out += struct.pack("<I", len(chunk) | 0x80000000) + chunk
    out += struct.pack("<I", 0)
    return bytes(out)


# ------------------------------------------------------------------------------------------------------------------
# tile records
# ------------------------------------------------------------------------------------------------------------------
def serialize_and_compress(obj) -> bytes:
    """src/preprocess/patch_gen_grid.py:145-146."""
    return lz4f_compress(pickle.dumps(obj))


def encode_record(name: str, image_hwc_u8: np.ndarray) -> bytes:
    """Value of one tile key (src/preprocess/patch_gen_grid.py:129-132): (name, raw bytes, shape)."""
    image = np.ascontiguousarray(image_hwc_u8, dtype=np.uint8)
    return serialize_and_compress((name, image.tobytes(), image.shape))


def encode_keys(n: int) -> bytes:
    """Value of b"__keys__" (src/preprocess/patch_gen_grid.py:138-140): the ascii keys b"0" .. b"n-1"."""
    return serialize_and_compress([u"{}".format(k).encode("ascii") for k in range(n)])


def decompress_and_deserialize(value: bytes):
    """src/read_data.py:235-243 / :330-337: record -> uint8 CHW tensor, channel order reversed (the reference calls
    cv2.cvtColor(image, cv2.COLOR_BGR2RGB), which swaps the first and third channel); None for an unreadable record."""
    try:
        _name, arr, shape = pickle.loads(lz4f_decompress(value))
    except Exception:
        return None
    image = np.frombuffer(arr, dtype=np.uint8).reshape(shape)
    image = np.ascontiguousarray(image[:, :, ::-1])
    return torch.from_numpy(image).permute(2, 0, 1)


class DirStore(Mapping):
    """A slide database as a directory: one file per key (name = the ascii key, ``__keys__`` included)."""

    def __init__(self, path):
        self.path = path

    def __getitem__(self, key: bytes) -> bytes:
        try:
            with open(os.path.join(self.path, key.decode("ascii")), "rb") as f:
                return f.read()
        except OSError:
            raise KeyError(key)

    def __iter__(self):
        return (n.encode("ascii") for n in os.listdir(self.path))

    def __len__(self):
        return len(os.listdir(self.path))


class _LmdbStore(Mapping):
    def __init__(self, path):
        import lmdb
        self.env = lmdb.open(path, subdir=os.path.isdir(path), readonly=True, lock=False, readahead=False, meminit=False)

    def __getitem__(self, key):
        with self.env.begin(write=False) as txn:
            v = txn.get(key)
        if v is None:
            raise KeyError(key)
        return v

    def __iter__(self):
        with self.env.begin(write=False) as txn:
            return iter([k for k, _ in txn.cursor()])

    def __len__(self):
        with self.env.begin(write=False) as txn:
            return txn.stat()["entries"]

    def close(self):
        self.env.close()


def open_tile_store(path):
    """``path``: a mapping (returned as is), a directory written by ``write_tile_store``, or an LMDB file as the reference
    writes them (src/preprocess/patch_gen_grid.py:92-133): read through the ``lmdb`` package when it is installed, else by
    the built-in read-only walker of the LMDB file format (rna_gan_amd.lmdb_ro)."""
    if isinstance(path, Mapping):
        return path
    if os.path.isdir(path) and not os.path.exists(os.path.join(path, "data.mdb")):
        return DirStore(path)
    try:
        return _LmdbStore(path)
    except ImportError:
        from .lmdb_ro import LmdbReadOnly
        return LmdbReadOnly(path)


def write_tile_store(path: str, tiles_hwc_u8, slide_id: str = "slide"):
    """Directory store with the reference's record format (keys b"0".., b"__keys__")."""
    os.makedirs(path, exist_ok=True)
    n = 0
    for i, tile in enumerate(tiles_hwc_u8):
        with open(os.path.join(path, str(i)), "wb") as f:
            f.write(encode_record("{0}_patch_{1}".format(slide_id, i), tile))
        n += 1
    with open(os.path.join(path, "__keys__"), "wb") as f:
        f.write(encode_keys(n))
    return n


# ------------------------------------------------------------------------------------------------------------------
# RNA table (src/histopathology_gan.py:131-151)
# ------------------------------------------------------------------------------------------------------------------
def load_slide_tables(path_csv, patch_data_path):
    """The slide table of a run (src/histopathology_gan.py:111-127): one CSV per tissue, each row tagged with its tile
    directory (``patch_data_path``) and the index of its CSV as the tissue id (``labels``), the tables concatenated in the
    order given (a single path may be passed as a string)."""
    import pandas as pd
    if isinstance(path_csv, str):
        path_csv, patch_data_path = [path_csv], [patch_data_path]
    if len(path_csv) != len(patch_data_path):
        raise ValueError("path_csv and patch_data_path must have one entry per tissue")
    tables = []
    for i, (csv_file, path) in enumerate(zip(path_csv, patch_data_path)):
        df = pd.read_csv(csv_file)
        df["patch_data_path"] = [path] * df.shape[0]
        df["labels"] = [i] * df.shape[0]
        tables.append(df)
    return pd.concat(tables) if len(tables) > 1 else tables[0]


def log_standardize_rna(df):
    """Returns (DataFrame with columns [rna_*, others], mean, scale).  rna_ columns: ln(x) with zeros left at 0
    (:133-136), then StandardScaler().fit_transform (:148-151): (x - mean) / sqrt(population variance), a zero
    variance replaced by 1."""
    rna_columns = [c for c in df.columns if "rna_" in c]
    other = [c for c in df.columns if "rna_" not in c]
    x = df[rna_columns].to_numpy(dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        lx = np.where(x == 0, 0.0, np.log(x))
    lx = np.where(np.isnan(lx), 0.0, lx)                           # np.log(x.replace(0, nan)).replace(nan, 0): NaNs of negatives too
    mean = lx.mean(axis=0)
    var = lx.var(axis=0)
    scale = np.sqrt(var)
    scale[scale < 10 * np.finfo(np.float64).eps] = 1.0            # sklearn: constant features are not scaled
    out = df[rna_columns + other].copy()
    out[rna_columns] = (lx - mean) / scale
    return out, mean, scale


# ------------------------------------------------------------------------------------------------------------------
# datasets (src/read_data.py:146-372)
# ------------------------------------------------------------------------------------------------------------------
class PatchRNADataset(Dataset):
    """src/read_data.py:266-372: rows of the (prepared) slide table -> a random sample of at most ``max_patches_total``
    tiles per slide (``random.sample(range(n_tiles), n_selected)`` per row, in row order, on the global ``random``
    generator as the reference does) -> items ``{"image", "rna_data", "labels"}``.  ``stores``: ``{db path: mapping}``
    overrides for ``open_tile_store`` (tests, in-memory stores)."""

    def __init__(self, patch_data_path, csv_path, img_size, transforms=None, max_patches_total=300, quick=False, le=None,
                 stores: Optional[Mapping[str, Any]] = None, with_rna=True):
        import pandas as pd
        self.patch_data_path, self.csv_path, self.img_size = patch_data_path, csv_path, img_size
        self.transforms, self.max_patches_total, self.quick, self.le = transforms, max_patches_total, quick, le
        self.keys: List[bytes] = []
        self.images: List[int] = []
        self.filenames: List[str] = []
        self.labels: List[torch.Tensor] = []
        self.lmdbs_path: List[str] = []
        self.rna_data_arrays: List[torch.Tensor] = []
        self.with_rna = with_rna
        self._stores = dict(stores or {})
        if isinstance(csv_path, str):
            table = pd.read_csv(csv_path)
            table["patch_data_path"] = [patch_data_path] * table.shape[0]
            table["labels"] = [0] * table.shape[0]
        else:
            table = csv_path
        if quick:
            table = table.sample(150)
        for _, row in table.iterrows():
            wsi = row["wsi_file_name"]
            rna = torch.tensor(row[[c for c in row.keys() if "rna_" in c]].values.astype(np.float32), dtype=torch.float32)
            label = np.asarray(row["labels"])
            if le is not None:
                label = le.transform(label.reshape(-1, 1))
            label = torch.tensor(label, dtype=torch.float32)
            path = os.path.join(row["patch_data_path"], wsi, wsi.replace(".svs", ".db"))
            try:
                # opened for the count and the key list only, then let go -- as the reference does (one lmdb.open per
                # row, src/read_data.py:306-312): no descriptor / mapping per slide is held for the dataset's lifetime
                store, owned = self._open(path)
                try:
                    n_patches = len(store) - 1                               # every entry but __keys__
                    keys = pickle.loads(lz4f_decompress(store[b"__keys__"]))
                finally:
                    if owned and hasattr(store, "close"):
                        store.close()
                index = random.sample(list(range(n_patches)), min(n_patches, max_patches_total))
            except OSError as e:
                import errno
                if e.errno in (errno.EMFILE, errno.ENFILE, errno.ENOMEM):     # resource exhaustion is not "a bad database"
                    raise
                print("Error with db {}".format(path))
                continue
            except Exception:
                print("Error with db {}".format(path))
                continue
            for i in index:
                self.images.append(i)
                self.filenames.append(wsi)
                self.labels.append(label)
                self.lmdbs_path.append(path)
                self.keys.append(keys[i])
                self.rna_data_arrays.append(rna)

    MAX_OPEN_STORES = 64          # per process: an LMDB environment is a descriptor + a mapping

    def _open(self, path):
        """(store, owned): a caller-supplied mapping (not owned), or a freshly opened store the caller closes."""
        st = self._stores.get(path)
        if st is not None:
            return st, False
        return open_tile_store(path if os.path.exists(path) else path[:-3]), True

    def _store(self, path):
        """Store for item access: caller-supplied mappings as they are; files opened lazily in the process that reads them
        (a DataLoader worker never inherits an environment opened by its parent -- LMDB forbids using one across fork)
        and kept in a small per-process LRU."""
        st = self._stores.get(path)
        if st is not None:
            return st
        pid = os.getpid()
        if getattr(self, "_lru_pid", None) != pid:
            self._lru, self._lru_pid = {}, pid
        st = self._lru.pop(path, None)
        if st is None:
            st, _ = self._open(path)
            while len(self._lru) >= self.MAX_OPEN_STORES:
                old = self._lru.pop(next(iter(self._lru)))
                if hasattr(old, "close"):
                    old.close()
        self._lru[path] = st                                                  # most recently used last
        return st

    def __getstate__(self):           # pickled into spawned DataLoader workers: open stores stay behind
        d = dict(self.__dict__)
        d.pop("_lru", None); d.pop("_lru_pid", None)
        return d

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        try:
            value = self._store(self.lmdbs_path[idx])[self.keys[idx]]
        except KeyError:
            value = None          # txn.get() of a missing key is None in the reference: decoded to image None, dropped by collate_fn
        image = decompress_and_deserialize(value)
        if image is not None and self.transforms is not None:
            image = self.transforms(image)
        if not self.with_rna:
            return image, self.labels[idx]
        return {"image": image, "rna_data": self.rna_data_arrays[idx], "labels": self.labels[idx]}


class PatchDataset(PatchRNADataset):
    """src/read_data.py:146-264: the same sampling without the RNA vector; items are (image, label)."""

    def __init__(self, *a, **k):
        k["with_rna"] = False
        super().__init__(*a, **k)


class ToFloatNormalize:
    """The reference's transform (src/histopathology_gan.py:106-109) on the host: uint8 CHW -> float / 255 -> (x - mean) /
    std.  The device-side equivalent is rg_u8_to_norm (HipOps.u8_to_norm): give the dataset ``transforms=None`` and
    normalise the collated uint8 batch on the GPU."""

    def __init__(self, mean=0.5, std=0.5):
        self.mean, self.std = mean, std

    def __call__(self, image_u8_chw):
        return (image_u8_chw.float() / 255.0 - self.mean) / self.std
