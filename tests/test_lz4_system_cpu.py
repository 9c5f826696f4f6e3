"""The tile records' LZ4 frame codec (rna_gan_amd/data.py) against the REAL liblz4 ("own LZ4
frame codec is checked only against hand-assembled frames and itself").

The reference stores a tile as lz4framed.compress(pickle(...)) (src/read_data.py:247-256 decompress_and_deserialize,
src/preprocess/patch_gen_grid.py:76-79); py-lz4framed is a thin binding of liblz4's LZ4F_* frame API.  That package is absent
here, the C library it wraps is installed (liblz4.so.1), so the pin goes through ctypes:

* frames written by LZ4F_compressFrame -- lz4framed.compress's call -- with the block sizes / block modes / checksum flags /
  content-size field the frame format allows are decoded by data.lz4f_decompress to the original bytes;
* frames written by data.lz4f_compress (what data.write_tile_store stores) are decoded by LZ4F_decompress;
* blocks: LZ4_compress_default -> data.lz4_block_decompress and data.lz4_block_compress -> LZ4_decompress_safe;
* xxHash32 of the frame descriptor / content checksum against libxxhash's XXH32 when that library is present.

Skipped when liblz4 cannot be loaded (nothing else on the image provides the format).
"""
import ctypes
import ctypes.util
import pickle

import numpy as np
import pytest

from rna_gan_amd import data as PD


def _load(name, soname):
    for cand in (ctypes.util.find_library(name), soname):
        if cand:
            try:
                return ctypes.CDLL(cand)
            except OSError:
                pass
    return None


LZ4 = _load("lz4", "liblz4.so.1")
XXH = _load("xxhash", "libxxhash.so.0")
needs_lz4 = pytest.mark.skipif(LZ4 is None, reason="liblz4 not installed")


class _FrameInfo(ctypes.Structure):          # LZ4F_frameInfo_t (lz4frame.h, v1.8+)
    _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int), ("contentChecksumFlag", ctypes.c_int),
                ("frameType", ctypes.c_int), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                ("blockChecksumFlag", ctypes.c_int)]


class _Prefs(ctypes.Structure):              # LZ4F_preferences_t
    _fields_ = [("frameInfo", _FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]


if LZ4 is not None:
    LZ4.LZ4F_compressFrameBound.restype = ctypes.c_size_t
    LZ4.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    LZ4.LZ4F_compressFrame.restype = ctypes.c_size_t
    LZ4.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    LZ4.LZ4F_isError.restype = ctypes.c_uint
    LZ4.LZ4F_isError.argtypes = [ctypes.c_size_t]
    LZ4.LZ4F_createDecompressionContext.restype = ctypes.c_size_t
    LZ4.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    LZ4.LZ4F_freeDecompressionContext.restype = ctypes.c_size_t
    LZ4.LZ4F_freeDecompressionContext.argtypes = [ctypes.c_void_p]
    LZ4.LZ4F_decompress.restype = ctypes.c_size_t
    LZ4.LZ4F_decompress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    LZ4.LZ4_compressBound.restype = ctypes.c_int
    LZ4.LZ4_compressBound.argtypes = [ctypes.c_int]
    LZ4.LZ4_compress_default.restype = ctypes.c_int
    LZ4.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    LZ4.LZ4_decompress_safe.restype = ctypes.c_int
    LZ4.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]


def _lib_compress_frame(data: bytes, prefs=None) -> bytes:
    p = ctypes.byref(prefs) if prefs is not None else None
    cap = LZ4.LZ4F_compressFrameBound(len(data), p)
    dst = ctypes.create_string_buffer(cap)
    n = LZ4.LZ4F_compressFrame(dst, cap, data, len(data), p)
    assert not LZ4.LZ4F_isError(n), "LZ4F_compressFrame failed"
    return dst.raw[:n]


def _lib_decompress_frame(frame: bytes, expect_len: int) -> bytes:
    ctx = ctypes.c_void_p()
    assert not LZ4.LZ4F_isError(LZ4.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100))
    try:
        out = bytearray()
        src = ctypes.create_string_buffer(frame, len(frame))
        pos = 0
        buf = ctypes.create_string_buffer(max(1 << 16, expect_len + 64))
        while pos < len(frame):
            dn = ctypes.c_size_t(len(buf))
            sn = ctypes.c_size_t(len(frame) - pos)
            r = LZ4.LZ4F_decompress(ctx, buf, ctypes.byref(dn), ctypes.byref(src, pos), ctypes.byref(sn), None)
            assert not LZ4.LZ4F_isError(r), "LZ4F_decompress rejected the frame"
            out += buf.raw[:dn.value]
            pos += sn.value
            if r == 0 and pos >= len(frame):
                break
            assert sn.value or dn.value, "no progress"
        assert r == 0, "frame incomplete for liblz4"
        return bytes(out)
    finally:
        LZ4.LZ4F_freeDecompressionContext(ctx)


def _payloads():
    rng = np.random.default_rng(11)
    tile = (rng.integers(0, 256, (64, 64, 3)) // 32 * 32).astype(np.uint8)      # posterised tile: matches and literals
    smooth = np.add.outer(np.arange(256), np.arange(256)).astype(np.uint8).tobytes() * 3
    return [b"", b"a", b"hello world", bytes(5000), b"abcabcabc" * 30000, rng.bytes(70000), smooth,
            pickle.dumps(("slide_0_tile_3", tile.tobytes(), tile.shape)), rng.bytes(4 * 1024 * 1024 + 17)[: (1 << 22) + 17],
            (b"0123456789abcdef" * 5 + rng.bytes(11)) * 9000]


@needs_lz4
def test_decoder_reads_frames_written_by_liblz4():
    """LZ4F_compressFrame with the defaults lz4framed.compress uses (NULL preferences: 64 KB linked blocks, no checksums)
    and with every block size / block mode / checksum / content-size combination of the format."""
    for c in _payloads():
        assert PD.lz4f_decompress(_lib_compress_frame(c)) == c
    variants = 0
    for bsid in (0, 4, 5, 6, 7):                       # default, 64 KB, 256 KB, 1 MB, 4 MB
        for mode in (0, 1):                            # linked (matches reach into previous blocks), independent
            for ccheck in (0, 1):
                for bcheck in (0, 1):
                    for with_size in (False, True):
                        for level in (0, 9):           # fast and HC parsers produce different sequences
                            for c in _payloads()[3:]:
                                if len(c) > 300000 and (level == 9 or ccheck or bcheck):
                                    continue               # checksums are verified by a pure-Python xxHash32 here
                                pr = _Prefs()
                                pr.frameInfo.blockSizeID = bsid
                                pr.frameInfo.blockMode = mode
                                pr.frameInfo.contentChecksumFlag = ccheck
                                pr.frameInfo.blockChecksumFlag = bcheck
                                pr.frameInfo.contentSize = len(c) if with_size else 0
                                pr.compressionLevel = level
                                f = _lib_compress_frame(c, pr)
                                assert PD.lz4f_decompress(f) == c, (bsid, mode, ccheck, bcheck, with_size, level, len(c))
                                variants += 1
    assert variants > 500
    # the reference's record: decompress_and_deserialize(lz4framed.compress(pickle.dumps(obj))) (src/read_data.py:247-256)
    img = np.arange(64 * 64 * 3, dtype=np.uint32).astype(np.uint8).reshape(64, 64, 3)
    got = PD.decompress_and_deserialize(_lib_compress_frame(pickle.dumps(("img", img.tobytes(), img.shape))))
    assert got.shape == (3, 64, 64) and np.array_equal(got.numpy(), img[:, :, ::-1].transpose(2, 0, 1))
    # concatenated frames and a corrupted content checksum
    two = _lib_compress_frame(b"one") + _lib_compress_frame(b"two" * 1000)
    assert PD.lz4f_decompress(two) == b"one" + b"two" * 1000
    pr = _Prefs()
    pr.frameInfo.contentChecksumFlag = 1
    bad = bytearray(_lib_compress_frame(b"checksummed" * 100, pr))
    bad[-1] ^= 0x5A
    with pytest.raises(ValueError):
        PD.lz4f_decompress(bytes(bad))


@needs_lz4
def test_liblz4_reads_frames_written_here():
    """What data.write_tile_store / encode_record / encode_keys store is a frame the real library (and so lz4framed) accepts."""
    for c in _payloads():
        for bs in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
            assert _lib_decompress_frame(PD.lz4f_compress(c, block_size=bs), len(c)) == c
    with pytest.raises(ValueError):
        PD.lz4f_compress(b"x" * 100, block_size=1000)              # not one of the format's block sizes
    rec = PD.encode_record("slide_0_tile_7", np.arange(32 * 32 * 3, dtype=np.uint8).reshape(32, 32, 3))
    name, raw, shape = pickle.loads(_lib_decompress_frame(rec, 1 << 16))
    assert name == "slide_0_tile_7" and tuple(shape) == (32, 32, 3) and raw == bytes(np.arange(32 * 32 * 3, dtype=np.uint8))
    assert pickle.loads(_lib_decompress_frame(PD.encode_keys(5), 1 << 12)) == [b"0", b"1", b"2", b"3", b"4"]


@needs_lz4
def test_block_codec_against_liblz4():
    for c in _payloads()[1:]:
        if len(c) > (1 << 21):
            continue
        cap = LZ4.LZ4_compressBound(len(c))
        dst = ctypes.create_string_buffer(cap)
        n = LZ4.LZ4_compress_default(c, dst, len(c), cap)
        assert n > 0
        assert PD.lz4_block_decompress(dst.raw[:n]) == c
        mine = PD.lz4_block_compress(c)
        out = ctypes.create_string_buffer(len(c) + 1)
        m = LZ4.LZ4_decompress_safe(mine, out, len(mine), len(c) + 1)
        assert m == len(c) and out.raw[:m] == c


@pytest.mark.skipif(XXH is None, reason="libxxhash not installed")
def test_xxhash32_against_libxxhash():
    XXH.XXH32.restype = ctypes.c_uint
    XXH.XXH32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint]
    rng = np.random.default_rng(5)
    for n in list(range(0, 40)) + [63, 64, 65, 1000, 65536, 100003]:
        b = rng.bytes(n)
        for seed in (0, 1, 0x9E3779B1):
            assert PD._xxh32(b, seed) == XXH.XXH32(b, n, seed), (n, seed)
