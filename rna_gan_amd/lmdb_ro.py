"""Read-only walker of an LMDB data file (``data.mdb`` / a ``subdir=False`` database file), no ``lmdb`` package needed.

The reference keeps every slide's tiles in one LMDB file (written by src/preprocess/patch_gen_grid.py:92-133 with
``lmdb.open(path, subdir=False, map_size=...)``, read by src/read_data.py:284-342 with ``readonly=True, lock=False`` and
``txn.get(key)`` / ``txn.stat()["entries"]`` on the unnamed main database).  This module restates the published on-disk
format of LMDB 0.9 (liblmdb ``mdb.c``: MDB_meta / MDB_db / MDB_page / MDB_node, 64-bit little-endian build, which is what
py-lmdb's wheels are) for exactly those two operations plus key iteration:

  file    = pages of ``psize`` bytes; pages 0 and 1 are the two meta pages, the one with the larger txnid is current
  page    = header {pgno u64, pad u16, flags u16, lower u16, upper u16 | overflow: pages u32} (16 bytes), then for
            branch / leaf pages an array of u16 node offsets ``ptrs[(lower - 16) / 2]`` (from the page start, key order)
  meta    = header + {magic 0xBEEFC0DE u32, version u32 (1), address u64, mapsize u64, dbs[2] (free-list DB, main DB),
            last_pg u64, txnid u64};  db = {pad u32 (dbs[0]: the page size), flags u16, depth u16, branch_pages u64,
            leaf_pages u64, overflow_pages u64, entries u64, root u64}  (48 bytes; root = 2^64-1: empty tree)
  node    = {lo u16, hi u16, flags u16, ksize u16, key bytes, data}: leaf -- data size = lo | hi << 16, data follows the
            key, or with F_BIGDATA (0x01) an u64 page number of an overflow page run whose payload starts 16 bytes in;
            branch -- child page = lo | hi << 16 | flags << 32, node 0's key is empty (less than everything)
  keys    = compared as byte strings (memcmp, then length) in the default main database

Only what the reference's databases use is supported: the unnamed main DB without MDB_DUPSORT / MDB_INTEGERKEY /
MDB_REVERSEKEY, no named sub-databases.  Anything else raises ``LmdbFormatError`` rather than returning wrong bytes.
No real LMDB file exists in the build container (neither liblmdb nor py-lmdb): the reader is tested against files
produced by an independent in-test writer of the same published format (tests/test_lmdb_cpu.py) -- parity with
liblmdb-written files is therefore UNPINNED and stated as such in DESIGN.md.
"""
from __future__ import annotations

import mmap
import os
import struct
from collections.abc import Mapping

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2, P_SUBP = 0x01, 0x02, 0x04, 0x08, 0x20, 0x40
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
MDB_REVERSEKEY, MDB_DUPSORT, MDB_INTEGERKEY = 0x02, 0x04, 0x08
PAGEHDRSZ = 16
P_INVALID = (1 << 64) - 1
_META = struct.Struct("<IIQQ")                    # magic, version, address, mapsize
_DB = struct.Struct("<IHHQQQQQ")                  # pad, flags, depth, branch, leaf, overflow, entries, root
_NODE = struct.Struct("<HHHH")


class LmdbFormatError(ValueError):
    pass


class LmdbReadOnly(Mapping):
    """``LmdbReadOnly(path)[key] -> bytes`` (KeyError when absent), ``len()`` = entries of the main database (what
    ``txn.stat()["entries"]`` reports), iteration = keys in order.  ``path``: the database file, or a directory holding
    ``data.mdb``.  The file is mapped read-only; values are copied out (``bytes``), as py-lmdb's ``txn.get`` does by default."""

    def __init__(self, path: str):
        if os.path.isdir(path):
            path = os.path.join(path, "data.mdb")
        self.path = path
        self._f = open(path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            self._f.close()
            raise LmdbFormatError("%s: too small for an LMDB file" % path)
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._size = size
        self._read_meta()

    # ---------------------------------------------------------------- meta
    def _meta_at(self, off):
        m = self._m
        if off + PAGEHDRSZ + _META.size + 2 * _DB.size + 16 > self._size:
            return None
        flags = struct.unpack_from("<H", m, off + 10)[0]
        magic, version, _, mapsize = _META.unpack_from(m, off + PAGEHDRSZ)
        if magic != MAGIC or not (flags & P_META):
            return None
        o = off + PAGEHDRSZ + _META.size
        free = _DB.unpack_from(m, o)
        main = _DB.unpack_from(m, o + _DB.size)
        last_pg, txnid = struct.unpack_from("<QQ", m, o + 2 * _DB.size)
        return {"version": version, "psize": free[0], "main": main, "last_pg": last_pg, "txnid": txnid, "mapsize": mapsize}

    def _read_meta(self):
        m0 = self._meta_at(0)
        if m0 is None:
            raise LmdbFormatError("%s: no LMDB meta page (magic 0xBEEFC0DE) at offset 0" % self.path)
        psize = m0["psize"]
        if psize < 512 or psize > 65536 or psize & (psize - 1):
            raise LmdbFormatError("%s: implausible page size %d" % (self.path, psize))
        m1 = self._meta_at(psize)
        meta = m0 if (m1 is None or m0["txnid"] >= m1["txnid"]) else m1      # the newer of the two meta pages
        if meta["version"] != 1:
            raise LmdbFormatError("%s: LMDB data version %d (this reader knows version 1)" % (self.path, meta["version"]))
        self.psize = psize
        _, flags, self.depth, _, _, _, self.entries, self.root = meta["main"]
        if flags & (MDB_REVERSEKEY | MDB_DUPSORT | MDB_INTEGERKEY):
            raise LmdbFormatError("%s: main database flags 0x%x (dupsort / integer / reverse keys) are not supported"
                                  % (self.path, flags))
        self.last_pg = meta["last_pg"]
        if self.root != P_INVALID and (self.root > self.last_pg or (self.last_pg + 1) * psize > self._size):
            raise LmdbFormatError("%s: truncated file (last page %d, file %d bytes)" % (self.path, self.last_pg, self._size))

    # ---------------------------------------------------------------- pages / nodes
    def _page(self, pgno):
        off = pgno * self.psize
        if pgno > self.last_pg or off + self.psize > self._size:
            raise LmdbFormatError("%s: page %d out of range" % (self.path, pgno))
        pg, _, flags, lower, upper = struct.unpack_from("<QHHHH", self._m, off)
        if pg != pgno:
            raise LmdbFormatError("%s: page %d carries page number %d" % (self.path, pgno, pg))
        return off, flags, lower, upper

    def _nkeys(self, lower):
        return (lower - PAGEHDRSZ) >> 1

    def _node(self, off, i):
        """(node offset, lo, hi, flags, ksize) of node i of the page at byte offset ``off``."""
        ptr = struct.unpack_from("<H", self._m, off + PAGEHDRSZ + 2 * i)[0]
        if ptr < PAGEHDRSZ or ptr + _NODE.size > self.psize:
            raise LmdbFormatError("%s: node offset %d outside its page" % (self.path, ptr))
        return (off + ptr,) + _NODE.unpack_from(self._m, off + ptr)

    def _key(self, noff, ksize):
        return self._m[noff + _NODE.size: noff + _NODE.size + ksize]

    def _leaf_value(self, noff, lo, hi, nflags, ksize):
        if nflags & (F_SUBDATA | F_DUPDATA):
            raise LmdbFormatError("%s: sub-database / duplicate records are not supported" % self.path)
        dsize = lo | (hi << 16)
        d = noff + _NODE.size + ksize
        if nflags & F_BIGDATA:
            ovpg = struct.unpack_from("<Q", self._m, d)[0]
            ooff = ovpg * self.psize
            if ovpg > self.last_pg:
                raise LmdbFormatError("%s: overflow page %d out of range" % (self.path, ovpg))
            pg, _, oflags, npages = struct.unpack_from("<QHHI", self._m, ooff)
            if pg != ovpg or not (oflags & P_OVERFLOW) or PAGEHDRSZ + dsize > npages * self.psize or \
                    ooff + PAGEHDRSZ + dsize > self._size:
                raise LmdbFormatError("%s: bad overflow page run at page %d" % (self.path, ovpg))
            return bytes(self._m[ooff + PAGEHDRSZ: ooff + PAGEHDRSZ + dsize])
        if d + dsize > (noff // self.psize + 1) * self.psize:
            raise LmdbFormatError("%s: leaf data crosses its page" % self.path)
        return bytes(self._m[d: d + dsize])

    # ---------------------------------------------------------------- Mapping
    def __len__(self):
        return int(self.entries)

    def stat(self):
        return {"psize": self.psize, "depth": self.depth, "entries": int(self.entries)}

    def __getitem__(self, key) -> bytes:
        key = bytes(key)
        if self.root == P_INVALID:
            raise KeyError(key)
        pgno = self.root
        for _ in range(64):                                       # a B+tree over 2^48 pages is far shallower
            off, flags, lower, _ = self._page(pgno)
            n = self._nkeys(lower)
            if flags & P_LEAF2:
                raise LmdbFormatError("%s: LEAF2 (dupfixed) pages are not supported" % self.path)
            if flags & P_BRANCH:
                # node 0's key is the implicit minimum: the child is the last node whose key <= the search key
                lo_i, hi_i = 1, n - 1
                child = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    noff, _, _, _, ks = self._node(off, mid)
                    if self._key(noff, ks) <= key:                # bytes compare = memcmp then length (mdb_cmp_memn)
                        child, lo_i = mid, mid + 1
                    else:
                        hi_i = mid - 1
                noff, lo, hi, nflags, _ = self._node(off, child)
                pgno = lo | (hi << 16) | (nflags << 32)
                continue
            if not (flags & P_LEAF):
                raise LmdbFormatError("%s: page %d is neither branch nor leaf (flags 0x%x)" % (self.path, pgno, flags))
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) >> 1
                noff, lo, hi, nflags, ks = self._node(off, mid)
                k = self._key(noff, ks)
                if k == key:
                    return self._leaf_value(noff, lo, hi, nflags, ks)
                if k < key:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            raise KeyError(key)
        raise LmdbFormatError("%s: tree deeper than 64 levels (cycle?)" % self.path)

    def _walk(self, pgno, depth=0):
        if depth > 64:
            raise LmdbFormatError("%s: tree deeper than 64 levels (cycle?)" % self.path)
        off, flags, lower, _ = self._page(pgno)
        n = self._nkeys(lower)
        if flags & P_BRANCH:
            for i in range(n):
                _, lo, hi, nflags, _ = self._node(off, i)
                yield from self._walk(lo | (hi << 16) | (nflags << 32), depth + 1)
        elif flags & P_LEAF and not flags & P_LEAF2:
            for i in range(n):
                yield self._node(off, i)
        else:
            raise LmdbFormatError("%s: unexpected page flags 0x%x in the main tree" % (self.path, flags))

    def __iter__(self):
        if self.root == P_INVALID:
            return iter(())
        return (bytes(self._key(noff, ks)) for noff, _, _, _, ks in self._walk(self.root))

    def items(self):
        if self.root == P_INVALID:
            return
        for noff, lo, hi, nflags, ks in self._walk(self.root):
            yield bytes(self._key(noff, ks)), self._leaf_value(noff, lo, hi, nflags, ks)

    def close(self):
        if self._m is not None:
            self._m.close()
            self._f.close()
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
