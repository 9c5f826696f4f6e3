"""Test infrastructure: a minimal WRITER of LMDB 0.9 data files (64-bit little-endian layout), independent of the
reader in rna_gan_amd/lmdb_ro.py -- it lays pages out the way liblmdb's mdb.c does (meta pages 0/1, nodes packed from the
top of a page with the u16 pointer array growing from the header, values beyond the node limit on overflow page runs,
branch nodes carrying the child page number in lo/hi/flags with an empty key on node 0).  No liblmdb / py-lmdb exists
in the build container, so this is the closest thing to a real file that can be produced here."""
import struct

MAGIC, P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0xBEEFC0DE, 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
HDR = 16
P_INVALID = (1 << 64) - 1


def _even(n):
    return (n + 1) & ~1


def write_lmdb(path, items, psize=4096, stale_first_meta=True):
    """items: {key bytes: value bytes}.  Returns a dict of layout facts (pages, depth, overflow pages) for assertions."""
    keys = sorted(items)                      # bytes order = memcmp then length = LMDB's default comparator
    nodemax = (((psize - HDR) // 2) & ~1) - 2
    pages = {}                                # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    def build_page(pgno, flags, nodes):
        """nodes: list of raw node bytes (already even-sized) in key order."""
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for nd in nodes:
            upper -= len(nd)
            buf[upper:upper + len(nd)] = nd
            ptrs.append(upper)
        lower = HDR + 2 * len(nodes)
        assert lower <= upper, "page overfull"
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into("<H", buf, HDR + 2 * i, p)
        pages[pgno] = bytes(buf)

    n_over = 0
    leaf_nodes = []                           # (key, raw node)
    for k in keys:
        v = items[k]
        if 8 + len(k) + len(v) > nodemax:
            npg = (HDR + len(v) + psize - 1) // psize
            pg = alloc(npg)
            n_over += npg
            run = bytearray(npg * psize)
            struct.pack_into("<QHHI", run, 0, pg, 0, P_OVERFLOW, npg)
            run[HDR:HDR + len(v)] = v
            for j in range(npg):
                pages[pg + j] = bytes(run[j * psize:(j + 1) * psize])
            raw = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", pg)
        else:
            raw = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        raw += b"\0" * (_even(len(raw)) - len(raw))
        leaf_nodes.append((k, raw))

    def pack_level(nodes, flags):
        """Greedy fill; returns [(first key, pgno)]."""
        out, cur, used = [], [], HDR
        for k, raw in nodes:
            if cur and used + len(raw) + 2 > psize:
                pg = alloc()
                build_page(pg, flags, [r for _, r in cur])
                out.append((cur[0][0], pg))
                cur, used = [], HDR
            cur.append((k, raw))
            used += len(raw) + 2
        if cur:
            pg = alloc()
            build_page(pg, flags, [r for _, r in cur])
            out.append((cur[0][0], pg))
        return out

    n_leaf = n_branch = 0
    depth = 0
    root = P_INVALID
    if leaf_nodes:
        level = pack_level(leaf_nodes, P_LEAF)
        n_leaf = len(level)
        depth = 1
        while len(level) > 1:
            nodes = []
            for i, (k, pg) in enumerate(level):
                kk = b"" if i == 0 else k
                raw = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(kk)) + kk
                raw += b"\0" * (_even(len(raw)) - len(raw))
                nodes.append((k, raw))
            # node 0 of EVERY branch page has an empty key
            grouped = pack_level(nodes, P_BRANCH)
            for (_, pg) in grouped:
                buf = bytearray(pages[pg])
                ptr0 = struct.unpack_from("<H", buf, HDR)[0]
                ks = struct.unpack_from("<H", buf, ptr0 + 6)[0]
                struct.pack_into("<H", buf, ptr0 + 6, 0)          # mdb.c never reads node 0's key on a branch page
                pages[pg] = bytes(buf)
            n_branch += len(grouped)
            level = grouped
            depth += 1
        root = level[0][1]
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, root_, entries):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        o = HDR
        struct.pack_into("<IIQQ", buf, o, MAGIC, 1, 0, (last_pg + 1) * psize)
        o += 24
        struct.pack_into("<IHHQQQQQ", buf, o, psize, 0, 0, 0, 0, 0, 0, P_INVALID)                # free-list DB (empty)
        o += 48
        struct.pack_into("<IHHQQQQQ", buf, o, 0, 0, depth if root_ != P_INVALID else 0, n_branch, n_leaf, n_over,
                         entries, root_)
        o += 48
        struct.pack_into("<QQ", buf, o, last_pg, txnid)
        return bytes(buf)

    # meta 0 = an OLDER transaction (empty tree) when stale_first_meta: a reader must take the newer one (meta 1)
    pages[0] = meta(0, 1, P_INVALID, 0) if stale_first_meta else meta(0, 3, root, len(keys))
    pages[1] = meta(1, 2, root, len(keys))
    with open(path, "wb") as f:
        for pg in range(last_pg + 1):
            f.write(pages.get(pg, b"\0" * psize))
    return {"depth": depth, "leaf_pages": n_leaf, "branch_pages": n_branch, "overflow_pages": n_over, "last_pg": last_pg}
