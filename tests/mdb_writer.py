"""Test infrastructure: writes a small LMDB environment file (data.mdb) from a dict of byte keys / values -- the layout of LMDB 0.9 (data
version 1, 64-bit little-endian) as described in mrn_amd/data/mdb.py: two meta pages, a B+tree of leaf / branch pages built bottom-up,
values above the node-size limit in overflow runs.  The build container has no LMDB implementation, so the fixtures of
tests/test_data_cpu.py::test_lmdb_* come from here ("self-pinned": reader and writer share one format description).  Key scheme of the
datasets: reference tools/create_lmdb_dataset.py:327-345 (`num-samples`, `label-%09d`, `image-%09d`)."""
import os
import struct

MAGIC, PAGEHDRSZ = 0xBEEFC0DE, 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
INVALID = (1 << 64) - 1


def _even(n):
    return (n + 1) & ~1


def write_environment(dirpath, items, page_size=4096, leaf_fill=1.0):
    """items: {bytes: bytes}.  leaf_fill < 1 leaves pages partly empty (more pages -> deeper trees for the same data)."""
    os.makedirs(dirpath, exist_ok=True)
    nodemax = (((page_size - PAGEHDRSZ) // 2) & ~1) - 2
    pages = {}                                   # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    def build_page(flags, nodes):
        """nodes: list of packed node byte strings (even length), already in key order"""
        pg = bytearray(page_size)
        upper = page_size
        ptrs = []
        for nd in nodes:
            upper -= len(nd)
            pg[upper:upper + len(nd)] = nd
            ptrs.append(upper)
        lower = PAGEHDRSZ + 2 * len(nodes)
        assert lower <= upper
        struct.pack_into("<%dH" % len(ptrs), pg, PAGEHDRSZ, *ptrs)
        return pg, flags, lower, upper

    def finish(pgno, pg, flags, lower, upper):
        struct.pack_into("<QHHHH", pg, 0, pgno, 0, flags, lower, upper)
        pages[pgno] = bytes(pg)

    # ---- leaves ----------------------------------------------------------------------------------------------------------
    keys = sorted(items)
    n_over = 0
    leaf_nodes = []
    for k in keys:
        v = items[k]
        if 8 + len(k) + len(v) > nodemax:
            npg = (PAGEHDRSZ + len(v) + page_size - 1) // page_size
            opg = alloc(npg)
            run = bytearray(npg * page_size)
            struct.pack_into("<QHHI", run, 0, opg, 0, P_OVERFLOW, npg)
            run[PAGEHDRSZ:PAGEHDRSZ + len(v)] = v
            for i in range(npg):                 # (only the first page of a run has a header; store the run page by page)
                pages[opg + i] = bytes(run[i * page_size:(i + 1) * page_size])
            n_over += npg
            body = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 1, len(k)) + k + struct.pack("<Q", opg)
        else:
            body = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        leaf_nodes.append((k, body + b"\0" * (_even(len(body)) - len(body))))
    budget = int((page_size - PAGEHDRSZ) * leaf_fill)
    level = []                                   # (first key, pgno) of every page of the current level
    cur, used = [], 0
    groups = []
    for k, nd in leaf_nodes:
        if cur and used + len(nd) + 2 > budget:
            groups.append(cur)
            cur, used = [], 0
        cur.append((k, nd))
        used += len(nd) + 2
    if cur:
        groups.append(cur)
    n_leaf = len(groups)
    for grp in groups:
        pgno = alloc()
        finish(pgno, *build_page(P_LEAF, [nd for _, nd in grp]))
        level.append((grp[0][0], pgno))
    # ---- branches --------------------------------------------------------------------------------------------------------
    depth, n_branch = (1 if level else 0), 0
    while len(level) > 1:
        nxt, cur, used = [], [], 0
        groups = []
        for i, (k, child) in enumerate(level):
            key = b"" if not cur else k              # node 0 of a branch page carries no key
            body = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(key)) + key
            nd = body + b"\0" * (_even(len(body)) - len(body))
            if cur and used + len(nd) + 2 > budget:
                groups.append(cur)
                cur, used = [], 0
                body = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, 0)
                nd = body
            cur.append((k, nd))
            used += len(nd) + 2
        if cur:
            groups.append(cur)
        for grp in groups:
            pgno = alloc()
            finish(pgno, *build_page(P_BRANCH, [nd for _, nd in grp]))
            nxt.append((grp[0][0], pgno))
            n_branch += 1
        level = nxt
        depth += 1
    root = level[0][1] if level else INVALID
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, main):
        pg = bytearray(page_size)
        struct.pack_into("<QHHHH", pg, 0, pgno, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", pg, PAGEHDRSZ, MAGIC, 1, 0, 1 << 30)
        base = PAGEHDRSZ + 24
        struct.pack_into("<IHHQQQQQ", pg, base, page_size, 0, 0, 0, 0, 0, 0, INVALID)          # free-list database (pad = page size)
        struct.pack_into("<IHHQQQQQ", pg, base + 48, *main)
        struct.pack_into("<QQ", pg, base + 96, last_pg if txnid else 1, txnid)
        return bytes(pg)
    empty = (0, 0, 0, 0, 0, 0, 0, INVALID)
    full = (0, 0, depth, n_branch, n_leaf, n_over, len(keys), root)
    with open(os.path.join(dirpath, "data.mdb"), "wb") as f:
        f.write(meta(0, 0, empty))               # the environment as created (transaction 0) ...
        f.write(meta(1, 1, full))                # ... and after the one write transaction: page 1 (txnid & 1) is current
        for pgno in range(2, next_pg[0]):
            f.write(pages[pgno])
    return {"depth": depth, "branch_pages": n_branch, "leaf_pages": n_leaf, "overflow_pages": n_over, "entries": len(keys)}
