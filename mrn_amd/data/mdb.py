"""Read-only walker of an LMDB environment file (`<leaf>/data.mdb`), for datasets written by the reference's
tools/create_lmdb_dataset.py:327-345 (keys `num-samples`, `label-%09d`, `image-%09d`) and read by data/dataset.py:44-112 through
`lmdb.open(root, readonly=True, lock=False)` + `txn.get(key)`.  The `lmdb` package is not part of the MI355X image, and a reader
needs none of its machinery (no locks, no transactions, no writes): the file is a B+tree of fixed-size pages that can be walked
straight from a memory map.

On-disk format (LMDB 0.9, data version 1, little-endian 64-bit -- what py-lmdb writes on x86-64 / aarch64 Linux):

  page header, 16 bytes   pgno u64 | pad u16 | flags u16 | lower u16, upper u16  (overflow pages: u32 page count instead)
        flags: 0x01 branch, 0x02 leaf, 0x04 overflow, 0x08 meta, 0x20 leaf2 (fixed-size keys: DUPFIXED sub-databases only)
        after the header: u16 node offsets (from the page start), (lower - 16) / 2 of them, sorted by key
  meta pages 0 and 1      header, then magic u32 = 0xBEEFC0DE | version u32 = 1 | address u64 | mapsize u64 |
        two 48-byte database records (free list, main): pad u32 (the free-list record's pad is the PAGE SIZE) | flags u16 |
        depth u16 | branch pages u64 | leaf pages u64 | overflow pages u64 | entries u64 | root pgno u64 (~0: empty) |
        then last pgno u64 | txnid u64.  The meta page with the larger txnid is the current one.
  node, 8-byte header     lo u16 | hi u16 | flags u16 | ksize u16 | key bytes | data
        leaf:   data size = lo | hi << 16; flags 0x01 (BIGDATA): the data field is the u64 page number of an overflow run whose payload
                starts 16 bytes into its first page; 0x02 / 0x04: sub-database / duplicate records (named databases, DUPSORT) --
                never present in a dataset environment, rejected here
        branch: child pgno = lo | hi << 16 | flags << 32; the key of node 0 is empty (smaller than everything)
  keys compare as byte strings (memcmp, shorter first on a tie), the default of an unnamed database.

Self-pinned: there is no LMDB implementation in the build container to generate a fixture with, so the tests exercise this walker
against files produced by tests/mdb_writer.py, written from the same format description (DESIGN.md says so).  It checks what it can
(magic, version, page flags, page numbers, bounds) and raises MdbError on anything it does not understand rather than guessing.
"""
import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
PAGEHDRSZ = 16
INVALID = (1 << 64) - 1


class MdbError(IOError):
    pass


class Environment:
    """lmdb.Environment stand-in for the read path of data/dataset.py: `env.begin(write=False)` -> object with `.get(key)`;
    usable as the context manager the reference uses (`with env.begin(write=False) as txn`)."""

    def __init__(self, path):
        self.path = os.path.join(path, "data.mdb") if os.path.isdir(path) else path
        self._f = open(self.path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            raise MdbError(f"{self.path}: too small for an LMDB environment ({size} bytes)")
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._size = size
        metas = []
        # the page size lives in the meta page itself; page 1 starts at that size, so read meta 0 first
        m0 = self._read_meta(0)
        self.page_size = m0["psize"]
        metas.append(m0)
        if self.page_size < 512 or self.page_size & (self.page_size - 1) or 2 * self.page_size > size:
            raise MdbError(f"{self.path}: implausible page size {self.page_size}")
        metas.append(self._read_meta(self.page_size))
        if metas[1]["psize"] != self.page_size:
            raise MdbError(f"{self.path}: the two meta pages disagree on the page size")
        self.meta = max(metas, key=lambda m: m["txnid"])
        main = self.meta["main"]
        if main["flags"]:                  # (REVERSEKEY / DUPSORT / INTEGERKEY ... change the key order or the record layout)
            raise MdbError(f"{self.path}: main database flags {main['flags']:#x} unsupported (plain byte-string keys only)")
        self.root = main["root"]
        self.entries = main["entries"]
        self.depth = main["depth"]

    # -- low level ---------------------------------------------------------------------------------------------------------
    def _read_meta(self, off):
        m = self._m
        pgno, _pad, flags = struct.unpack_from("<QHH", m, off)
        if not flags & P_META:
            raise MdbError(f"{self.path}: page at offset {off} is not a meta page (flags {flags:#x})")
        magic, version = struct.unpack_from("<II", m, off + PAGEHDRSZ)
        if magic != MAGIC:
            raise MdbError(f"{self.path}: bad magic {magic:#x}")
        if version != 1:
            raise MdbError(f"{self.path}: data version {version} unsupported (LMDB 0.9 writes 1)")
        dbs = []
        base = off + PAGEHDRSZ + 8 + 16
        for i in range(2):
            pad, dflags, depth, branch, leaf, over, entries, root = struct.unpack_from("<IHHQQQQQ", m, base + 48 * i)
            dbs.append({"pad": pad, "flags": dflags, "depth": depth, "branch_pages": branch, "leaf_pages": leaf, "overflow_pages": over,
                        "entries": entries, "root": root})
        last_pg, txnid = struct.unpack_from("<QQ", m, base + 96)
        return {"pgno": pgno, "psize": dbs[0]["pad"], "free": dbs[0], "main": dbs[1], "last_pg": last_pg, "txnid": txnid}

    def _page(self, pgno):
        off = pgno * self.page_size
        if pgno > self.meta["last_pg"] or off + self.page_size > self._size:
            raise MdbError(f"{self.path}: page {pgno} lies outside the file")
        got, _pad, flags, lower, upper = struct.unpack_from("<QHHHH", self._m, off)
        if got != pgno:
            raise MdbError(f"{self.path}: page {pgno} carries page number {got}")
        return off, flags, lower, upper

    def _nodes(self, off, lower):
        n = (lower - PAGEHDRSZ) // 2
        return struct.unpack_from("<%dH" % n, self._m, off + PAGEHDRSZ) if n else ()

    def _key(self, off, ptr):
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._m, off + ptr)
        k0 = off + ptr + 8
        return lo, hi, flags, ksize, self._m[k0:k0 + ksize]

    def _leaf_value(self, off, ptr):
        lo, hi, flags, ksize, _ = self._key(off, ptr)
        if flags & (F_SUBDATA | F_DUPDATA):
            raise MdbError(f"{self.path}: sub-database / duplicate records are not supported (node flags {flags:#x})")
        size = lo | (hi << 16)
        d0 = off + ptr + 8 + ksize
        if flags & F_BIGDATA:
            (opg,) = struct.unpack_from("<Q", self._m, d0)
            ooff, oflags, pages_lo, pages_hi = self._page(opg)
            if not oflags & P_OVERFLOW:
                raise MdbError(f"{self.path}: page {opg} is not an overflow page (flags {oflags:#x})")
            npages = pages_lo | (pages_hi << 16)
            if PAGEHDRSZ + size > npages * self.page_size or ooff + PAGEHDRSZ + size > self._size:
                raise MdbError(f"{self.path}: overflow run at page {opg} ({npages} pages) cannot hold {size} bytes")
            return bytes(self._m[ooff + PAGEHDRSZ:ooff + PAGEHDRSZ + size])
        if ptr + 8 + ksize + size > self.page_size:
            raise MdbError(f"{self.path}: node at {off + ptr} overruns its page")
        return bytes(self._m[d0:d0 + size])

    # -- the API data/dataset.py uses ------------------------------------------------------------------------------------------
    def get(self, key, default=None):
        """the value stored under `key` (bytes) or `default`"""
        key = bytes(key)
        if self.root == INVALID:
            return default
        pgno = self.root
        for _ in range(64):                                  # (a tree is never deeper; guards against a cycle in a corrupt file)
            off, flags, lower, _upper = self._page(pgno)
            ptrs = self._nodes(off, lower)
            if flags & P_LEAF2:
                raise MdbError(f"{self.path}: fixed-size-key leaf pages (DUPFIXED) are not supported")
            if flags & P_BRANCH:
                # last node whose key <= key; node 0's key is implicit -infinity
                lo_, hi_ = 1, len(ptrs) - 1
                idx = 0
                while lo_ <= hi_:
                    mid = (lo_ + hi_) // 2
                    if self._key(off, ptrs[mid])[4] <= key:
                        idx, lo_ = mid, mid + 1
                    else:
                        hi_ = mid - 1
                l, h, f, _ks, _k = self._key(off, ptrs[idx])
                pgno = l | (h << 16) | (f << 32)
                continue
            if not flags & P_LEAF:
                raise MdbError(f"{self.path}: page {pgno} is neither branch nor leaf (flags {flags:#x})")
            lo_, hi_ = 0, len(ptrs) - 1
            while lo_ <= hi_:
                mid = (lo_ + hi_) // 2
                k = self._key(off, ptrs[mid])[4]
                if k == key:
                    return self._leaf_value(off, ptrs[mid])
                if k < key:
                    lo_ = mid + 1
                else:
                    hi_ = mid - 1
            return default
        raise MdbError(f"{self.path}: B+tree deeper than 64 levels (corrupt file?)")

    def items(self):
        """every (key, value) in key order (a cursor walk)"""
        if self.root == INVALID:
            return
        stack = [self.root]
        while stack:
            pgno = stack.pop()
            off, flags, lower, _ = self._page(pgno)
            ptrs = self._nodes(off, lower)
            if flags & P_BRANCH:
                kids = []
                for p in ptrs:
                    l, h, f, _ks, _k = self._key(off, p)
                    kids.append(l | (h << 16) | (f << 32))
                stack.extend(reversed(kids))
            elif flags & P_LEAF and not flags & P_LEAF2:
                for p in ptrs:
                    yield bytes(self._key(off, p)[4]), self._leaf_value(off, p)
            else:
                raise MdbError(f"{self.path}: unexpected page flags {flags:#x} at page {pgno}")

    def stat(self):
        m = self.meta["main"]
        return {"psize": self.page_size, "depth": m["depth"], "branch_pages": m["branch_pages"], "leaf_pages": m["leaf_pages"],
                "overflow_pages": m["overflow_pages"], "entries": m["entries"]}

    def begin(self, write=False, **_):
        if write:
            raise MdbError("read-only environment")
        return _Txn(self)

    def close(self):
        if self._m is not None:
            self._m.close()
            self._f.close()
            self._m = None

    def __bool__(self):
        return True


class _Txn:
    def __init__(self, env):
        self.env = env

    def get(self, key, default=None):
        return self.env.get(key, default)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def open_environment(root):
    """what LmdbDataset opens: the `lmdb` package when it is installed (lmdb.open(root, readonly=True, lock=False, ...), as the reference
    does), this module's walker otherwise"""
    try:
        import lmdb
    except ImportError:
        return Environment(root)
    return lmdb.open(root, max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)
