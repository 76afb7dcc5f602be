"""ctypes front-end of oracle/libcobs_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under cobs_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcobs_oracle.so")

ERR_NAMES = {
    0: "OK", 1: "ERR_OPEN", 2: "ERR_FORMAT", 3: "ERR_QUERY_TOO_SHORT",
    4: "ERR_INVALID_BASE", 5: "ERR_QUERY_TOO_LONG", 6: "ERR_GEOMETRY", 7: "ERR_ARG",
}


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


def build(force=False, native=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "cobs_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(os.path.join(_HERE, "cobs_oracle.h"))):
        return _LIB_PATH
    march = "native" if native else "x86-64-v2"
    cmd = ["gcc", "-O3", "-march=" + march, "-msse2", "-fPIC", "-std=gnu11", "-shared",
           "-o", _LIB_PATH, src, "-lpthread", "-lm"]
    subprocess.check_call(cmd, cwd=_HERE)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u64, u32, sz, vp, cp = C.c_uint64, C.c_uint32, C.c_size_t, C.c_void_p, C.c_char_p
    L.oracle_xxh64.restype = u64
    L.oracle_xxh64.argtypes = [cp, sz, u64]
    L.oracle_canonicalize_kmer.restype = C.c_int
    L.oracle_canonicalize_kmer.argtypes = [cp, vp, sz]
    L.oracle_term_hashes.restype = None
    L.oracle_term_hashes.argtypes = [cp, sz, u32, C.c_int, u64, vp, vp]
    L.oracle_random_sequence.restype = None
    L.oracle_random_sequence.argtypes = [vp, sz, u64]
    L.oracle_mt19937_sequence.restype = None
    L.oracle_mt19937_sequence.argtypes = [vp, sz, vp, u32, C.c_int]
    L.oracle_open.restype = C.c_int
    L.oracle_open.argtypes = [cp, C.POINTER(vp)]
    L.oracle_from_memory.restype = C.c_int
    L.oracle_from_memory.argtypes = [C.c_int, u32, C.c_uint8, u64, u64, u32, vp, u32, vp, C.POINTER(vp)]
    L.oracle_synthetic.restype = C.c_int
    L.oracle_synthetic.argtypes = [C.c_int, u32, C.c_uint8, u64, u64, u32, vp, u32, u64, C.POINTER(vp)]
    L.oracle_close.restype = None
    L.oracle_close.argtypes = [vp]
    for name, rt in [("term_size", u32), ("canonicalize", u32), ("num_hashes", u64),
                     ("page_size", u64), ("row_size", u64), ("counts_size", u64),
                     ("num_pages", u32), ("num_docs", u32), ("data_offset", u64)]:
        f = getattr(L, "oracle_" + name)
        f.restype = rt
        f.argtypes = [vp]
    L.oracle_signature_size.restype = u64
    L.oracle_signature_size.argtypes = [vp, u32]
    L.oracle_doc_name.restype = cp
    L.oracle_doc_name.argtypes = [vp, u32]
    L.oracle_plant.restype = C.c_int
    L.oracle_plant.argtypes = [vp, C.c_char_p, C.c_size_t, vp, vp, C.c_size_t, u64]
    L.oracle_synth_fill.restype = None
    L.oracle_synth_fill.argtypes = [C.c_int, u64, u64, u32, u32, u32, u64, u64, u64, vp]
    L.oracle_counts.restype = C.c_int
    L.oracle_counts.argtypes = [vp, cp, sz, C.c_int, vp, C.POINTER(C.c_int)]
    L.oracle_search.restype = C.c_int
    L.oracle_search.argtypes = [vp, sz, cp, sz, C.c_double, sz, C.c_int, vp, vp, vp, sz, C.POINTER(sz)]
    L.oracle_search_many.restype = sz
    L.oracle_search_many.argtypes = [vp, sz, cp, vp, sz, C.c_double, sz, C.c_int, C.c_double,
                                     C.POINTER(C.c_double), C.POINTER(u64)]
    L.oracle_timers.restype = None
    L.oracle_timers.argtypes = [vp, C.c_int]
    L.oracle_last_error.restype = cp
    L.oracle_last_error.argtypes = []
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().oracle_last_error().decode("utf-8", "replace"))


def xxh64(data: bytes, seed: int = 0) -> int:
    return int(lib().oracle_xxh64(data, len(data), seed))


def canonicalize_kmer(kmer: bytes):
    """-> (canonical bytes incl. NULs for invalid characters, good flag)"""
    out = C.create_string_buffer(len(kmer))
    good = lib().oracle_canonicalize_kmer(kmer, out, len(kmer))
    return out.raw, bool(good)


def term_hashes(seq: bytes, k: int, canonicalize: int, num_hashes: int):
    """-> (uint64 [T, H] full 64-bit hashes, bool [T] 'all bases valid')"""
    T = max(len(seq) - k + 1, 0)
    out = np.zeros((T, num_hashes), dtype=np.uint64)
    good = np.ones(T, dtype=np.uint8)
    if T:
        lib().oracle_term_hashes(seq, len(seq), k, canonicalize, num_hashes, out.ctypes.data,
                                 good.ctypes.data)
    return out, good.astype(bool)


def random_sequence(size: int, seed: int) -> bytes:
    """cobs::random_sequence (minstd_rand0 % 4 -> ACGT), reference util/misc.cpp:32-35"""
    out = C.create_string_buffer(size)
    lib().oracle_random_sequence(out, size, seed)
    return out.raw


class Mt19937Queries:
    """benchmark-fpr query stream: one std::mt19937(seed) shared by all queries
    (reference src/cobs.cpp:709-720)."""

    def __init__(self, seed: int):
        self._state = (C.c_uint32 * 625)()
        self._seed = seed
        self._first = 1

    def next(self, size: int) -> bytes:
        out = C.create_string_buffer(size)
        lib().oracle_mt19937_sequence(out, size, self._state, self._seed, self._first)
        self._first = 0
        return out.raw


def synth_row(kind, seed, page_size, num_pages, num_docs, page, row, nbytes, byte_begin=0):
    out = np.zeros(nbytes, dtype=np.uint8)
    lib().oracle_synth_fill(kind, seed, page_size, num_pages, num_docs, page, row, byte_begin,
                            nbytes, out.ctypes.data)
    return out


class Index:
    """One opened index (classic or compact): file, caller memory or procedural."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    @classmethod
    def open(cls, path):
        h = C.c_void_p()
        _check(lib().oracle_open(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def from_memory(cls, kind, term_size, canonicalize, num_hashes, page_size, signature_sizes,
                    num_docs, pages):
        """pages: list of C-contiguous uint8 arrays, one per sub-index"""
        sigs = np.ascontiguousarray(signature_sizes, dtype=np.uint64)
        ptrs = (C.c_void_p * len(pages))(*[p.ctypes.data for p in pages])
        h = C.c_void_p()
        _check(lib().oracle_from_memory(kind, term_size, canonicalize, num_hashes, page_size,
                                        len(pages), sigs.ctypes.data, num_docs, ptrs, C.byref(h)))
        return cls(h, keep=(sigs, ptrs, pages))

    @classmethod
    def synthetic(cls, kind, term_size, canonicalize, num_hashes, page_size, signature_sizes,
                  num_docs, seed):
        sigs = np.ascontiguousarray(signature_sizes, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().oracle_synthetic(kind, term_size, canonicalize, num_hashes, page_size,
                                      len(sigs), sigs.ctypes.data, num_docs, seed, C.byref(h)))
        return cls(h, keep=(sigs,))

    def plant(self, text, docs, keep_permille=1000, salt=0):
        """procedural index only: documents `docs` additionally contain the terms of `text` (the share keep_permille[i] / 1000
        of them) -- same rule as cobs_amd.Search.plant"""
        if isinstance(text, str):
            text = text.encode()
        docs = np.ascontiguousarray(docs, dtype=np.uint32)
        keep = np.ascontiguousarray(np.broadcast_to(np.asarray(keep_permille, dtype=np.uint32), docs.shape))
        _check(lib().oracle_plant(self._h, text, len(text), docs.ctypes.data, keep.ctypes.data, len(docs), int(salt)))

    def close(self):
        if self._h:
            lib().oracle_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    term_size = property(lambda s: int(lib().oracle_term_size(s._h)))
    canonicalize = property(lambda s: int(lib().oracle_canonicalize(s._h)))
    num_hashes = property(lambda s: int(lib().oracle_num_hashes(s._h)))
    page_size = property(lambda s: int(lib().oracle_page_size(s._h)))
    row_size = property(lambda s: int(lib().oracle_row_size(s._h)))
    counts_size = property(lambda s: int(lib().oracle_counts_size(s._h)))
    num_pages = property(lambda s: int(lib().oracle_num_pages(s._h)))
    num_docs = property(lambda s: int(lib().oracle_num_docs(s._h)))
    data_offset = property(lambda s: int(lib().oracle_data_offset(s._h)))

    def signature_size(self, page=0):
        return int(lib().oracle_signature_size(self._h, page))

    def doc_name(self, doc):
        return lib().oracle_doc_name(self._h, doc).decode()

    def counts(self, query: bytes, threads=1, want_width=False):
        out = np.zeros(self.counts_size, dtype=np.uint32)
        w = C.c_int(0)
        _check(lib().oracle_counts(self._h, query, len(query), threads, out.ctypes.data, C.byref(w)))
        return (out, w.value) if want_width else out


def search(indexes, query: bytes, threshold=0.0, num_results=0, threads=1):
    """cobs::ClassicSearch::search -> list of (index_no, doc_id, doc_name, score)"""
    if isinstance(indexes, Index):
        indexes = [indexes]
    cap = sum(ix.counts_size for ix in indexes)
    hs = (C.c_void_p * len(indexes))(*[ix._h for ix in indexes])
    oi = np.zeros(max(cap, 1), dtype=np.uint32)
    od = np.zeros(max(cap, 1), dtype=np.uint32)
    os_ = np.zeros(max(cap, 1), dtype=np.uint32)
    n = C.c_size_t(0)
    _check(lib().oracle_search(hs, len(indexes), query, len(query), float(threshold), int(num_results),
                               threads, oi.ctypes.data, od.ctypes.data, os_.ctypes.data, cap, C.byref(n)))
    return [(int(oi[i]), int(od[i]), indexes[int(oi[i])].doc_name(int(od[i])), int(os_[i]))
            for i in range(n.value)]


def search_arrays(indexes, query: bytes, threshold=0.0, num_results=0, threads=1):
    """cobs::ClassicSearch::search, columnar: -> (index_no, doc_id, score) uint32 arrays in result order
    (no per-result Python objects: ranking 100 000 documents per query)"""
    if isinstance(indexes, Index):
        indexes = [indexes]
    cap = sum(ix.counts_size for ix in indexes)
    hs = (C.c_void_p * len(indexes))(*[ix._h for ix in indexes])
    oi = np.zeros(max(cap, 1), dtype=np.uint32)
    od = np.zeros(max(cap, 1), dtype=np.uint32)
    os_ = np.zeros(max(cap, 1), dtype=np.uint32)
    n = C.c_size_t(0)
    _check(lib().oracle_search(hs, len(indexes), query, len(query), float(threshold), int(num_results),
                               threads, oi.ctypes.data, od.ctypes.data, os_.ctypes.data, cap, C.byref(n)))
    return oi[:n.value], od[:n.value], os_[:n.value]


def search_many(indexes, queries, threshold=0.0, num_results=0, threads=1, seconds=5.0):
    """time ClassicSearch::search over `queries` inside C -> (queries done, seconds)"""
    if isinstance(indexes, Index):
        indexes = [indexes]
    hs = (C.c_void_p * len(indexes))(*[ix._h for ix in indexes])
    text = b"".join(queries)
    offs = np.zeros(len(queries) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    el, ck = C.c_double(0), C.c_uint64(0)
    n = lib().oracle_search_many(hs, len(indexes), text, offs.ctypes.data, len(queries), float(threshold),
                                 int(num_results), threads, float(seconds), C.byref(el), C.byref(ck))
    return int(n), el.value


def search_many_parallel(indexes, queries, workers, threshold=-1.0, num_results=0, seconds=5.0):
    """`workers` host threads, each running search_many (one thread per query) over its own
    share of the queries at the same time -- the throughput a caller gets from the reference's
    algorithm by serving independent queries on all cores.  -> (queries done, seconds)"""
    import threading
    workers = max(1, min(int(workers), len(queries)))
    res = [None] * workers

    def work(i):
        res[i] = search_many(indexes, queries[i::workers], threshold, num_results, threads=1, seconds=seconds)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return sum(r[0] for r in res), max(r[1] for r in res)


def timers(reset=False):
    t = (C.c_double * 5)()
    lib().oracle_timers(t, 1 if reset else 0)
    return dict(zip(["hashes", "io", "and", "add", "sort"], list(t)))
