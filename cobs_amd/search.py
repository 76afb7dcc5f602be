"""Host-side mirror of the reference's Python operator API for the query path.

Reference: python/module.cpp:351-386 --
    cobs_index.Search(path).search(query, threshold=0.0, num_results=0)
        -> list[SearchResult(doc_name: str, score: int)]
Same names, argument meaning and defaults.  Everything is computed by
libcobs_gpu.so on the GPU; this file only marshals arguments.  Where the
reference terminates the process on bad input (short query, non-ACGT base,
unreadable index: classic_search.cpp:431-433, :93-96, :61-63) this mirror raises
CobsGpuError carrying the C-ABI status instead.
"""
import collections.abc
import ctypes as C
import os
import sys

import numpy as np

from . import _capi
from ._capi import CobsGpuError, Hit, IndexInfo, Options, Synth, check


class SearchResult:
    """cobs::SearchResult (query/search.hpp:17-27) as exposed by python/module.cpp:351-362."""
    __slots__ = ("doc_name", "score")

    def __init__(self, doc_name="", score=0):
        self.doc_name = doc_name
        self.score = score

    def __repr__(self):
        return "SearchResult(doc_name=%r, score=%d)" % (self.doc_name, self.score)

    def __eq__(self, other):
        return (isinstance(other, SearchResult) and self.doc_name == other.doc_name
                and self.score == other.score)


class ResultList(collections.abc.Sequence):
    """The result of one query when it is long (the default call returns every document of the index): behaves
    like the reference's list of SearchResult -- len(), indexing, slicing, iteration, equality with a list -- but
    creates a SearchResult only when one is looked at.  100 000 results cost 0.1 ms instead of the ~100 ms a list
    of 100 000 Python objects takes to build; .doc / .score / .file_no are the columns (numpy)."""
    __slots__ = ("_search", "file_no", "doc", "score")

    def __init__(self, search, seg):
        self._search = search
        self.file_no, self.doc, self.score = seg["file_no"], seg["doc"], seg["score"]

    def __len__(self):
        return len(self.doc)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return SearchResult(self._search._names(int(self.file_no[i]))[int(self.doc[i])], int(self.score[i]))

    def __eq__(self, other):
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))

    def __repr__(self):
        return "ResultList(%d results, first %r)" % (len(self), self[0] if len(self) else None)


def _split_segments(hits, offs, nq, threshold, num_results, search):
    """per-query result lists of cobs_gpu_sharded_search_batch_split on THIS rank.  In the all-documents call the
    ranks share the ranking: a rank fills only the places of the queries it owns and writes only hit_offsets[q + 1]
    of those (hit_offsets[q], the start of its first owned query, belongs to the previous owner -- another
    process's array unless one process holds all ranks).  Every query yields exactly one result per real document
    there, so a query's segment is taken from its END; queries of other owners come back empty.  Any other call
    (threshold > 0 or a limit) returns every query on every rank with complete offsets."""
    topk = num_results if num_results < search.total_counts else 0
    if threshold <= 0.0 and topk == 0:
        per_query = sum(int(search.info(f).num_docs) for f in range(search.num_files))
        return [hits[int(offs[q + 1]) - per_query:int(offs[q + 1])].tolist() if int(offs[q + 1]) >= per_query and offs[q + 1] > 0 else []
                for q in range(nq)]
    rows = hits[:int(offs[nq])].tolist()
    return [rows[int(offs[q]):int(offs[q + 1])] for q in range(nq)]


def _as_bytes(q):
    return q.encode("latin-1") if isinstance(q, str) else bytes(q)


def _options(device, shard_rank, shard_count, hbm_budget=0, shard_mode=0):
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.device = device
    o.shard_rank = shard_rank
    o.shard_count = shard_count
    o.shard_mode = shard_mode
    o.hbm_budget_bytes = int(hbm_budget)
    return o


class Search:
    """cobs_index.Search: open one or several index files (classic or compact,
    auto-detected per file) and query them on the GPU."""

    def __init__(self, path, device=-1, shard_rank=0, shard_count=1, hbm_budget=0, shard_mode=0, _handle=None):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
            return
        paths = [path] if isinstance(path, (str, bytes, os.PathLike)) else list(path)
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        opts = _options(device, shard_rank, shard_count, hbm_budget, shard_mode)
        check(self._lib.cobs_gpu_open(arr, len(paths), C.byref(opts), C.byref(self._h)))

    @classmethod
    def synthetic(cls, kind, signature_sizes, num_docs, page_size=0, term_size=31, canonicalize=1,
                  num_hashes=1, seed=1, device=-1, shard_rank=0, shard_count=1, hbm_budget=0, shard_mode=0):
        """Procedural index generated directly in HBM (benchmark / large parity runs)."""
        lib = _capi.load()
        sigs = (C.c_uint64 * len(signature_sizes))(*[int(s) for s in signature_sizes])
        d = Synth()
        d.kind = 1 if kind in (1, "compact") else 0
        d.term_size, d.canonicalize, d.num_pages = term_size, canonicalize, len(signature_sizes)
        d.num_hashes, d.page_size, d.num_docs, d.seed = num_hashes, page_size, num_docs, seed
        d.signature_sizes = C.cast(sigs, C.POINTER(C.c_uint64))
        h = C.c_void_p()
        opts = _options(device, shard_rank, shard_count, hbm_budget, shard_mode)
        check(lib.cobs_gpu_open_synthetic(C.byref(d), C.byref(opts), C.byref(h)))
        return cls(None, _handle=h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cobs_gpu_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- geometry ------------------------------------------------------------
    @property
    def num_files(self):
        return int(self._lib.cobs_gpu_num_files(self._h))

    def info(self, file_no=0):
        i = IndexInfo()
        check(self._lib.cobs_gpu_info(self._h, file_no, C.byref(i)))
        return i

    @property
    def total_counts(self):
        return int(self._lib.cobs_gpu_total_counts(self._h))

    @property
    def local_counts(self):
        return int(self._lib.cobs_gpu_local_counts(self._h))

    def doc_name(self, file_no, doc):
        return self._lib.cobs_gpu_doc_name(self._h, file_no, doc).decode()

    def _names(self, file_no):
        """document names of one file, fetched once (search() of a 100k-document index names
        100k results per query)"""
        cache = self.__dict__.setdefault("_name_cache", {})
        if file_no not in cache:
            fn, h = self._lib.cobs_gpu_doc_name, self._h
            cache[file_no] = [fn(h, file_no, d).decode() for d in range(int(self.info(file_no).num_docs))]
        return cache[file_no]

    def signature_size(self, file_no=0, page=0):
        return int(self._lib.cobs_gpu_signature_size(self._h, file_no, page))

    def read_row(self, file_no, page, row, nbytes):
        out = np.zeros(nbytes, dtype=np.uint8)
        check(self._lib.cobs_gpu_read_row(self._h, file_no, page, row, out.ctypes.data, nbytes))
        return out

    def page_columns(self, file_no, page):
        """-> (col0, ncols): the row bytes of sub-index `page` this shard holds"""
        c0, n = C.c_uint64(0), C.c_uint64(0)
        check(self._lib.cobs_gpu_page_columns(self._h, file_no, page, C.byref(c0), C.byref(n)))
        return int(c0.value), int(n.value)

    def set_tuning(self, key, value):
        """per-handle tuning hook of the scan launch (see cobs_gpu_set_tuning)"""
        check(self._lib.cobs_gpu_set_tuning(self._h, key.encode(), int(value)))

    def read_rows(self, file_no, page, row0, nrows, out=None):
        """bulk D2H of whole rows of one held sub-index -> uint8 [nrows, held row bytes]"""
        width = self.page_columns(file_no, page)[1]
        if out is None:
            out = np.empty((nrows, width), dtype=np.uint8)
        check(self._lib.cobs_gpu_read_rows(self._h, file_no, page, row0, nrows, out.ctypes.data, width))
        return out

    def plant(self, text, docs, keep_permille=1000, salt=0, file_no=0):
        """True positives for a resident index (cobs_gpu_plant): documents `docs` additionally contain the terms of `text`,
        document docs[i] the share keep_permille[i] / 1000 of them (one value = the same for all)."""
        if isinstance(text, str):
            text = text.encode()
        docs = np.ascontiguousarray(docs, dtype=np.uint32)
        keep = np.ascontiguousarray(np.broadcast_to(np.asarray(keep_permille, dtype=np.uint32), docs.shape))
        u32p = C.POINTER(C.c_uint32)
        check(self._lib.cobs_gpu_plant(self._h, file_no, text, len(text), docs.ctypes.data_as(u32p),
                                       keep.ctypes.data_as(u32p), len(docs), int(salt)))

    # -- queries -------------------------------------------------------------
    def search(self, query, threshold=0.0, num_results=0):
        """Same contract as cobs_index.Search.search (python/module.cpp:372-386)."""
        if type(self)._search_batch_call is not Search._search_batch_call:
            return self.search_batch([query], threshold, num_results)[0]
        # one query: cobs_gpu_search straight into a reused result buffer (no numpy marshalling of a batch of one)
        q = query if type(query) is bytes else _as_bytes(query)
        total = self.total_counts
        if num_results > 0:
            cap = min(num_results, total)
        elif threshold <= 0:
            cap = total
        else:
            cap = 1024
        n = C.c_size_t(0)
        while True:
            hits = self._result_buffer(max(cap, 1))
            st = self._lib.cobs_gpu_search(self._h, q, len(q), float(threshold), int(num_results),
                                           C.cast(hits.ctypes.data, C.POINTER(Hit)), len(hits), C.byref(n))
            if st == _capi.ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            check(st)
            break
        seg = hits[:n.value]
        if n.value <= 64:
            return [SearchResult(self.doc_name(f, d), s) for (f, d, s) in seg.tolist()]
        return ResultList(self, seg)

    HIT_DTYPE = np.dtype([("file_no", "<u4"), ("doc", "<u4"), ("score", "<u4")])
    reuse_result_buffers = True     # see _result_buffer

    def search_arrays(self, queries, threshold=0.0, num_results=0, out=None):
        """-> (offsets uint64 [nq + 1], hits structured array of (file_no, doc, score)):
        the hits of query i are hits[offsets[i]:offsets[i + 1]], in result order.  The columnar form of
        search(): no per-result Python objects (the default call returns every document of the index)."""
        qs = [q if type(q) is bytes else _as_bytes(q) for q in queries]
        offsets = np.zeros(len(qs) + 1, dtype=np.uint64)
        np.cumsum(np.fromiter(map(len, qs), dtype=np.uint64, count=len(qs)), out=offsets[1:])
        return self.search_packed(b"".join(qs), offsets, threshold, num_results, out=out)

    @staticmethod
    def _packed_args(text, offsets):
        """-> (char** queries, size_t* lens, nq, what keeps them alive) for queries packed back to back"""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nq = len(offsets) - 1
        if nq < 0:
            raise ValueError("offsets needs nq + 1 entries")
        if isinstance(text, np.ndarray):
            text = np.ascontiguousarray(text, dtype=np.uint8)
            base, size = text.ctypes.data, text.size
        else:
            text = bytes(text) if not isinstance(text, bytes) else text
            base, size = C.cast(C.c_char_p(text), C.c_void_p).value or 0, len(text)
        if nq and (int(offsets[-1]) > size or np.any(offsets[1:] < offsets[:-1])):
            raise ValueError("offsets are not ascending positions inside text")
        ptrs_np = (offsets[:-1] + np.uint64(base)) if nq else np.zeros(1, dtype=np.uint64)
        lens_np = (offsets[1:] - offsets[:-1]) if nq else np.zeros(1, dtype=np.uint64)
        ptrs_np = np.ascontiguousarray(ptrs_np)
        lens_np = np.ascontiguousarray(lens_np)
        arr = C.cast(ptrs_np.ctypes.data, C.POINTER(C.c_char_p))
        lens = C.cast(lens_np.ctypes.data, C.POINTER(C.c_size_t))
        return arr, lens, nq, (text, ptrs_np, lens_np)

    def search_packed(self, text, offsets, threshold=0.0, num_results=0, out=None):
        """The same for queries packed back to back: query i is text[offsets[i]:offsets[i + 1]]
        (text: bytes or a uint8 array, e.g. the sequence lines of a FASTQ block; offsets: nq + 1
        integers).  No per-query Python objects: the pointer array is built with numpy.
        out: an optional HIT_DTYPE array to receive the hits (a caller that repeats large calls keeps
        one buffer instead of page-faulting a fresh one per call); it is used when large enough."""
        arr, lens, nq, _keep = self._packed_args(text, offsets)
        if num_results > 0:
            cap = nq * min(num_results, self.total_counts)
        elif threshold <= 0:
            cap = nq * self.total_counts          # every document is a result
        elif type(self)._search_batch_call is not Search._search_batch_call:
            cap = 16 * nq + 1024                  # (several GPUs behind one handle: grown on demand, the call repeated)
        else:
            # the number of hits is only known when the search has run (ERR_CAPACITY reports it, and the WHOLE call would
            # be made a second time): the view call collects them in the arena the library grows as the passes come
            # home; the finished lists are copied out of it once (12 bytes per hit)
            hp, op, bad = C.POINTER(Hit)(), C.POINTER(C.c_size_t)(), C.c_size_t(0)
            check(self._lib.cobs_gpu_search_batch_view(self._h, arr, lens, nq, float(threshold), 0,
                                                       C.byref(hp), C.byref(op), C.byref(bad)))
            offs = np.array(np.ctypeslib.as_array(op, shape=(nq + 1,)).view(np.uint64))
            n = int(offs[nq])
            if out is not None and out.dtype == self.HIT_DTYPE and out.flags.c_contiguous and out.size >= n:
                hits = out
            else:
                hits = self._result_buffer(max(n, 1))
            if n:
                C.memmove(hits.ctypes.data, hp, n * 12)
            return offs, hits[:n]
        cap = max(1, cap)
        offs = np.zeros(nq + 1, dtype=np.uint64)
        bad = C.c_size_t(0)
        while True:
            if out is not None and out.dtype == self.HIT_DTYPE and out.flags.c_contiguous and out.size >= cap:
                hits, cap = out, out.size
            else:
                hits = self._result_buffer(cap)
            st = self._search_batch_call(
                arr, lens, nq, float(threshold), int(num_results),
                C.cast(hits.ctypes.data, C.POINTER(Hit)), cap,
                C.cast(offs.ctypes.data, C.POINTER(C.c_size_t)), C.byref(bad))
            if st == _capi.ERR_CAPACITY and int(offs[nq]) > cap:
                cap = int(offs[nq])
                continue
            check(st)
            break
        return offs, hits[:int(offs[nq])]

    def _search_batch_call(self, *args):
        return self._lib.cobs_gpu_search_batch(self._h, *args)

    def _result_buffer(self, cap):
        """a HIT_DTYPE array of `cap` records for the results of a call.  A buffer of an EARLIER call that nothing
        references any more (neither the array that call returned nor a slice or ResultList made from it: they all keep
        the raw buffer alive through .base) is used again -- its pages are there, where a fresh 307 MB array (the default
        call of 256 queries x 100 000 documents) costs 75 000 first-touch page faults, 5.6 ms against 3.4.  A caller that
        drops the results of one call before it makes the next never allocates; one that keeps them gets fresh memory."""
        need = cap * self.HIT_DTYPE.itemsize
        pool = self.__dict__.setdefault("_result_pool", [])
        # "Nothing references it" is read off CPython's reference count of the raw buffer: exact there for every numpy
        # object made from the results (views keep the raw buffer through .base); a raw address taken from a result
        # (.ctypes.data) dangles once the result is dropped whether or not the memory is used again.  An interpreter
        # without reference counts gets a fresh buffer per call, and Search.reuse_result_buffers = False turns the
        # reuse off for a caller that wants that anyway [ADVICE r5].
        if not (self.reuse_result_buffers and sys.implementation.name == "cpython"):
            return np.empty(max(need, 1), dtype=np.uint8)[:need].view(self.HIT_DTYPE)
        for i in range(len(pool)):
            # references: the pool's, and getrefcount's argument
            if pool[i].nbytes >= need and sys.getrefcount(pool[i]) == 2:
                return pool[i][:need].view(self.HIT_DTYPE)
        raw = np.empty(max(need, 1), dtype=np.uint8)
        if need >= (1 << 20):
            # keep a few large buffers (small results are cheap to allocate): free ones first to go
            if len(pool) >= 3:
                free = [i for i in range(len(pool)) if sys.getrefcount(pool[i]) == 2]
                del pool[free[0] if free else 0]
            pool.append(raw)
        return raw[:need].view(self.HIT_DTYPE)

    def search_view(self, queries, threshold=0.0, num_results=0):
        """cobs_gpu_search_batch_view: the results stay in the arena the LIBRARY keeps on this handle.
        -> (offsets uint64 [nq + 1], hits HIT_DTYPE array) -- both views of that arena, valid until the next search call on
        this handle (copy what has to live longer)."""
        qs = [q if type(q) is bytes else _as_bytes(q) for q in queries]
        nq = len(qs)
        arr = (C.c_char_p * max(nq, 1))(*qs)
        lens = (C.c_size_t * max(nq, 1))(*[len(q) for q in qs])
        hp, op, bad = C.POINTER(Hit)(), C.POINTER(C.c_size_t)(), C.c_size_t(0)
        check(self._lib.cobs_gpu_search_batch_view(self._h, arr, lens, nq, float(threshold), int(num_results),
                                                   C.byref(hp), C.byref(op), C.byref(bad)))
        offs = np.ctypeslib.as_array(op, shape=(nq + 1,)).view(np.uint64)
        n = int(offs[nq])
        if n == 0:
            return offs, np.zeros(0, dtype=self.HIT_DTYPE)
        raw = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(n * 12,))
        return offs, raw.view(self.HIT_DTYPE)

    def sharded_search_arrays(self, comm, queries, threshold=0.0, num_results=0, split=False):
        """cobs_gpu_sharded_search_batch[_split]: collective over `comm` (every rank, same queries)
        -> (offsets uint64 [nq + 1], hits HIT_DTYPE array), the GLOBAL results in result order (split: see below).
        queries: a list, or a (text, offsets) pair of queries packed back to back."""
        if isinstance(queries, tuple):         # (text, offsets): queries packed back to back, as search_packed takes them
            arr, lens, nq, _keep = self._packed_args(*queries)
        else:
            qs = [q if type(q) is bytes else _as_bytes(q) for q in queries]
            nq = len(qs)
            arr = (C.c_char_p * max(nq, 1))(*qs)
            lens = (C.c_size_t * max(nq, 1))(*[len(q) for q in qs])
        if num_results > 0:
            cap = nq * min(num_results, self.total_counts)
        elif threshold <= 0:
            cap = nq * self.total_counts
        else:
            # (a buffer that proves too small costs a second, collective, run of the whole search: sized by the hits per
            # query of the earlier thresholded calls on this handle -- the same on every rank, as the results are)
            cap = max(16 * nq, int((0.0 if split else self.__dict__.get("_sharded_hits_per_query", 0.0)) * nq * 1.25)) + 1024
        cap = max(1, cap)
        offs = np.zeros(nq + 1, dtype=np.uint64)
        bad = C.c_size_t(0)
        fn = self._lib.cobs_gpu_sharded_search_batch_split if split else self._lib.cobs_gpu_sharded_search_batch
        while True:
            hits = self._result_buffer(cap)        # (a buffer of an earlier call that nothing references any more: its pages are there)
            offs[:] = 0
            st = fn(self._h, comm._h, arr, lens, nq, float(threshold), int(num_results),
                    C.cast(hits.ctypes.data, C.POINTER(Hit)), cap,
                    C.cast(offs.ctypes.data, C.POINTER(C.c_size_t)), C.byref(bad))
            if st == _capi.ERR_CAPACITY and int(offs[nq]) > cap:
                cap = int(offs[nq])
                continue
            check(st)
            break
        if threshold > 0 and num_results == 0 and nq and not split:
            self._sharded_hits_per_query = max(int(offs[nq]) / nq, 0.95 * self.__dict__.get("_sharded_hits_per_query", 0.0))
        return offs, hits[:int(offs[nq])] if not split else hits

    def sharded_search_hits(self, comm, queries, threshold=0.0, num_results=0, split=False):
        """-> per query the GLOBAL list of (file_no, doc, score) in result order.
        split=True: cobs_gpu_sharded_search_batch_split -- for the all-documents search the ranks share the
        ranking and this rank's return holds the results of the queries it owns (empty lists for the others)."""
        offs, hits = self.sharded_search_arrays(comm, queries, threshold, num_results, split)
        nq = len(queries)
        if split:
            return _split_segments(hits, offs, nq, threshold, num_results, self)
        rows = hits[:int(offs[nq])].tolist()
        return [rows[int(offs[q]):int(offs[q + 1])] for q in range(nq)]

    def search_hits(self, queries, threshold=0.0, num_results=0):
        """-> per query: list of (file_no, doc, score) in result order."""
        offs, hits = self.search_arrays(queries, threshold, num_results)
        rows = hits.tolist()
        return [rows[int(offs[q]):int(offs[q + 1])] for q in range(len(queries))]

    def search_batch(self, queries, threshold=0.0, num_results=0):
        offs, hits = self.search_arrays(queries, threshold, num_results)
        out = []
        for q in range(len(queries)):
            seg = hits[int(offs[q]):int(offs[q + 1])]
            if len(seg) <= 64:
                out.append([SearchResult(self.doc_name(f, d), s) for (f, d, s) in seg.tolist()])
                continue
            # many results (the default call ranks every document): a lazy sequence over the columns
            out.append(ResultList(self, seg))
        return out

    def counts(self, query):
        """Raw per-document counts (incl. padding slots) over all files, uint32."""
        q = _as_bytes(query)
        out = np.zeros(self.total_counts, dtype=np.uint32)
        check(self._lib.cobs_gpu_counts(self._h, q, len(q), out.ctypes.data, out.size))
        return out

    @property
    def graph_replays(self):
        """small host-API passes served by a captured hipGraph so far"""
        return int(self._lib.cobs_gpu_graph_replays(self._h))

    @property
    def host_passes(self):
        """device passes the host-buffer calls have launched on this handle so far"""
        return int(self._lib.cobs_gpu_host_passes(self._h))

    def stream_counters(self):
        """out-of-core handles: (chunks fetched row by row, chunks copied whole) over all passes so far"""
        c = (C.c_uint64 * 2)()
        check(self._lib.cobs_gpu_stream_counters(self._h, C.byref(c)))
        return int(c[0]), int(c[1])

    def stream_traffic(self):
        """... and (bytes of looked-up rows fetched, bytes of whole chunks copied): what the passes asked of PCIe"""
        c = (C.c_uint64 * 4)()
        check(self._lib.cobs_gpu_stream_traffic(self._h, C.byref(c)))
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])

    def stream_plan(self):
        """out-of-core handles: (bytes of one stream buffer, bytes of the streamed files kept resident beside the buffers,
        row bytes a whole-chunk pass moves over PCIe, streamed chunks)"""
        c = (C.c_uint64 * 4)()
        check(self._lib.cobs_gpu_stream_plan(self._h, C.byref(c)))
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])

    def timers(self, reset=False):
        t = (C.c_double * 5)()
        check(self._lib.cobs_gpu_timers(self._h, C.byref(t), 1 if reset else 0))
        return dict(zip(["hashes", "h2d", "scan", "d2h", "rank"], list(t)))


class MultiSearch(Search):
    """The same Search over SEVERAL GPUs of one node from ONE process (cobs_gpu_multi_*): the
    index is sharded by sub-index block over `devices`, the library runs one worker thread per
    device and the exchange over RCCL; search / search_batch / search_arrays / search_packed
    return exactly what Search returns on one GPU."""

    def __init__(self, path, devices, hbm_budget=0, shard_mode=0):
        self._lib = _capi.load()
        self._m = C.c_void_p()
        self._h = C.c_void_p()
        paths = [path] if isinstance(path, (str, bytes, os.PathLike)) else list(path)
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        devs = (C.c_int * max(1, len(devices)))(*[int(d) for d in devices])
        opts = _options(-1, 0, 1, hbm_budget, shard_mode)
        check(self._lib.cobs_gpu_multi_open(arr, len(paths), devs, len(devices), C.byref(opts), C.byref(self._m)))
        # rank 0's shard handle answers the geometry / name calls of the base class (not owned)
        self._h = C.c_void_p(self._lib.cobs_gpu_multi_index(self._m, 0))

    def close(self):
        if getattr(self, "_m", None):
            self._lib.cobs_gpu_multi_close(self._m)
            self._m = C.c_void_p()
            self._h = C.c_void_p()

    @property
    def comm_size(self):
        return int(self._lib.cobs_gpu_multi_size(self._m))

    def shard(self, rank):
        """geometry view of one rank's shard (a borrowed handle: valid until close())"""
        h = self._lib.cobs_gpu_multi_index(self._m, rank)
        if not h:
            raise IndexError(rank)
        v = Search(None, _handle=C.c_void_p(h))
        v.close = lambda: None          # the multi handle owns it
        return v

    def search_view(self, queries, threshold=0.0, num_results=0):
        raise NotImplementedError("the library-owned result arena belongs to a one-device handle; use search_arrays")

    def _search_batch_call(self, *args):
        return self._lib.cobs_gpu_multi_search_batch(self._m, *args)

    def counts(self, query):
        raise NotImplementedError("raw counts are per shard: use shard(rank).counts(query)")


class _DevArray:
    """__cuda_array_interface__ view of HBM owned by libcobs_gpu (zero copy into torch)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}
        self._owner = owner


class Batch:
    """Device-resident query batch (cobs_gpu_batch_*): inputs are copied to HBM
    once, run() launches the hot path asynchronously, counts stay in HBM."""

    def __init__(self, search, max_queries=0, max_query_len=0, _borrowed=None):
        self._s = search
        self._lib = search._lib
        self._h = C.c_void_p()
        self._owned = _borrowed is None
        if _borrowed is not None:           # a sub-batch of a ShardedBatch: the library object owns it
            self._h = C.c_void_p(_borrowed)
        else:
            check(self._lib.cobs_gpu_batch_create(search._h, max_queries, max_query_len, C.byref(self._h)))
        self.nq = 0

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._lib.cobs_gpu_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_queries(self, queries):
        qs = [_as_bytes(q) for q in queries]
        arr = (C.c_char_p * max(len(qs), 1))(*qs)
        lens = (C.c_size_t * max(len(qs), 1))(*[len(q) for q in qs])
        check(self._lib.cobs_gpu_batch_set_queries(self._h, arr, lens, len(qs)))
        self.nq = len(qs)

    def run(self, threshold=0.0, stream=0):
        check(self._lib.cobs_gpu_batch_run(self._h, float(threshold), C.c_void_p(stream)))

    def run_hits(self, threshold, stream=0):
        """hot path without score rows: only the documents reaching the threshold are recorded"""
        check(self._lib.cobs_gpu_batch_run_hits(self._h, float(threshold), C.c_void_p(stream)))

    def run_topk(self, threshold=0.0, num_results=10, stream=0, keep_counts=True):
        """hot path + on-device selection of the num_results best documents per query;
        keep_counts=False: no score rows (K2 selects per tile, K3 merges the candidates)"""
        fn = self._lib.cobs_gpu_batch_run_topk if keep_counts else self._lib.cobs_gpu_batch_run_topk_only
        check(fn(self._h, float(threshold), int(num_results), C.c_void_p(stream)))

    def sync(self, stream=0):
        bad = C.c_size_t(0)
        check(self._lib.cobs_gpu_batch_sync(self._h, C.c_void_p(stream), C.byref(bad)))

    def counts_device(self):
        """-> (device pointer, element bytes, row stride bytes)"""
        eb, rs = C.c_uint32(0), C.c_uint64(0)
        p = self._lib.cobs_gpu_batch_counts_device(self._h, C.byref(eb), C.byref(rs))
        return p, eb.value, rs.value

    def counts_tensor(self):
        """torch view [nq, local_counts]: uint8 scores when no query of the batch has more than 255
        terms (the reference's uint8_t path), else int16 / int32 bit patterns of u16 / u32 scores."""
        import torch
        p, eb, rs = self.counts_device()
        n = self._s.local_counts
        if self.nq == 0 or n == 0:
            return torch.empty((self.nq, n), dtype={1: torch.uint8, 2: torch.int16}.get(eb, torch.int32),
                               device="cuda")
        return torch.as_tensor(_DevArray(p, (self.nq, n), {1: "|u1", 2: "<i2"}.get(eb, "<i4"), self), device="cuda")

    # -- multi-GPU exchange (native RCCL, comm.cpp) ---------------------------------------
    def exchange_counts(self, comm, mode=_capi.XCHG_ALLTOALL, stream=0):
        check(self._lib.cobs_gpu_batch_exchange_counts(self._h, comm._h, int(mode), C.c_void_p(stream)))

    def exchange_hits(self, comm, stream=0):
        """-> True if some shard's hit pool overflowed (lists incomplete)"""
        over = C.c_int(0)
        check(self._lib.cobs_gpu_batch_exchange_hits(self._h, comm._h, C.c_void_p(stream), C.byref(over)))
        return bool(over.value)

    def exchange_hits_owned(self, comm, stream=0):
        """every hit record to the rank that owns its query -> (overflowed, first owned query, owned queries)"""
        over, q0, qn = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        check(self._lib.cobs_gpu_batch_exchange_hits_owned(self._h, comm._h, C.c_void_p(stream), C.byref(over),
                                                           C.byref(q0), C.byref(qn)))
        return bool(over.value), int(q0.value), int(qn.value)

    def bucketed_hits(self, nranks):
        """diagnostics: the hit pool of the last run bucketed by query owner as `nranks` ranks would
        -> (counts per owner, records uint32 [n][4] = query, file, doc, score)"""
        counts = (C.c_uint64 * nranks)()
        n = C.c_size_t(0)
        cap = 1 << 16
        while True:
            rec = np.zeros((cap, 4), dtype=np.uint32)
            st = self._lib.cobs_gpu_batch_bucketed_hits(self._h, nranks, counts, rec.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                        cap, C.byref(n))
            if st == _capi.ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            check(st)
            return list(counts), rec[:n.value]

    def exchange_topk(self, comm, stream=0):
        check(self._lib.cobs_gpu_batch_exchange_topk(self._h, comm._h, C.c_void_p(stream)))

    def exchange_bytes(self):
        return int(self._lib.cobs_gpu_batch_exchange_bytes(self._h))

    def global_counts_tensor(self):
        """after exchange_counts: (q_begin, q_count, torch view [q_count, total_counts]) of the
        assembled rows this rank holds (global document order)"""
        import torch
        q0, qn, eb, rs = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        p = self._lib.cobs_gpu_batch_global_counts_device(self._h, C.byref(q0), C.byref(qn), C.byref(eb), C.byref(rs))
        n = self._s.total_counts
        dt = {1: torch.uint8, 2: torch.int16}.get(eb.value, torch.int32)
        if not p or qn.value == 0 or n == 0:
            return int(q0.value), int(qn.value), torch.empty((int(qn.value), n), dtype=dt, device="cuda")
        t = torch.as_tensor(_DevArray(p, (int(qn.value), n), {1: "|u1", 2: "<i2"}.get(eb.value, "<i4"), self),
                            device="cuda")
        return int(q0.value), int(qn.value), t

    def counts_host(self, query_no):
        out = np.zeros(self._s.total_counts, dtype=np.uint32)
        check(self._lib.cobs_gpu_batch_counts_host(self._h, query_no, out.ctypes.data, out.size))
        return out

    def score_histogram(self, nbins):
        """distribution of the scores of the last run over its real documents (`cobs benchmark-fpr --dist`) -> uint64[nbins]"""
        h = np.zeros(nbins, dtype=np.uint64)
        check(self._lib.cobs_gpu_batch_score_histogram(self._h, h.ctypes.data_as(C.POINTER(C.c_uint64)), nbins))
        return h

    def hits_host(self, query_no, num_results=0):
        cap = max(1, self._s.total_counts if num_results == 0 else min(num_results, self._s.total_counts))
        hits = (Hit * cap)()
        n = C.c_size_t(0)
        check(self._lib.cobs_gpu_batch_hits_host(self._h, query_no, int(num_results), hits, cap, C.byref(n)))
        return [(hits[i].file_no, hits[i].doc, hits[i].score) for i in range(n.value)]

    def stats(self):
        s = (C.c_uint64 * 4)()
        check(self._lib.cobs_gpu_batch_stats(self._h, C.byref(s)))
        return {"algorithmic_bytes": int(s[0]), "scan_launches": int(s[1]), "kmer_lookups": int(s[2]),
                "table_bytes": int(s[3])}

    def kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        check(self._lib.cobs_gpu_batch_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return {"scan_ms": a.value, "hash_ms": b.value}


__all__ = ["Search", "SearchResult", "Batch", "CobsGpuError"]


class ShardedBatch:
    """The device-resident form of the sharded search (cobs_gpu_sharded_batch_*): ONE batch of queries, uploaded once;
    step() scans it on this rank's shard and exchanges the count rows over the communicator -- cut into sub-batches
    whose hashing, scan and exchange overlap inside the library (three streams tied by events), also across steps.
    Collective: every rank makes the same calls.  sub(i) are the sub-batches as Batch views (global / local count rows)."""

    def __init__(self, search, comm, sub_batches=1):
        self._s, self._comm = search, comm
        self._lib = search._lib
        self._h = C.c_void_p()
        check(self._lib.cobs_gpu_sharded_batch_create(search._h, comm._h, int(sub_batches), C.byref(self._h)))
        self.nq = 0
        self._subs = []

    def close(self):
        if getattr(self, "_h", None):
            for b in self._subs:
                b.close()
            self._lib.cobs_gpu_sharded_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_queries(self, queries):
        qs = [_as_bytes(q) for q in queries]
        arr = (C.c_char_p * max(len(qs), 1))(*qs)
        lens = (C.c_size_t * max(len(qs), 1))(*[len(q) for q in qs])
        check(self._lib.cobs_gpu_sharded_batch_set_queries(self._h, arr, lens, len(qs)))
        self.nq = len(qs)
        self._subs = []
        for i in range(int(self._lib.cobs_gpu_sharded_batch_subs(self._h))):
            q0, qn = C.c_size_t(0), C.c_size_t(0)
            h = self._lib.cobs_gpu_sharded_batch_sub(self._h, i, C.byref(q0), C.byref(qn))
            b = Batch(self._s, _borrowed=h)
            b.nq, b.q_begin = int(qn.value), int(q0.value)
            self._subs.append(b)

    @property
    def subs(self):
        return list(self._subs)

    def step(self, threshold=0.0, mode=_capi.XCHG_ALLTOALL):
        check(self._lib.cobs_gpu_sharded_batch_step(self._h, float(threshold), int(mode)))

    def sync(self):
        bad = C.c_size_t(0)
        check(self._lib.cobs_gpu_sharded_batch_sync(self._h, C.byref(bad)))

    def times(self):
        """after sync(): per step, summed over the sub-batches, averaged over the steps since the previous call"""
        t = (C.c_double * 8)()
        check(self._lib.cobs_gpu_sharded_batch_times(self._h, C.byref(t)))
        return {"scan_ms": t[0], "hash_ms": t[1], "exchange_ms": t[2], "algorithmic_bytes": int(t[3]),
                "received_bytes": int(t[4]), "scan_launches": int(t[5]), "steps": int(t[6])}
