#!/usr/bin/env python3
"""scripts/probes/fresh_buffer_probe.py -- the default call (threshold 0, no limit: every document of every query ranked,
256 queries x 100 000 documents = 307 MB of results) into a KEPT result array, into a FRESH one per call, and into a fresh
one that was advised MADV_HUGEPAGE first: how much of the fresh-buffer cost is first-touch page faults, and whether
transparent huge pages take it away on this host."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402

libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
MADV_HUGEPAGE = 14


def main():
    cfg = bench.c3_config()
    s = bench.make_index(cfg, 0)
    nd = 256
    queries = bench.make_queries(nd, 1000)
    text = np.frombuffer(b"".join(queries), dtype=np.uint8)
    offs = np.zeros(nd + 1, dtype=np.uint64)
    np.cumsum([len(q) for q in queries], out=offs[1:])
    n = nd * s.total_counts
    out = {"thp_enabled": open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(),
           "thp_defrag": open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip()}
    keep = np.zeros(n, dtype=s.HIT_DTYPE)
    s.search_packed(text, offs, 0.0, 0, out=keep)

    def timed(make):
        best = None
        for _ in range(5):
            buf = make()
            t0 = time.perf_counter()
            s.search_packed(text, offs, 0.0, 0, out=buf)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            del buf
        return round(best * 1e3, 3)

    def fresh():
        return np.empty(n, dtype=s.HIT_DTYPE)

    def fresh_huge():
        b = np.empty(n, dtype=s.HIT_DTYPE)
        a0 = (b.ctypes.data + (1 << 21) - 1) & ~((1 << 21) - 1)
        ln = (b.ctypes.data + b.nbytes - a0) & ~((1 << 21) - 1)
        rc = libc.madvise(a0, ln, MADV_HUGEPAGE)
        out.setdefault("madvise_rc", rc)
        return b

    out["kept_ms"] = timed(lambda: keep)
    out["fresh_ms"] = timed(fresh)
    out["fresh_hugepage_ms"] = timed(fresh_huge)
    out["kept_again_ms"] = timed(lambda: keep)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
