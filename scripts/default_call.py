#!/usr/bin/env python3
"""scripts/default_call.py -- the reference's default call (threshold 0, no limit: EVERY document of every query in rank
order, classic_search.cpp:134-156) on the C3 index, 256 queries per call, with the ordered records crossing PCIe as
one u32 (rank_pack 1) or as 8-byte pairs (rank_pack 0): interleaved A/B inside one process.
COBS_GPU_TRACE=1 prints where the host side of each call waits."""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    if os.environ.get("BIND_NUMA"):          # caller (and its result array) on the GPU's NUMA node
        import torch  # noqa: F401
        n = bench.gpu_numa_node(0)
        print("GPU on NUMA node %s: process bound to %d of its CPUs" % (n, bench.bind_to_numa_node(n)))
    cfg = bench.c3_config()
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    queries = bench.make_queries(nq, 1000)
    text = np.frombuffer(b"".join(queries), dtype=np.uint8)
    offsets = np.zeros(nq + 1, dtype=np.uint64)
    np.cumsum([len(q) for q in queries], out=offsets[1:])
    keep = np.zeros(nq * s.total_counts, dtype=s.HIT_DTYPE)
    windows = [int(v) for v in os.environ.get("RANK_WINDOWS_KIB", "").split(",") if v]
    if windows:          # A/B of the piece size of the PCIe | expansion pipeline, packed records
        wt = {w: [] for w in windows}
        for rnd in range(8):
            for w in windows:
                s.set_tuning("rank_window_kib", w)
                t0 = time.perf_counter()
                s.search_packed(text, offsets, 0.0, 0, out=keep)
                if rnd:
                    wt[w].append(time.perf_counter() - t0)
        for w in windows:
            med = statistics.median(wt[w])
            print("rank_window_kib %6d: median %.3f ms  min %.3f  -> %.1f k queries/s" % (w, med * 1e3, min(wt[w]) * 1e3, nq / med / 1e3))
        s.set_tuning("rank_window_kib", 0)
    times = {0: [], 1: []}
    ref = None
    for rnd in range(6):
        for pack in (1, 0):
            s.set_tuning("rank_pack", pack)
            t0 = time.perf_counter()
            offs, hits = s.search_packed(text, offsets, 0.0, 0, out=keep)
            dt = time.perf_counter() - t0
            if rnd:
                times[pack].append(dt)
            if ref is None:
                ref = hits.copy()
            elif rnd == 1:
                assert np.array_equal(ref, hits), "packed and pair records differ"
    for pack in (1, 0):
        med = statistics.median(times[pack])
        rec = 4 if pack else 8
        print("rank_pack %d: %d queries x %d documents  median %.3f ms  min %.3f  -> %.1f k queries/s, %.1f GB/s of %d-byte records"
              % (pack, nq, len(hits) // nq, med * 1e3, min(times[pack]) * 1e3, nq / med / 1e3, len(hits) * rec / med / 1e9, rec))


if __name__ == "__main__":
    main()
