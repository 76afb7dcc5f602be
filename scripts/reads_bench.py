#!/usr/bin/env python3
"""scripts/reads_bench.py -- host-buffer API (query text in host memory -> hits in host memory)
for batches of sequencing reads against the C3 index, threshold 0.8 (the `cobs query` default)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402

cfg = bench.c3_config()
s = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
for bp, nq in ((50, 100000), (100, 100000), (150, 100000), (250, 50000)):
    reads = bench.make_queries(nq, bp - 30, seed=bp)
    s.search_arrays(reads[:1000], 0.8, 0)
    s.timers(reset=True)
    t0 = time.perf_counter()
    offs, hits = s.search_arrays(reads, 0.8, 0)
    dt = time.perf_counter() - t0
    tm = s.timers(reset=True)
    import numpy as np
    text = np.frombuffer(b"".join(reads), dtype=np.uint8)
    po = np.arange(nq + 1, dtype=np.uint64) * np.uint64(bp)
    t0 = time.perf_counter()
    offs2, hits2 = s.search_packed(text, po, 0.8, 0)
    dtp = time.perf_counter() - t0
    assert np.array_equal(offs, offs2) and np.array_equal(hits, hits2)
    s.timers(reset=True)
    print("%3d-bp reads x %6d  threshold 0.8: %7.1f ms  %9.0f reads/s  (%.1f M k-mer lookups/s; K1 %.1f ms, K2 %.1f ms, "
          "query staging %.1f ms, ranking %.1f ms, hits %d); packed input %.1f ms = %.0f reads/s"
          % (bp, nq, dt * 1e3, nq / dt, nq * (bp - 30) / dt / 1e6, tm["hashes"] * 1e3, tm["scan"] * 1e3,
             tm["h2d"] * 1e3, tm["rank"] * 1e3, len(hits), dtp * 1e3, nq / dtp), flush=True)
