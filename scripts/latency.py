#!/usr/bin/env python3
"""scripts/latency.py -- small-batch behaviour of the host API on the C3 index."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402

cfg = bench.c3_config()
s = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
qs = bench.make_queries(4096, 1000)
b = cobs_amd.Batch(s)
for n in (1, 4, 16, 64, 256, 1024, 4096):
    b.set_queries(qs[:n])
    for _ in range(3):
        b.run_topk(0.0, 10)
    b.sync()
    b.kernel_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        b.run_topk(0.0, 10)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / reps
    b.sync()
    ms = b.kernel_ms()
    # host API: text H2D + K1 + K2 + K3 + D2H of the top-10 + ordering
    s.search_hits(qs[:n], 0.0, 10)
    t0 = time.perf_counter()
    for _ in range(5):
        s.search_hits(qs[:n], 0.0, 10)
    host = (time.perf_counter() - t0) / 5
    print("nq=%5d  device pass %8.3f ms (scan %7.3f hash %6.3f)  %9.0f q/s   host API top-10 %8.3f ms  %9.0f q/s"
          % (n, dev * 1e3, ms["scan_ms"], ms["hash_ms"], n / dev, host * 1e3, n / host), flush=True)

# single-query latency of the host API with and without the captured graph (device pass replayed with one launch)
for n in (1, 4, 16):
    for g in (0, -1):
        s.set_tuning("graph", g)
        for _ in range(4):
            s.search_hits(qs[:n], 0.0, 10)
        t0 = time.perf_counter()
        for i in range(200):
            s.search_hits(qs[i % 64:i % 64 + n], 0.0, 10)
        host = (time.perf_counter() - t0) / 200
        print("nq=%5d  host API top-10, graph %s: %7.1f us per call (%d replays so far)"
              % (n, "off" if g == 0 else "on ", host * 1e6, s.graph_replays), flush=True)

# single queries of FOUR lengths in rotation (a service sees a handful of read lengths): every shape keeps
# its captured graph (the current one plus three older ones)
rot = [qs[i][:L] for i, L in enumerate((100, 150, 250, 331))]
for g in (0, -1):
    s.set_tuning("graph", g)
    for i in range(12):
        s.search_hits([rot[i % 4]], 0.0, 10)
    r0 = s.graph_replays
    t0 = time.perf_counter()
    for i in range(400):
        s.search_hits([rot[i % 4]], 0.0, 10)
    host = (time.perf_counter() - t0) / 400
    print("nq=    1  four query lengths in rotation, graph %s: %7.1f us per call (%d of 400 calls replayed)"
          % ("off" if g == 0 else "on ", host * 1e6, s.graph_replays - r0), flush=True)

# single queries of TWENTY lengths of one shape class (131..150 bp) in rotation: graphs are keyed by shape class
for g in (0, -1):
    s.set_tuning("graph", g)
    many = [qs[i][:131 + i] for i in range(20)]
    for i in range(40):
        s.search_hits([many[i % 20]], 0.0, 10)
    r0 = s.graph_replays
    t0 = time.perf_counter()
    for i in range(400):
        s.search_hits([many[i % 20]], 0.0, 10)
    host = (time.perf_counter() - t0) / 400
    print("nq=    1  twenty query lengths of one shape class in rotation, graph %s: %7.1f us per call (%d of 400 calls replayed)"
          % ("off" if g == 0 else "on ", host * 1e6, s.graph_replays - r0), flush=True)

# the reference's default call, search(query) = threshold 0, num_results 0: EVERY document ranked.
# On the device (rank_kernels.hip: the ordered records cross PCIe) vs by host threads (the score rows cross);
# into a fresh result array per call vs into one the caller keeps (no page faults on 307 MB).
import numpy as np
keep = np.zeros(256 * s.total_counts, dtype=s.HIT_DTYPE)
for n in (1, 16, 64, 256, 1024):
    for dr in (0, 1):
        s.set_tuning("device_rank", dr)
        for reuse in (False, True):
            out = keep if reuse and n <= 256 else None
            s.search_arrays(qs[:n], 0.0, 0, out=out)
            s.timers(reset=True)
            reps = 5 if n < 256 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                offs, hits = s.search_arrays(qs[:n], 0.0, 0, out=out)
            host = (time.perf_counter() - t0) / reps
            tm = s.timers(reset=True)
            print("nq=%5d  all %d documents ranked per query, %s, %s result buffer: %9.3f ms  %8.0f q/s  %6.2f GB/s of records (phases, ms: %s)"
                  % (n, len(hits) // n, "on the device" if dr else "by host threads", "kept " if out is not None else "fresh",
                     host * 1e3, n / host, len(hits) * 12 / host / 1e9,
                     " ".join("%s %.3f" % (k, v / reps * 1e3) for k, v in tm.items())), flush=True)
s.set_tuning("device_rank", 1)

# the Python mirror of that call: 100k SearchResult objects per query
s.search(qs[0])
t0 = time.perf_counter()
for _ in range(20):
    r = s.search(qs[0])
print("python Search.search(query) -> %d SearchResult objects: %.2f ms" % (len(r), (time.perf_counter() - t0) / 20 * 1e3))
