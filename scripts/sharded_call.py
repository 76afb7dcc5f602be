#!/usr/bin/env python3
"""scripts/sharded_call.py -- the PRODUCT call of the multi-GPU layout next to the one-GPU call, on one rank over real RCCL.

cobs_gpu_sharded_search_batch (what cobs_gpu_multi_search_batch / cobs_gpu::ShardedClassicSearch / `cobs_gpu_query -d`
run per rank) against cobs_gpu_search_batch, same index, same queries packed back to back, same thresholds: 10 000
planted queries at the CLI's default threshold 0.8 (264 000 hits), the random batch (0 hits), a limit of 10, and the
reference's default call (every document, 256 queries).  VERDICT r5 item 1: within 5 % at the thresholded call.
-> profiles/r06_latency.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402
from cobs_amd.distributed import Comm  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = bench.c3_config()
cfg["plants"] = bench.planted_documents(cfg, 1000)
s = bench.make_index(cfg, 0)
comm = Comm(Comm.unique_id(), 0, 1, 0)
comm.set_timeout(60000)


packed = bench.pack_queries
rand = bench.make_queries(nq, 1000)
hitq = bench.planted_queries(cfg["plants"], nq, 1000)
cases = [("threshold 0.8, planted queries", packed(hitq), 0.8, 0, 5),
         ("threshold 0.8, random queries", packed(rand), 0.8, 0, 5),
         ("threshold 0, limit 10", packed(rand), 0.0, 10, 5),
         ("threshold 0, every document, 256 queries", packed(rand[:256]), 0.0, 0, 5)]
print("# one MI355X, C3 index (18.4 GB resident), %d queries x 1000 k-mers; best of 5 calls after one warm-up; ms per call" % nq)
print("# %-44s %12s %12s %8s %10s" % ("call", "one-GPU API", "sharded API", "ratio", "hits"))
for name, pk, thr, k, reps in cases:
    one = sh = None
    s.search_packed(pk[0], pk[1], thr, k)
    for _ in range(reps):
        t0 = time.perf_counter()
        o1, h1 = s.search_packed(pk[0], pk[1], thr, k)
        dt = time.perf_counter() - t0
        one = dt if one is None else min(one, dt)
    s.sharded_search_arrays(comm, pk, thr, k)
    for _ in range(reps):
        t0 = time.perf_counter()
        o2, h2 = s.sharded_search_arrays(comm, pk, thr, k)
        dt = time.perf_counter() - t0
        sh = dt if sh is None else min(sh, dt)
    same = bool(np.array_equal(np.asarray(o1, dtype=np.uint64), o2) and np.array_equal(h1, h2))
    print("  %-44s %12.3f %12.3f %8.3f %10d %s" % (name, one * 1e3, sh * 1e3, sh / one, len(h2), "" if same else "RESULTS DIFFER"), flush=True)
    assert same, name
