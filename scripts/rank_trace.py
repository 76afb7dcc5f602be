#!/usr/bin/env python3
"""scripts/rank_trace.py -- where a 256-query default call (every document ranked) spends its time"""
import os, sys, time
import numpy as np
os.environ["COBS_GPU_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, cobs_amd
cfg = bench.c3_config()
s = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
qs = bench.make_queries(256, 1000)
keep = np.zeros(256 * s.total_counts, dtype=s.HIT_DTYPE)
text = np.frombuffer(b"".join(qs), dtype=np.uint8)
offsets = np.zeros(257, dtype=np.uint64); np.cumsum([len(q) for q in qs], out=offsets[1:])
for i in range(4):
    t0 = time.perf_counter()
    s.search_packed(text, offsets, 0.0, 0, out=keep)
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
