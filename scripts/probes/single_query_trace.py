#!/usr/bin/env python3
"""scripts/probes/single_query_trace.py -- where the time of ONE query goes (host API, top-10, captured graph):
   cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -o sq -- python scripts/probes/single_query_trace.py
prints the wall time per call; the trace holds the device side of every call (nodes of the replayed graph)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402

cfg = bench.c3_config(float(os.environ.get("SCALE", "1.0")))
s = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
qs = bench.make_queries(64, 1000)
for _ in range(8):
    s.search_hits(qs[:1], 0.0, 10)
ts = []
for i in range(200):
    t0 = time.perf_counter_ns()
    s.search_hits(qs[i % 64:i % 64 + 1], 0.0, 10)
    ts.append(time.perf_counter_ns() - t0)
ts.sort()
print("single query, top-10, graph: median %.1f us  min %.1f us  p90 %.1f us  (%d replays)" % (ts[100] / 1e3, ts[0] / 1e3, ts[180] / 1e3, s.graph_replays))
# the same through the raw view call (no Python list building)
