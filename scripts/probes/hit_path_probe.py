"""scripts/probes/hit_path_probe.py -- the thresholded call (0.8, hits only) on queries WITH hits and on random queries,
alternating, with the library's phase timers: is the scan itself slower when records are appended, or is it order /
first-call effects?  Also a Batch-level scan of both query sets (kernel_ms: HIP events around K2 only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
import cobs_amd

dev = 0
cfg = B.c3_config(1.0)
cfg["num_hashes"] = 1
cfg["plants"] = B.planted_documents(cfg, 1000)
s = B.make_index(cfg, dev)
rq = B.make_queries(10000, 1000)
hq = B.planted_queries(cfg["plants"], 10000, 1000)

def packed(qs):
    t = np.frombuffer(b"".join(qs), dtype=np.uint8)
    o = np.zeros(len(qs) + 1, dtype=np.uint64)
    np.cumsum([len(q) for q in qs], out=o[1:])
    return t, o

sets = {"random": packed(rq), "planted": packed(hq)}
for name in ("random", "planted", "random", "planted"):
    s.search_packed(*sets[name], 0.8, 0)
for rnd in range(4):
    for name in ("random", "planted"):
        s.timers(reset=True)
        t0 = time.perf_counter()
        offs, hits = s.search_packed(*sets[name], 0.8, 0)
        dt = time.perf_counter() - t0
        tm = s.timers()
        print("%-8s call %.3f ms  hits %7d  %s" % (name, dt * 1e3, len(hits), {k: round(v * 1e3, 3) for k, v in tm.items()}), flush=True)
b = cobs_amd.Batch(s)
for name, qs in (("random", rq), ("planted", hq), ("random", rq), ("planted", hq)):
    b.set_queries(qs)
    for _ in range(2):
        b.run(0.8, 0)
    b.sync(); b.kernel_ms()
    for _ in range(5):
        b.run(0.8, 0)
    b.sync()
    print("batch", name, b.kernel_ms(), "hits", b.stats().get("hits"), flush=True)
