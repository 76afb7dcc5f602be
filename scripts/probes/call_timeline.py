"""scripts/probes/call_timeline.py -- host-buffer calls one after the other with pauses between them, to be run under
`rocprofv3 --kernel-trace --output-format csv`; scripts/probes/timeline_bursts.py then groups the kernel trace into the
bursts of the calls and says how much of each call the device was busy and with what."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B

cfg = B.c3_config(1.0)
cfg["num_hashes"] = 1
cfg["plants"] = B.planted_documents(cfg, 1000)
s = B.make_index(cfg, 0)
rq = B.make_queries(10000, 1000)
hq = B.planted_queries(cfg["plants"], 10000, 1000)

def packed(qs):
    t = np.frombuffer(b"".join(qs), dtype=np.uint8)
    o = np.zeros(len(qs) + 1, dtype=np.uint64)
    np.cumsum([len(q) for q in qs], out=o[1:])
    return t, o

keep = np.zeros(256 * s.total_counts, dtype=s.HIT_DTYPE)
calls = [
    ("default call, 256 queries, kept array", lambda: s.search_packed(*packed(rq[:256]), 0.0, 0, out=keep)),
    ("top-10, 10k queries", lambda: s.search_packed(*packed(rq), 0.0, 10)),
    ("threshold 0.8, planted, 10k", lambda: s.search_packed(*packed(hq), 0.8, 0)),
    ("threshold 0.8, random, 10k", lambda: s.search_packed(*packed(rq), 0.8, 0)),
    ("one query, top-10", lambda: s.search(rq[0], 0.0, 10)),
    ("one query, threshold 0.8", lambda: s.search(hq[0], 0.8, 0)),
]
for name, fn in calls:
    for _ in range(2):
        fn()
    time.sleep(0.05)
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print("CALL %-40s %.3f ms" % (name, dt * 1e3), flush=True)
        time.sleep(0.05)
