"""scripts/py_overhead.py -- what the Python mirror adds to a 256-query default call (every document ranked): packed text vs a
list of queries, a result array the caller keeps vs a fresh one per call (page faults on 307 MB)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, cobs_amd
cfg = bench.c3_config()
s = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
qs = bench.make_queries(256, 1000)
keep = np.zeros(256 * s.total_counts, dtype=s.HIT_DTYPE)
text = np.frombuffer(b"".join(qs), dtype=np.uint8)
offsets = np.zeros(257, dtype=np.uint64); np.cumsum([len(q) for q in qs], out=offsets[1:])
def t(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("search_packed(np text, kept): %.3f ms" % t(lambda: s.search_packed(text, offsets, 0.0, 0, out=keep)))
print("search_packed(bytes text, kept): %.3f ms" % t(lambda: s.search_packed(b"".join(qs), offsets, 0.0, 0, out=keep)))
print("search_arrays(list, kept): %.3f ms" % t(lambda: s.search_arrays(qs, 0.0, 0, out=keep)))
print("search_arrays(list, fresh): %.3f ms" % t(lambda: s.search_arrays(qs, 0.0, 0)))
