#!/usr/bin/env python3
"""scripts/e2e_pipe.py -- host-buffer API on the C3 batch with and without pipelined passes
(tuning key pipe_chars = 0 disables the cut into four passes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402
cfg = bench.c3_config()
s = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
qs = bench.make_queries(10000, 1000)
import numpy as np
text = np.frombuffer(b"".join(qs), dtype=np.uint8); offs = np.arange(10001, dtype=np.uint64) * np.uint64(1030)
for pc in ("0", "4194304"):
    s.set_tuning("pipe_chars", int(pc))
    for t, lim in ((0.8, 0), (0.0, 10)):
        s.search_packed(text, offs, t, lim)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); s.search_packed(text, offs, t, lim); ts.append(time.perf_counter() - t0)
        print("pipe_chars=%s threshold=%g limit=%d: %.2f ms (min %.2f)" % (pc, t, lim, 1e3 * sorted(ts)[2], 1e3 * min(ts)))
