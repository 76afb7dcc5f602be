"""scripts/probes/default_call_probe.py [window_kib ...] -- the reference's default call (threshold 0, no limit) of 256
queries on the C3 index with the slot-stream form on and off, best of 7, and (COBS_GPU_TRACE=1) the library's own split of
the ranking leg."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B

cfg = B.c3_config(1.0)
cfg["num_hashes"] = 1
s = B.make_index(cfg, 0)
rq = B.make_queries(256, 1000)
text = np.frombuffer(b"".join(rq), dtype=np.uint8)
offs = np.zeros(257, dtype=np.uint64)
np.cumsum([len(q) for q in rq], out=offs[1:])
keep = np.zeros(256 * s.total_counts, dtype=s.HIT_DTYPE)
windows = [int(x) for x in sys.argv[1:]] or [16384]
for win in windows:
    s.set_tuning("rank_window_kib", win)
    for slim, nseg in ((1, 0), (1, 1), (0, 1), (1, 0), (1, 1)) if not os.environ.get("PROBE_SHORT") else ((1, 0), (1, 0), (1, 0)):
        s.set_tuning("rank_slim", slim)
        s.set_tuning("rank_segments", nseg)
        for nq in ((256, 64, 16) if not os.environ.get("PROBE_SHORT") else (256,)):
            o = np.ascontiguousarray(offs[:nq + 1])
            t = text[:int(o[-1])]
            for _ in range(2):
                s.search_packed(t, o, 0.0, 0, out=keep)
            best = 1e9
            for _ in range(7):
                t0 = time.perf_counter()
                s.search_packed(t, o, 0.0, 0, out=keep)
                best = min(best, time.perf_counter() - t0)
            print("window %6d KiB  rank_slim %d  rank_segments %d  default call of %3d queries: %.3f ms = %.1f k queries/s"
                  % (win, slim, nseg, nq, best * 1e3, nq / best / 1e3), flush=True)
