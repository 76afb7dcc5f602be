#!/usr/bin/env python3
"""scripts/ab_topk.py SHAPE K cfg [cfg ...] -- interleaved A/B of tuning keys on the top-k pass without score rows
(K2's tile_topk epilogue + K3 pool merge), like scripts/ab.py for the score-writing pass."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    shape, k = sys.argv[1], int(sys.argv[2])
    configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[3:]] or [{}]
    cfg = bench.c3_config()
    if shape.startswith("reads"):
        nq, kmers = 40000, int(shape[5:]) - 30
    else:
        nq, kmers = 10000, 1000
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    b = cobs_amd.Batch(s)
    b.set_queries(bench.make_queries(nq, kmers))
    keys = sorted({kk for c in configs for kk in c})
    times = [[] for _ in configs]
    for rnd in range(7):
        for ci, c in enumerate(configs):
            for kk in keys:
                s.set_tuning(kk, 0)
            for kk, v in c.items():
                s.set_tuning(kk, int(v))
            for _ in range(3):
                b.run_topk(0.0, k, 0, keep_counts=False)
            b.sync()
            ms = b.kernel_ms()["scan_ms"]
            if rnd:
                times[ci].append(ms)
    for c, t in zip(configs, times):
        print("%-30s top-%d scan median %.3f ms  min %.3f  max %.3f" % (c, k, statistics.median(t), min(t), max(t)))


if __name__ == "__main__":
    main()
