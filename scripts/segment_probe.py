#!/usr/bin/env python3
"""scripts/segment_probe.py -- would row-range segments pay?  The rows of one large sub-index
(4 M rows: a 128-byte tile slice is 512 MB, twice the Infinity Cache) are looked up ~2.5 times
per batch; cut into R row ranges whose tile slices fit the cache, every line would come from HBM
once.  Emulation with the existing kernels, WITHOUT the confound of round 1's probe (R-fold score
traffic): R sub-indexes of 4M/R rows, queries of 1000/R k-mers, hits-only passes (no score rows).
Same lookups, same gathered bytes, same rows in total."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def run(name, sigs, nq, kmers, tuning=None, hits=True, reps=5):
    docs = len(sigs) * 12544
    s = cobs_amd.Search.synthetic("compact", sigs, docs, page_size=1568, seed=1)
    for k, v in (tuning or {}).items():
        s.set_tuning(k, v)
    b = cobs_amd.Batch(s)
    b.set_queries(bench.make_queries(nq, kmers))
    f = (lambda: b.run_hits(0.99)) if hits else (lambda: b.run(0.0))
    for _ in range(2):
        f()
    b.sync()
    b.kernel_ms()
    for _ in range(reps):
        f()
    b.sync()
    ms = b.kernel_ms()["scan_ms"]
    gb = b.stats()["algorithmic_bytes"] / 1e9
    print("%-58s scan %8.3f ms   %7.1f GB/s algorithmic" % (name, ms, gb / ms * 1e3), flush=True)
    del b, s
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("1 x 4M rows, T=1000, 10k queries (today)", [4000000], 10000, 1000)
    for R in (2, 4, 8, 16):
        T = 1000 // R
        for tun in ({}, {"tile_w": 8, "waves": 1}, {"tile_w": 8, "waves": 2}, {"tile_w": 16, "waves": 1}):
            run("%2d x %4dk rows, T=%3d, 10k queries %s" % (R, 4000 // R, T, tun), [4000000 // R] * R, 10000, T, tun)
    run("1 x 2.7M rows, T=1000 (today)", [2700000], 10000, 1000)
    for R in (4, 6):
        run("%d x %dk rows, T=%d" % (R, 2700 // R, 1000 // R), [2700000 // R] * R, 10000, 1000 // R, {"tile_w": 8, "waves": 1})
