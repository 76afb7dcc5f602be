#!/usr/bin/env python3
"""scripts/shapes.py -- scan-kernel throughput over index / query shapes (tuning aid)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def geo(lo, hi, n):
    if n == 1:
        return [lo]
    r = (hi / lo) ** (1.0 / (n - 1))
    return [int(lo * r ** i) for i in range(n)]


def run(name, kind, sigs, ndocs, page_size, nq, kmers, H=1, steps=5, tuning=None):
    s = cobs_amd.Search.synthetic(kind, sigs, ndocs, page_size=page_size, num_hashes=H, seed=1)
    for k, v in (tuning or {}).items():
        s.set_tuning(k, v)
    qs = bench.make_queries(nq, kmers)
    b = cobs_amd.Batch(s)
    b.set_queries(qs)
    for _ in range(2):
        b.run(0.0)
    b.sync()
    b.kernel_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run(0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    b.sync()
    ms = b.kernel_ms()
    st = b.stats()
    gb = st["algorithmic_bytes"] / 1e9
    info = s.info(0)
    print("%-44s idx=%6.1fGB alg=%7.2fGB scan=%8.3fms hash=%7.3fms step=%8.3fms  %7.1f GB/s (%.1f%%)  %9.0f q/s"
          % (name, info.hbm_bytes / 1e9, gb, ms["scan_ms"], ms["hash_ms"], dt * 1e3,
             gb / ms["scan_ms"] * 1e3, gb / ms["scan_ms"] * 1e3 / 80, nq / dt), flush=True)
    del b, s


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    def on(k):
        return "all" in which or k in which
    if on("c3"):
        run("C3 compact 100k docs ps1568 P8 T1000", "compact", bench.c3_config()["signature_sizes"], 100000, 1568, 10000, 1000)
    if on("c2"):
        run("C2 classic 10k docs S1M Q1k T1000", "classic", [1000000], 10000, 0, 1000, 1000)
        run("C2 classic 10k docs S1M Q10k T1000", "classic", [1000000], 10000, 0, 10000, 1000)
    if on("c4"):
        run("C4 compact 1M docs ps512 P245 Q1k", "compact", geo(100000, 1600000, 245), 1000000, 512, 1000, 1000)
    if on("small"):
        run("compact 100k docs ps128 P98 Q4k", "compact", geo(250000, 4000000, 98), 100000, 128, 4000, 1000)
        run("compact 6k docs ps8 P98 Q4k", "compact", geo(250000, 4000000, 98), 6200, 8, 4000, 1000)
        run("classic 1000 docs S4M Q10k", "classic", [4000000], 1000, 0, 10000, 1000)
    if on("h3"):
        run("C3-shape H=3 Q4k", "compact", bench.c3_config()["signature_sizes"], 100000, 1568, 4000, 1000, H=3)
    if on("short"):
        run("C3 T=100 Q20k", "compact", bench.c3_config()["signature_sizes"], 100000, 1568, 20000, 100)
        run("C3 T=20 Q20k", "compact", bench.c3_config()["signature_sizes"], 100000, 1568, 20000, 20)
        run("C3 T=5000 Q2k", "compact", bench.c3_config()["signature_sizes"], 100000, 1568, 2000, 5000)
    if on("huge"):
        run("compact 25k docs ps1568 P2 S=30M Q2k", "compact", [30000000, 30000000], 25000, 1568, 2000, 1000)
        run("compact 100k docs ps512 P25 S=6M Q2k", "compact", [6000000] * 25, 100000, 512, 2000, 1000)
    if on("reads"):
        for bp in (100, 150, 250):
            run("C3 %d-bp reads Q40k" % bp, "compact", bench.c3_config()["signature_sizes"], 100000, 1568, 40000, bp - 30)
    if on("vpage"):
        # emulation of a row-range split of a 4M-row sub-index into R passes: same lookups and
        # gathered bytes, R x fewer rows per scanned slice, R x fewer terms per work-group
        for R in (1, 2, 4, 8, 16):
            run("8 x S=%dk T=%d Q=%dk (R=%d)" % (4000 // R, 1000 // R, 10 * R, R), "compact",
                [4000000 // R] * 8, 100000, 1568, 10000 * R, 1000 // R)
    if on("mq"):
        # multi-query work-groups vs the default geometry (per-handle tuning keys mq / tile_w / waves)
        c3 = bench.c3_config()["signature_sizes"]
        for T, nq in ((20, 40000), (70, 40000), (120, 40000), (220, 40000), (500, 20000), (1000, 10000)):
            for mq, w, nw in ((0, 0, 0), (1, 8, 1), (1, 8, 2), (1, 8, 4), (1, 16, 1), (1, 16, 2), (1, 16, 4), (1, 32, 2)):
                run("C3 T=%d Q=%dk mq=%d W=%d NW=%d" % (T, nq // 1000, mq, w, nw), "compact", c3, 100000, 1568, nq, T, steps=3,
                    tuning={"mq": mq, "tile_w": w, "waves": nw})
    if on("ssweep"):
        # throughput vs slice working set at a fixed query length: 8 equal sub-indexes of S rows
        for S in (8000000, 4000000, 2000000, 1000000, 500000, 250000, 125000, 62500):
            run("8 x S=%dk T=1000 Q=10k" % (S // 1000), "compact", [S] * 8, 100000, 1568, 10000, 1000)
