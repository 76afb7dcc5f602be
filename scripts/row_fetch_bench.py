#!/usr/bin/env python3
"""scripts/row_fetch_bench.py [scale] [budget_gb] -- the out-of-core path by batch size: BASELINE configs[4]'s index
(the C3 geometry as a .cobs_compact FILE, 18.4 GB at scale 1) under an HBM budget, batches of 1 ... 10 000 queries:
the row-selective pass (rows fetched from the registered mapping, fetch_kernels.hip), whole-chunk streaming, and the
engine's own per-chunk choice.  Prints one line per (batch size, mode): pass time, chunks fetched / streamed, the
PCIe rate of what moved."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    budget = int(float(sys.argv[2]) * 1e9) if len(sys.argv) > 2 else int(6e9)
    cfg = bench.c3_config(scale)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "row_fetch_%g.cobs_compact" % scale)
    if not os.path.exists(path):
        t0 = time.perf_counter()
        cobs_amd.write_synthetic(path, "compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
        print("wrote %.2f GB in %.1f s" % (os.path.getsize(path) / 1e9, time.perf_counter() - t0), flush=True)
    size = os.path.getsize(path)
    t0 = time.perf_counter()
    s = cobs_amd.Search(path, hbm_budget=budget)
    print("opened under a budget of %.1f GB in %.2f s (hbm %.2f GB)" % (budget / 1e9, time.perf_counter() - t0, s.info(0).hbm_bytes / 1e9), flush=True)
    qs = bench.make_queries(10000, 1000)
    P, ps = len(cfg["signature_sizes"]), cfg["page_size"]
    ref = None
    sizes = [int(a) for a in sys.argv[3:] if a.isdigit()] or [1, 4, 16, 64, 256, 1024, 4096, 10000]
    for nq in sizes:
        b = cobs_amd.Batch(s)
        b.set_queries(qs[:nq])
        for mode, key, val in (("rows", "row_fetch_alpha", 0), ("whole", "row_fetch", 0), ("auto", "row_fetch_alpha", 1)):
            s.set_tuning("row_fetch", 1)
            s.set_tuning(key, val)
            b.run(0.0)
            b.sync()
            row0 = b.counts_host(0)
            if ref is None:
                ref = row0
            ok = np.array_equal(row0, ref)
            f0, w0 = s.stream_counters()
            steps = 3 if nq >= 1024 else 10
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                b.run(0.0)
            b.sync()
            dt = (time.perf_counter() - t0) / steps
            f1, w1 = s.stream_counters()
            nf, nw = (f1 - f0) // steps, (w1 - w0) // steps
            looked = nq * 1008 * P * ps                       # bytes of looked-up rows incl. block padding
            print("nq=%5d  %-5s  pass %9.3f ms  %9.0f queries/s   chunks: %2d by rows, %2d whole   looked-up rows %8.1f MB%s  bit-exact=%s"
                  % (nq, mode, dt * 1e3, nq / dt, nf, nw, looked / 1e6,
                     "  -> %.1f GB/s over PCIe" % (looked / dt / 1e9) if nw == 0 else
                     ("  (file %.1f GB -> %.1f GB/s)" % (size / 1e9, size / dt / 1e9) if nf == 0 else ""), ok), flush=True)
        del b
    if "--keep" not in sys.argv:
        os.remove(path)


if __name__ == "__main__":
    main()
