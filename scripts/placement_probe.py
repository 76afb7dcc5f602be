#!/usr/bin/env python3
"""scripts/placement_probe.py SHAPE -- does WHERE a batch's buffers land in HBM change the scan time?  Round 4 saw the same
binary / box / configuration differ by 5.6 % between processes and by < 0.3 % inside one.  Here, inside one process: the
index stays, the batch (score rows, row-index table, query text) is re-created several times behind spacer allocations
of different sizes, and each incarnation is timed."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "reads50"
    cfg = bench.c3_config()
    if shape.startswith("reads"):
        nq, kmers = 40000, int(shape[5:]) - 30
    else:
        nq, kmers = 10000, 1000
    queries = bench.make_queries(nq, kmers)
    for reopen in range(2):
        s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
        spacers = []
        for i, mb in enumerate((0, 3, 64, 515, 1, 2049)):
            if mb:
                spacers.append(torch.empty(mb << 20, dtype=torch.uint8, device="cuda"))
            b = cobs_amd.Batch(s)
            b.set_queries(queries)
            t = []
            for r in range(5):
                for _ in range(3):
                    b.run(0.0)
                b.sync()
                ms = b.kernel_ms()["scan_ms"]
                if r:
                    t.append(ms)
            p, eb, rs = b.counts_device()
            print("index #%d  batch #%d (spacer %4d MiB)  counts at 0x%x  scan median %.3f ms  min %.3f  max %.3f"
                  % (reopen, i, mb, p, statistics.median(t), min(t), max(t)), flush=True)
            del b
        del spacers, s
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
