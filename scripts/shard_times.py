#!/usr/bin/env python3
"""scripts/shard_times.py [N] [mode] [queries] [budget GB] -- the scan time of every shard of an N-way sharded C3 index, one after another on
this GPU: what each rank of an N-GPU run spends scanning (the step is the slowest rank).  Byte-balanced shards (mode 0)
are not time-balanced if a byte of a large sub-index costs more than a byte of a small one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    budget = int(float(sys.argv[4]) * 1e9) if len(sys.argv) > 4 else 0      # GB per shard: the shards of the C3 FILE, streamed
    cfg = bench.c4_config() if os.environ.get("SHAPE") == "c4" else bench.c3_config()      # SHAPE=c4: BASELINE configs[3]
    path = None
    if budget:
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cobs_c5_1.cobs_compact")
        if not os.path.exists(path):
            cobs_amd.write_synthetic(path, cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    queries = bench.make_queries(nq, 1000)
    times = []
    for r in range(n):
        if path:
            s = cobs_amd.Search(path, shard_rank=r, shard_count=n, shard_mode=mode, hbm_budget=budget)
        else:
            s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1,
                                          shard_rank=r, shard_count=n, shard_mode=mode)
        b = cobs_amd.Batch(s)
        b.set_queries(queries)
        for _ in range(3):
            b.run(0.0)
        b.sync()
        b.kernel_ms()
        import time
        reps = 2 if path else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            b.run(0.0)
        b.sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ms = b.kernel_ms()
        if path:
            ms["scan_ms"] = wall            # a streamed shard's pass is bound by its copies: the wall time of a pass
        info = s.info(0)
        algo = b.stats()["algorithmic_bytes"]
        times.append(ms["scan_ms"])
        print("shard %d/%d: slots %7d  hbm %.2f GB  pages %d..%d  scan %.3f ms  hash %.3f ms  %.0f GB/s algorithmic"
              % (r, n, info.slot_count, info.hbm_bytes / 1e9, info.first_page, info.end_page - 1, ms["scan_ms"], ms["hash_ms"],
                 algo / ms["scan_ms"] / 1e6), flush=True)
        del b, s
        torch.cuda.empty_cache()
    print("sum %.3f ms  max %.3f ms  mean %.3f ms  -> balance (mean / max) %.3f" % (sum(times), max(times), sum(times) / n, sum(times) / n / max(times)))


if __name__ == "__main__":
    main()
