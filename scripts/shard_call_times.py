#!/usr/bin/env python3
"""scripts/shard_call_times.py [N] -- what ONE rank of an N-GPU node spends in the PRODUCT call of the multi-GPU layout:
every shard of the N-way split of C3 opened in turn on this GPU, cobs_gpu_sharded_search_batch over a one-rank
communicator (everything of the call but the xGMI transfers: upload of the WHOLE batch's text -- every rank hashes every
query --, K1, the shard's scan, the agreement, the ordering of its hits, the hand-over), for the planted batch at
threshold 0.8, the random batch, and a limit of 10.  The step of the node is the slowest rank.  Timers: the library's own
(host staging | K1 | scan | results)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402
from cobs_amd.distributed import Comm  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    cfg = bench.c3_config()
    cfg["plants"] = bench.planted_documents(cfg, 1000)
    comm = Comm(Comm.unique_id(), 0, 1, 0)
    rand = bench.pack_queries(bench.make_queries(nq, 1000))
    hitq = bench.pack_queries(bench.planted_queries(cfg["plants"], nq, 1000))
    cases = (("threshold 0.8, planted", hitq, 0.8, 0), ("threshold 0.8, random", rand, 0.8, 0), ("limit 10", rand, 0.0, 10))
    worst = {c[0]: 0.0 for c in cases}
    print("# C3 in %d shards (work-balanced), %d queries x 1000 k-mers, best of 5 calls; ms per call | library timers of that call (ms)" % (n, nq))
    for r in range(n):
        s = bench.make_index(cfg, 0, r, n)
        line = "shard %d/%d:" % (r, n)
        for name, pk, thr, k in cases:
            s.sharded_search_arrays(comm, pk, thr, k)
            best, tm = None, None
            for _ in range(5):
                s.timers(reset=True)
                t0 = time.perf_counter()
                o, h = s.sharded_search_arrays(comm, pk, thr, k)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, tm = dt, s.timers()
            worst[name] = max(worst[name], best)
            line += "  %s %.3f ms (staging %.2f K1 %.2f scan %.2f results %.2f; %d records)" % (
                name, best * 1e3, tm["h2d"] * 1e3, tm["hashes"] * 1e3, tm["scan"] * 1e3, tm["rank"] * 1e3, len(h))
        print(line, flush=True)
        s.close()
    for name, _, _, _ in cases:
        print("slowest rank, %s: %.3f ms -> %.0f queries/s for the node (without the xGMI transfers)" % (name, worst[name] * 1e3, nq / worst[name]))


if __name__ == "__main__":
    main()
