#!/usr/bin/env python3
"""tests/mock_rccl/run_bench_ranks.py N -- TEST INFRASTRUCTURE (see run_ranks.py).  bench.py's sharded flow itself
(class ShardedRun: sub-batches, K1 on the batch's own stream, K2 on the scan stream, the native exchange on a third
stream, events across steps) with N ranks as N python threads over the stand-in communicator: after three steps every
rank's assembled rows -- the all-to-all to query owners -- are compared element by element with the oracle's."""
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
assert "libmockrccl.so" in os.environ.get("LD_PRELOAD", ""), "run me through tests/test_gpu_mock_ranks.py (LD_PRELOAD=cobs_amd/libmockrccl.so)"

import bench  # noqa: E402
from cobs_amd.distributed import Comm  # noqa: E402
from oracle import oracle  # noqa: E402


def rank_main(r, N, uid, cfg, queries, nsub, want, errors):
    try:
        import torch
        torch.cuda.set_device(0)
        comm = Comm(uid, r, N, device=0)
        run = bench.ShardedRun(cfg, queries, N, r, 0, comm, nsub=nsub)
        for _ in range(3):
            run.step()
        scan_ms, hash_ms, xchg_ms, algo, moved = run.finish()
        assert scan_ms > 0 and (N == 1 or moved > 0), (r, scan_ms, moved)
        seen = 0
        for i in range(len(run.sub)):
            base = i * len(queries) // len(run.sub)
            q0, qn, rows = run.owned_rows(i)
            got = bench._as_int64(rows).cpu().numpy()
            assert np.array_equal(got, want[base + q0:base + q0 + qn]), (r, i, q0, qn)
            seen += qn
        assert seen > 0 or len(queries) < N
        comm.close()
    except BaseException:
        errors.append((r, traceback.format_exc()))


def main():
    N = int(sys.argv[1])
    oracle.build()
    oracle.lib()
    cfg = bench.c3_config(0.01)
    queries = bench.make_queries(61, 200, seed=5)            # not a multiple of the rank count or of the sub-batches
    ix = oracle.Index.synthetic(1, cfg["term_size"], cfg["canonicalize"], cfg["num_hashes"], cfg["page_size"],
                                cfg["signature_sizes"], cfg["num_docs"], cfg["seed"])
    want = np.stack([ix.counts(q) for q in queries]).astype(np.int64)
    for nsub in (1, 2, 3):
        uid = Comm.unique_id()
        errors = []
        threads = [threading.Thread(target=rank_main, args=(r, N, uid, cfg, queries, nsub, want, errors)) for r in range(N)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            for r, tb in errors:
                print("rank %d (nsub %d):\n%s" % (r, nsub, tb), file=sys.stderr)
            raise SystemExit(1)
    # the HIT path with real record counts (round 5): the index with planted documents, queries that are mutated windows
    # of the planted sequences, the hits-only scan at the CLI's default threshold, the sizes-first exchange of the records --
    # to every rank, and routed to query owners -- and the pools put into result order on the device
    cfg = bench.c3_config(0.01)
    cfg["plants"] = bench.planted_documents(cfg, 200, n_seq=8, docs_per_seq=24)
    hq = bench.planted_queries(cfg["plants"], 61, 200)
    gen = bench.oracle_index(cfg, cfg["plants"])
    want_hits = [[(f, d, sc) for (f, d, _n, sc) in oracle.search(gen, q, 0.8)] for q in hq]
    assert sum(len(w) for w in want_hits) > 5 * len(hq)
    uid = Comm.unique_id()
    errors = []

    def hits_main(r):
        try:
            import torch
            import cobs_amd
            torch.cuda.set_device(0)
            comm = Comm(uid, r, N, device=0)
            s = bench.make_index(cfg, 0, r, N)
            b = cobs_amd.Batch(s)
            b.set_queries(hq)
            for _ in range(2):
                b.run_hits(0.8)
                b.sync()
                over, q0, qn = b.exchange_hits_owned(comm)
                assert not over and (q0, qn) == (len(hq) * r // N, len(hq) * (r + 1) // N - len(hq) * r // N)
                assert N == 1 or qn == 0 or b.exchange_bytes() > 0
                for i in range(q0, q0 + qn):
                    assert b.hits_host(i, 0) == want_hits[i], (r, i)
                    assert b.hits_host(i, 3) == want_hits[i][:3], (r, i)
                b.run(0.8)
                b.sync()
                assert b.exchange_hits(comm) is False
                for i in range(len(hq)):
                    assert b.hits_host(i, 0) == want_hits[i], (r, i, "to every rank")
            del b
            comm.close()
            s.close()
        except BaseException:
            errors.append((r, traceback.format_exc()))

    threads = [threading.Thread(target=hits_main, args=(r,)) for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        for r, tb in errors:
            print("rank %d (hits):\n%s" % (r, tb), file=sys.stderr)
        raise SystemExit(1)
    print("ok 4")


if __name__ == "__main__":
    main()
