#!/usr/bin/env python3
"""tests/mock_rccl/run_batch_ranks.py N [SEED] -- TEST INFRASTRUCTURE (see run_ranks.py).  The BATCH-level exchange entry
points of comm.cpp -- what bench.py's sharded flow and a torch.distributed launcher call, one process per GPU there --
with N ranks as N python threads of this process sharing GPU 0 over the stand-in communicator: every rank opens its
shard, runs the same batch and calls cobs_gpu_batch_exchange_counts in all three forms (all-gather, all-to-all to
query owners, all-reduce), _exchange_hits, _exchange_hits_owned and _exchange_topk; what each rank holds afterwards
is compared with the oracle.  Prints "ok <cases>"."""
import os
import sys
import tempfile
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("COBS_GPU_ROW_RANGE_MIN", "48")
assert "libmockrccl.so" in os.environ.get("LD_PRELOAD", ""), "run me through tests/test_gpu_mock_ranks.py (LD_PRELOAD=cobs_amd/libmockrccl.so)"

import cobs_amd  # noqa: E402
from cobs_amd import _capi  # noqa: E402
from cobs_amd.distributed import Comm  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import cases  # noqa: E402


def rank_main(r, N, uid, paths, queries, mode, budget, want, ixs, errors):
    try:
        import torch
        torch.cuda.set_device(0)
        s = cobs_amd.Search(paths if len(paths) > 1 else paths[0], device=0, shard_rank=r, shard_count=N, shard_mode=mode, hbm_budget=budget)
        comm = Comm(uid, r, N, device=0)
        assert (comm.rank, comm.size) == (r, N)
        b = cobs_amd.Batch(s)
        b.set_queries(queries)
        nq = len(queries)
        for xm in (_capi.XCHG_ALLGATHER, _capi.XCHG_ALLTOALL, _capi.XCHG_REDUCE):
            b.run(0.0)
            b.exchange_counts(comm, xm)
            b.sync()
            q0, qn, t = b.global_counts_tensor()
            if xm == _capi.XCHG_ALLTOALL:
                assert (q0, qn) == (nq * r // N, nq * (r + 1) // N - nq * r // N), (r, q0, qn)
            else:
                assert (q0, qn) == (0, nq), (r, xm, q0, qn)
            got = t.cpu().numpy().astype(np.int64) & (0xFF if t.element_size() == 1 else 0xFFFF if t.element_size() == 2 else 0xFFFFFFFF)
            assert np.array_equal(got, want[q0:q0 + qn]), (r, xm)
            for i in range(q0, q0 + qn):                  # host readers follow the global view
                assert b.hits_host(i, 4) == cases.oracle_results(ixs, queries[i], 0.0, 4), (r, xm, i)
        for t in (0.3, 0.05):
            b.run(t)
            b.sync()
            over = b.exchange_hits(comm)                  # every record to every rank
            assert over is False
            for i, q in enumerate(queries):
                assert b.hits_host(i, 0) == cases.oracle_results(ixs, q, t, 0), (r, t, i)
            b.run_hits(t)
            b.sync()
            over, q0, qn = b.exchange_hits_owned(comm)    # ... to the owner of its query
            assert (over, q0, qn) == (False, nq * r // N, nq * (r + 1) // N - nq * r // N), (r, over, q0, qn)
            for i in range(q0, q0 + qn):
                assert b.hits_host(i, 0) == cases.oracle_results(ixs, queries[i], t, 0), (r, t, i, "owned")
        for k in (1, 6, 40):
            b.run_topk(0.0, k)
            b.sync()
            b.exchange_topk(comm)
            for i, q in enumerate(queries):
                if sum((len(q) - ix.term_size + 1) * ix.num_hashes for ix in ixs) <= 1:
                    continue        # (a single hash in total: index order, needs the rows -- include/cobs_gpu_batch.h)
                got_k, want_k = b.hits_host(i, k), cases.oracle_results(ixs, q, 0.0, k)
                assert got_k == want_k, (r, k, i, len(q), got_k[:8], want_k[:8])
        del b
        comm.close()
        s.close()
    except BaseException:
        errors.append((r, traceback.format_exc()))


def main():
    N = int(sys.argv[1])
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    oracle.lib()
    rng = np.random.default_rng(77 + 1013 * N + 100003 * seed)
    tmp = tempfile.mkdtemp(prefix="cobs_mock_batch_")
    for idx in range(4):
        k = 31
        paths = []
        q_long = oracle.random_sequence(700, 9500 + idx)
        for f in range(int(rng.choice([1, 2]))):
            H = int(rng.choice([1, 2]))
            ps = int(rng.choice([16, 64, 136]))
            P = int(rng.integers(2, 9))
            D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
            sigs = [int(x) for x in rng.integers(150, 1500, size=P)]
            paths.append(cases.make_compact(os.path.join(tmp, "b%d_%d.cobs_compact" % (idx, f)), D, ps, sigs, H, k, 1, 0.3, 60 * idx + f,
                                            planted={0: 1.0, D - 1: 0.7, D // 2: 0.9}, query=q_long[:300]))
        nq = int(rng.integers(1, 12))
        queries = [q_long[o:o + int(n)] for o, n in zip(rng.integers(0, 40, size=nq), rng.choice([k, 40, 100, 300], size=nq))]
        mode = int(rng.integers(0, 3))
        ixs = [oracle.Index.open(p) for p in paths]
        want = np.stack([np.concatenate([ix.counts(q) for ix in ixs]) for q in queries]).astype(np.int64)
        uid = Comm.unique_id()
        errors = []
        threads = [threading.Thread(target=rank_main, args=(r, N, uid, paths, queries, mode, 0, want, ixs, errors)) for r in range(N)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            for r, tb in errors:
                print("rank %d:\n%s" % (r, tb), file=sys.stderr)
            raise SystemExit(1)
    print("ok 4")


if __name__ == "__main__":
    main()
