#!/usr/bin/env python3
"""tests/mock_rccl/run_ranks.py N [SEED] -- TEST INFRASTRUCTURE.  Runs in its own process with
LD_PRELOAD=cobs_amd/libmockrccl.so (tests/test_gpu_mock_ranks.py sets it; the library is the shipped libcobs_gpu.so): the device-list handle of
the C ABI over N ranks that share GPU 0 (cobs_gpu_multi_open with devices [0] * N: N worker threads, one communicator,
every search ONE collective cobs_gpu_sharded_search_batch[_split] -- multi.cpp + comm.cpp as an N-GPU node runs them)
on random tie-heavy and ordinary inputs, against the oracle.  Prints "ok <cases>" or raises."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("COBS_GPU_ROW_RANGE_MIN", "48")
assert "libmockrccl.so" in os.environ.get("LD_PRELOAD", ""), "run me through tests/test_gpu_mock_ranks.py (LD_PRELOAD=cobs_amd/libmockrccl.so)"

import cobs_amd  # noqa: E402
from cobs_amd import _capi  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import cases  # noqa: E402


def DEVICES(n):
    """the device list of the handle: ordinals 0 .. N-1 when the stand-in virtualises them (MOCK_RCCL_VIRTUAL_DEVICES, all of
    them the one GPU), else N times device 0 (the stand-in's marker symbol lets ranks share a device)"""
    return list(range(n)) if int(os.environ.get("MOCK_RCCL_VIRTUAL_DEVICES", "0")) >= n else [0] * n


def main():
    N = int(sys.argv[1])
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    oracle.lib()
    rng = np.random.default_rng(4242 + 1009 * N + 100003 * seed)
    tmp = tempfile.mkdtemp(prefix="cobs_mock_ranks_")
    done = 0
    for idx in range(10):
        k = int(rng.choice([15, 31]))
        ties = bool(rng.random() < 0.5)
        paths = []
        for f in range(int(rng.choice([1, 2]))):
            H = int(rng.choice([1, 2]))
            dens = float(rng.choice([0.3, 0.6]))
            q_long = oracle.random_sequence(700, 9000 + idx)
            if rng.random() < 0.3:
                D, S = int(rng.integers(300, 4000)), int(rng.integers(200, 1500))
                paths.append(cases.make_classic(os.path.join(tmp, "m%d_%d.cobs_classic" % (idx, f)), D, S, H, k, 1, dens, 40 * idx + f,
                                                planted={0: 1.0, D - 1: 0.7}, query=q_long[:300]))
            else:
                ps = int(rng.choice([16, 64, 136]))
                P = int(rng.integers(2, 9))
                D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
                sigs = [int(x) for x in rng.integers(150, 1500, size=P)]
                paths.append(cases.make_compact(os.path.join(tmp, "m%d_%d.cobs_compact" % (idx, f)), D, ps, sigs, H, k, 1, dens, 40 * idx + f,
                                                planted={0: 1.0, D - 1: 0.7, D // 2: 0.9}, query=q_long[:300]))
        nq = int(rng.integers(1, 14))
        if ties:       # 1..9 terms: every limit cuts through a run of equal scores
            queries = [q_long[o:o + k - 1 + int(rng.integers(1, 10))] for o in rng.integers(0, 150, size=nq)]
        else:
            queries = [q_long[o:o + int(n)] for o, n in zip(rng.integers(0, 40, size=nq), rng.choice([k, 60, 100, 300, 650], size=nq))]
        mode = int(rng.integers(0, 3))
        budget = 0
        if rng.random() < 0.3:
            budget = int(sum(os.path.getsize(p) for p in paths) * 0.7 / N) + 70000
        try:
            s = cobs_amd.MultiSearch(paths if len(paths) > 1 else paths[0], DEVICES(N), hbm_budget=budget, shard_mode=mode)
        except cobs_amd.CobsGpuError as e:
            assert budget and e.status == _capi.ERR_CAPACITY, (paths, budget, e)
            continue
        assert s.comm_size == N
        if rng.random() < 0.5:       # pass cuts (the workspace limit is a property of every rank's handle: the same on all)
            pb = int(rng.integers(1, 5)) * 2 * 2 * s.total_counts
            for r in range(N):
                s.shard(r).set_tuning("pass_bytes", pb)
        ixs = [oracle.Index.open(p) for p in paths]
        total = sum(ix.num_docs for ix in ixs)
        combos = [(0.0, 0), (float(rng.choice([0.2, 0.5, 1.0])), 0), (0.01, 0)]
        for lim in rng.choice([1, 2, 3, 5, 13, 100, total, total + 9], size=3, replace=False):
            combos.append((float(rng.choice([0.0, 0.0, 0.5])), int(lim)))
        for t, lim in combos:
            want = [cases.oracle_results(ixs, q, t, lim) for q in queries]
            got = s.search_hits(queries, t, lim)
            assert got == want, (N, idx, paths, mode, budget, t, lim)
        # a query with a character outside ACGT: every rank reports it, the call names the first one
        if len(queries) >= 2:
            bad = list(queries)
            bad[1] = bad[1][:3] + b"N" + bad[1][4:]
            try:
                s.search_hits(bad, 0.0, 3)
                raise AssertionError("an invalid query went through")
            except cobs_amd.CobsGpuError as e:
                assert e.status == _capi.ERR_INVALID_BASE and "(query 1)" in str(e), str(e)
            assert s.search_hits(queries, 0.0, 3) == [cases.oracle_results(ixs, q, 0.0, 3) for q in queries]
        s.close()
        done += 1
    assert done >= 5
    if N <= 4:
        # every rank's hit pool (1 Mi records) overflows: the ranks agree to repeat the pass with score rows and
        # exchange those (comm.cpp: `over`), the all-to-all to query owners and the shared ranking at a larger size
        ps, P = 136, 5
        D = P * 8 * ps - 37
        sigs = [700, 900, 800, 1000, 600]
        q_long = oracle.random_sequence(900, 12345)
        path = cases.make_compact(os.path.join(tmp, "big.cobs_compact"), D, ps, sigs, 1, 31, 1, 0.4, 999,
                                  planted={0: 1.0, D - 1: 0.7}, query=q_long[:300])
        nq = 260 * N
        queries = [q_long[o:o + int(n)] for o, n in zip(rng.integers(0, 200, size=nq), rng.choice([40, 60, 100, 300], size=nq))]
        ix = oracle.Index.open(path)
        s = cobs_amd.MultiSearch(path, DEVICES(N))
        for t, lim in ((0.01, 0), (0.0, 0)):
            want = [cases.oracle_results([ix], q, t, lim) for q in queries]
            assert sum(len(w) for w in want) > N * (1 << 20)
            assert s.search_hits(queries, t, lim) == want, (N, "pool overflow", t, lim)
        s.close()
        done += 1
    print("ok %d" % done)


if __name__ == "__main__":
    main()
