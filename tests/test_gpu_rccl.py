"""GPU: the native multi-GPU path of libcobs_gpu.so (cobs_amd/csrc/comm.cpp).

A one-GPU box can only host a ONE-rank RCCL communicator (RCCL refuses two ranks on one
device), but that runs the very code an 8-GPU node runs: ncclGetUniqueId / ncclCommInitRank,
the layout all-gather, ncclAllGather of the count slices typed as bytes, the strided assembly
into global rows, the sizes-first hit exchange, the top-k gather and
cobs_gpu_sharded_search_batch.  The N > 1 arithmetic (slot layouts of work- / byte-balanced shards,
assembly, merge order) is covered on the same GPU by opening every shard in turn, and across
processes by tests/test_gpu_sharded.py / tests/test_distributed_cpu.py."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(gpu_lib):
    from cobs_amd.distributed import Comm
    c = Comm(Comm.unique_id(), 0, 1, device=0)
    assert c.size == 1 and c.rank == 0             # ncclCommCount / ncclCommUserRank
    yield c
    c.close()


def _files(oracle, tmp_path):
    q_long = oracle.random_sequence(500, 17)
    pa = cases.make_compact(cases.tmp(tmp_path, "s.cobs_compact"), 2400, 64, [900, 1000, 1100, 1200, 1300], 1, 31, 1,
                            0.3, 3, planted={5: 1.0, 700: 0.9, 1500: 0.6, 2300: 0.97}, query=q_long)
    pb = cases.make_classic(cases.tmp(tmp_path, "s.cobs_classic"), 1000, 1501, 2, 31, 1, 0.3, 4,
                            planted={9: 1.0, 990: 0.8}, query=q_long)
    return [pa, pb], [q_long, q_long[:31], q_long[:250], q_long[100:340], q_long[7:60]]


def test_one_rank_exchange_is_the_real_collective(gpu_lib, oracle, tmp_path, comm):
    from cobs_amd import _capi
    paths, queries = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    want = np.stack([np.concatenate([ix.counts(q) for ix in ixs]) for q in queries])
    s = gpu_lib.Search(paths, device=0, shard_rank=0, shard_count=1)
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    for mode in (_capi.XCHG_ALLGATHER, _capi.XCHG_ALLTOALL, _capi.XCHG_REDUCE):
        b.run(0.0)
        b.exchange_counts(comm, mode)
        b.sync()
        q0, qn, t = b.global_counts_tensor()
        assert (q0, qn) == (0, len(queries))
        assert np.array_equal(t.cpu().numpy().astype(np.int64) & 0xFFFF, want)
        assert b.exchange_bytes() == 0               # nothing crosses the fabric on one rank
        for i in range(len(queries)):                # host readers follow the global view
            assert np.array_equal(b.counts_host(i), want[i])
            assert b.hits_host(i, 4) == cases.oracle_results(ixs, queries[i], 0.0, 4)
    # hit lists (sizes first) and top-k candidates
    b.run(0.3)
    b.sync()
    assert b.exchange_hits(comm) is False
    for i, q in enumerate(queries):
        assert b.hits_host(i, 0) == cases.oracle_results(ixs, q, 0.3, 0)
    # ... routed to query owners (device-side bucketing + grouped send / recv; one rank owns every query)
    for t in (0.3, 0.05):
        b.run_hits(t)
        b.sync()
        over, q0, qn = b.exchange_hits_owned(comm)
        assert (over, q0, qn) == (False, 0, len(queries))
        for i, q in enumerate(queries):
            assert b.hits_host(i, 0) == cases.oracle_results(ixs, q, t, 0)
        assert b.exchange_bytes() == 0
        # the device-side bucketing for the rank counts an 8-GPU node uses: bucket j holds exactly the records of
        # the queries rank j owns, nothing is lost or duplicated
        want_recs = sorted((i, f, d, sc) for i, q in enumerate(queries) for (f, d, sc) in cases.oracle_results(ixs, q, t, 0))
        for N in (1, 2, 3, 5, 8):
            counts, rec = b.bucketed_hits(N)
            assert sum(counts) == len(rec) == len(want_recs)
            assert sorted(map(tuple, rec.tolist())) == want_recs
            pos = 0
            for j in range(N):
                q0, q1 = len(queries) * j // N, len(queries) * (j + 1) // N
                seg = rec[pos:pos + counts[j]]
                assert ((seg[:, 0] >= q0) & (seg[:, 0] < q1)).all(), (N, j)
                pos += counts[j]
    b.run_topk(0.0, 6)
    b.sync()
    b.exchange_topk(comm)
    for i, q in enumerate(queries):
        assert b.hits_host(i, 6) == cases.oracle_results(ixs, q, 0.0, 6)


def test_sharded_search_batch_one_rank(gpu_lib, oracle, tmp_path, comm):
    paths, queries = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    s = gpu_lib.Search(paths, device=0)
    for t, lim in ((0.0, 0), (0.3, 0), (0.3, 4), (0.0, 6), (0.95, 0)):
        got = s.sharded_search_hits(comm, queries, t, lim)
        assert got == [cases.oracle_results(ixs, q, t, lim) for q in queries], (t, lim)
        # the ranks share the ranking of the all-documents search (count rows all-to-all to query owners, every rank
        # writes its queries' results at their final places); every other search is the call above
        assert s.sharded_search_hits(comm, queries, t, lim, split=True) == got, (t, lim)
    s.set_tuning("pass_bytes", 2 * s.total_counts * 2 * 2)      # two queries per pass: owners per pass, offsets continue
    want = [cases.oracle_results(ixs, q, 0.0, 0) for q in queries]
    assert s.sharded_search_hits(comm, queries, 0.0, 0, split=True) == want
    assert s.sharded_search_hits(comm, queries, 0.0, 0) == want
    s.set_tuning("pass_bytes", 0)
    # bad input is reported with the query's index, as in the single-GPU call
    from cobs_amd import _capi
    bad = list(queries)
    bad[2] = bad[2][:40] + b"N" + bad[2][41:]
    with pytest.raises(gpu_lib.CobsGpuError) as e:
        s.sharded_search_hits(comm, bad, 0.0, 3)
    assert e.value.status == _capi.ERR_INVALID_BASE and "(query 2)" in str(e.value)
    # pass cut by the workspace limit
    s.set_tuning("pass_bytes", 2 * (s.local_counts + s.total_counts))
    assert s.sharded_search_hits(comm, queries, 0.0, 0) == [cases.oracle_results(ixs, q, 0.0, 0) for q in queries]


def test_streamed_shard_under_a_budget(gpu_lib, oracle, tmp_path, comm):
    """BASELINE configs[4]: sharding x out-of-core streaming -- the shard does not fit its HBM
    budget, chunks are streamed, the exchange sees the same slices"""
    paths, queries = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    s = gpu_lib.Search(paths, device=0, hbm_budget=260 * 1024)        # the 0.5 MB compact file is streamed
    assert s.info(0).hbm_bytes <= 260 * 1024
    for t, lim in ((0.0, 0), (0.3, 0), (0.0, 5)):
        assert s.sharded_search_hits(comm, queries, t, lim) == [cases.oracle_results(ixs, q, t, lim) for q in queries]


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("budget", [0, 200 * 1024])
def test_shards_of_every_mode_assemble_to_the_whole(gpu_lib, oracle, tmp_path, mode, budget):
    """every shard of 2..5 (cuts inside sub-indexes in the work-balanced mode 0 and the byte-balanced mode 2) computes exactly its slot range;
    their concatenation is the oracle's vector.  With a budget every shard also streams."""
    q = oracle.random_sequence(500, 3)
    ratio = 16.0 ** (1.0 / 7.0)
    pc = cases.make_compact(cases.tmp(tmp_path, "sh.cobs_compact"), 8 * 8 * 48 - 9, 48,
                            [int(150 * ratio ** p) for p in range(8)], 2, 31, 1, 0.3, 8)
    pk = cases.make_classic(cases.tmp(tmp_path, "sh.cobs_classic"), 3000, 1999, 1, 31, 1, 0.3, 9)
    for p in (pc, pk):
        want = oracle.Index.open(p).counts(q)
        for n in (2, 3, 5):
            total = np.zeros_like(want)
            covered = 0
            for r in range(n):
                s = gpu_lib.Search(p, shard_rank=r, shard_count=n, shard_mode=mode, hbm_budget=budget)
                c = s.counts(q)
                i = s.info(0)
                outside = np.ones(len(c), dtype=bool)
                outside[i.slot_begin:i.slot_begin + i.slot_count] = False
                assert not c[outside].any()
                assert i.slot_begin == covered or i.slot_count == 0
                covered += i.slot_count
                total += c
                # hits of the shard's own documents, selected on the device
                if i.slot_count:
                    got = s.search_hits([q], 0.31, 0)[0]
                    ref = [h for h in cases.oracle_results([oracle.Index.open(p)], q, 0.31, 0)
                           if i.slot_begin <= h[1] < i.slot_begin + i.slot_count]
                    assert got == ref
            assert covered == len(want)
            assert np.array_equal(total, want)


def test_multi_handle_device_list(gpu_lib, oracle, tmp_path):
    """cobs_gpu_multi_*: the device list behind one handle of the C ABI (worker thread per device
    and the communicator live inside the library).  On this box the list is [0]: one rank, the
    same open / collective / close sequence as eight."""
    from cobs_amd import _capi
    paths, queries = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    m = gpu_lib.MultiSearch(paths, devices=[0])
    assert m.comm_size == 1 and m.num_files == 2
    assert m.total_counts == sum(ix.counts(queries[0]).size for ix in ixs)
    for t, lim in ((0.0, 0), (0.3, 0), (0.3, 4), (0.0, 6), (0.95, 0)):
        assert m.search_hits(queries, t, lim) == [cases.oracle_results(ixs, q, t, lim) for q in queries], (t, lim)
    # the reference's surface: names and scores of one query
    res = m.search(queries[0], 0.3)
    want = oracle.search(ixs, queries[0], 0.3)
    assert [(r.doc_name, r.score) for r in res] == [(n, s) for (_, _, n, s) in want]
    # errors come back from the worker with the query's index; the handle stays usable
    bad = list(queries)
    bad[1] = b"ACGT"
    with pytest.raises(gpu_lib.CobsGpuError) as e:
        m.search_hits(bad, 0.0, 3)
    assert e.value.status == _capi.ERR_QUERY_TOO_SHORT
    assert m.search_hits(queries[:2], 0.0, 2) == [cases.oracle_results(ixs, q, 0.0, 2) for q in queries[:2]]
    assert m.search_hits([], 0.0, 0) == []
    assert m.shard(0).info(0).slot_count == m.info(0).slot_count
    m.close()
    m.close()
    # streamed shards under a budget
    m = gpu_lib.MultiSearch(paths, devices=[0], hbm_budget=260 * 1024)
    assert m.info(0).hbm_bytes <= 260 * 1024
    assert m.search_hits(queries, 0.3, 0) == [cases.oracle_results(ixs, q, 0.3, 0) for q in queries]
    del m
    # everything that can fail on one rank alone is refused before the ranks would meet
    for devs in ([0, 0], [0, 99], [-1], []):
        with pytest.raises(gpu_lib.CobsGpuError) as e:
            gpu_lib.MultiSearch(paths, devices=devs)
        assert e.value.status == _capi.ERR_ARG
    with pytest.raises(gpu_lib.CobsGpuError):
        gpu_lib.MultiSearch(str(tmp_path / "missing.cobs_compact"), devices=[0])


def test_native_sharded_search_random_ties(gpu_lib, oracle, tmp_path, comm, monkeypatch):
    """cobs_gpu_sharded_search_batch (the collective an N-GPU node calls, here over a one-rank RCCL communicator) on
    tie-heavy inputs: queries of 1..9 terms on dense filters, one or two files, resident and streamed, random pass
    cuts, limits through runs of equal scores, thresholds, the all-documents call with and without the shared
    ranking -- against the oracle"""
    import os
    from cobs_amd import _capi
    monkeypatch.setenv("COBS_GPU_ROW_RANGE_MIN", "48")
    rng = np.random.default_rng(2718 + 100003 * int(os.environ.get("COBS_FUZZ_SEED", "0")))
    done = 0
    for idx in range(16):
        k = int(rng.choice([15, 31]))
        paths = []
        for f in range(int(rng.choice([1, 2]))):
            H = int(rng.choice([1, 2]))
            dens = float(rng.choice([0.3, 0.6]))
            if rng.random() < 0.3:
                D, S = int(rng.integers(300, 4000)), int(rng.integers(200, 1500))
                paths.append(cases.make_classic(cases.tmp(tmp_path, "n%d_%d.cobs_classic" % (idx, f)), D, S, H, k, 1, dens, 30 * idx + f))
            else:
                ps = int(rng.choice([16, 64, 136]))
                P = int(rng.integers(2, 7))
                D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
                sigs = [int(x) for x in rng.integers(150, 1500, size=P)]
                paths.append(cases.make_compact(cases.tmp(tmp_path, "n%d_%d.cobs_compact" % (idx, f)), D, ps, sigs, H, k, 1, dens, 30 * idx + f))
        q_long = oracle.random_sequence(200, 8000 + idx)
        queries = [q_long[o:o + k - 1 + int(rng.integers(1, 10))] for o in rng.integers(0, 150, size=int(rng.integers(1, 14)))]
        budget = int(sum(os.path.getsize(p) for p in paths) * 0.5) if rng.random() < 0.4 else 0
        try:
            s = gpu_lib.Search(paths, device=0, hbm_budget=budget)
        except gpu_lib.CobsGpuError as e:
            assert budget and e.status == _capi.ERR_CAPACITY, (paths, budget, e)
            continue
        if rng.random() < 0.5:
            s.set_tuning("pass_bytes", int(rng.integers(1, 5)) * (s.local_counts + s.total_counts) * 2)
        ixs = [oracle.Index.open(p) for p in paths]
        total = sum(ix.num_docs for ix in ixs)
        combos = [(0.0, 0), (float(rng.choice([0.2, 0.5, 1.0])), 0)]
        for lim in rng.choice([1, 2, 3, 5, 13, 100, total, total + 9], size=3, replace=False):
            combos.append((float(rng.choice([0.0, 0.0, 0.5])), int(lim)))
        for t, lim in combos:
            want = [cases.oracle_results(ixs, q, t, lim) for q in queries]
            assert s.sharded_search_hits(comm, queries, t, lim) == want, (paths, budget, t, lim)
            assert s.sharded_search_hits(comm, queries, t, lim, split=True) == want, (paths, budget, t, lim, "split")
        done += 1
    assert done >= 8


def test_a_collective_that_does_not_finish_in_time_fails_instead_of_hanging(gpu_lib, oracle, tmp_path):
    """round 5, on the REAL RCCL: with a time limit on the communicator (cobs_gpu_comm_set_timeout) a stream wait the
    library performs around a collective gives up -- here the stream is held by a host function queued in front of
    the all-gather of the pool fills, the stand-in for a peer that never arrives -- the communicator is aborted
    (ncclCommAbort) and says why, later calls fail at once, the batch and the index stay usable without it"""
    import time
    from cobs_amd.distributed import Comm
    paths, queries = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    s = gpu_lib.Search(paths, device=0, shard_rank=0, shard_count=1)
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    c = Comm(Comm.unique_id(), 0, 1, device=0)
    assert "communicator ok" in c.state()
    b.run(0.3)
    b.sync()
    assert b.exchange_hits(c) is False                      # healthy: no limit needed
    assert "returned" in c.state() and "INSIDE" not in c.state()
    c.set_timeout(150)
    st = torch.cuda.Stream()
    b.run(0.3)
    b.sync()
    # (a host function queued on the stream sleeps 2 s: tests/mock_rccl/stall.cpp, built here with g++)
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libstall.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(root, "tests", "mock_rccl", "stall.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64",
                           "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    stall = ctypes.CDLL(so)
    stall.stall_stream.argtypes = [ctypes.c_void_p, ctypes.c_long]
    t0 = time.time()
    assert stall.stall_stream(st.cuda_stream, 2000) == 0
    with pytest.raises(gpu_lib.CobsGpuError) as e:
        b.exchange_hits(c, st.cuda_stream)
    waited = time.time() - t0
    assert e.value.status == 11 and "did not complete within 150 ms" in str(e.value), str(e.value)
    assert waited < 10.0
    assert "BROKEN" in c.state()
    torch.cuda.synchronize()
    with pytest.raises(gpu_lib.CobsGpuError) as e2:
        b.exchange_hits(c)
    assert "unusable after an earlier failure" in str(e2.value)
    c.close()
    # the batch goes on with a fresh communicator
    c2 = Comm(Comm.unique_id(), 0, 1, device=0)
    b.run(0.3)
    b.sync()
    assert b.exchange_hits(c2) is False
    for i, q in enumerate(queries):
        assert b.hits_host(i, 0) == cases.oracle_results(ixs, q, 0.3, 0)
    pf = c2.preflight(timeout_ms=20000)                     # the real collectives of a one-rank communicator
    assert pf["allgather_us"] > 0 and pf["allreduce_us"] > 0
    c2.close()


def test_sharded_search_is_a_pipeline_of_passes(gpu_lib, oracle, tmp_path, comm):
    """round 6: cobs_gpu_sharded_search_batch cuts a call into passes that overlap inside the library (upload + K1 of
    pass i+1 | K2 of pass i | agreement, exchange and ordering of pass i-1 on the exchange stream; three scratch
    batches in rotation; ONE all-gathered status record per pass).  A call of 1..9 passes, every result path -- hits
    through the pool, a limit through the shards' best-of lists (and through score rows where a query has a single hash),
    every document through the row exchange, the shared ranking -- against the one-GPU call and the oracle; an invalid
    query in a LATER pass is named by its index in the call and leaves nothing in flight (the next call works); a
    result buffer that is too small reports the needed size."""
    from cobs_amd import _capi
    import ctypes as C
    paths, base = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    q_long = base[0]
    queries = [q_long[(7 * i) % 200:(7 * i) % 200 + 31 + (37 * i) % 260] for i in range(23)]
    queries[4] = q_long[:31]                          # a single hash in the compact file: index order, rows travel
    s = gpu_lib.Search(paths, device=0)
    cases_ = ((0.0, 0), (0.3, 0), (0.9, 0), (0.3, 4), (0.0, 6), (0.0, 1))
    for per_pass in (0, 9, 3):                        # one pass | three passes | eight passes (the rotation wraps twice)
        s.set_tuning("pass_bytes", 0 if per_pass == 0 else per_pass * 2 * s.total_counts * 2)
        for t, lim in cases_:
            want = [cases.oracle_results(ixs, q, t, lim) for q in queries]
            assert s.sharded_search_hits(comm, queries, t, lim) == want, (per_pass, t, lim)
            assert s.search_hits(queries, t, lim) == want, (per_pass, t, lim)
            if lim == 0:
                assert s.sharded_search_hits(comm, queries, t, lim, split=True) == want, (per_pass, t, lim, "split")
        # an invalid character in the last pass: every earlier pass has been launched, agreed on and exchanged by then
        bad = list(queries)
        bad[21] = bad[21][:12] + b"N" + bad[21][13:]
        for t, lim in ((0.0, 0), (0.3, 0), (0.0, 3)):
            with pytest.raises(gpu_lib.CobsGpuError) as e:
                s.sharded_search_hits(comm, bad, t, lim)
            assert e.value.status == _capi.ERR_INVALID_BASE and "(query 21)" in str(e.value), str(e.value)
            assert s.sharded_search_hits(comm, queries[:5], t, lim) == [cases.oracle_results(ixs, q, t, lim) for q in queries[:5]]
        # a query shorter than the term size (host-side check of a later pass)
        short = list(queries)
        short[20] = b"ACGT"
        with pytest.raises(gpu_lib.CobsGpuError) as e:
            s.sharded_search_hits(comm, short, 0.3, 0)
        assert e.value.status == _capi.ERR_QUERY_TOO_SHORT and "(query 20)" in str(e.value), str(e.value)
    # capacity: the call says what it needs, nothing is written past the buffer
    lib = _capi.load()
    nq = len(queries)
    arr = (C.c_char_p * nq)(*queries)
    lens = (C.c_size_t * nq)(*[len(q) for q in queries])
    offs = (C.c_size_t * (nq + 1))()
    hits = (_capi.Hit * 8)()
    bad_q = C.c_size_t(0)
    want = [cases.oracle_results(ixs, q, 0.3, 0) for q in queries]
    st = lib.cobs_gpu_sharded_search_batch(s._h, comm._h, arr, lens, nq, 0.3, 0, hits, 8, offs, C.byref(bad_q))
    assert st == _capi.ERR_CAPACITY and offs[nq] == sum(len(w) for w in want) > 8


def test_sharded_batch_object(gpu_lib, oracle, tmp_path, comm):
    """cobs_gpu_sharded_batch_*: the device-resident form of the sharded search that bench.py --gpus N times -- the batch
    cut into sub-batches whose hashing, scan and exchange overlap on the library's own three streams, also across
    steps.  1, 2, 3 and 7 sub-batches (more than some have queries), every exchange mode, several steps in flight:
    the rows every sub-batch ends up with equal the oracle's, the times and byte counts are there, an invalid query is
    reported by its index in the batch."""
    from cobs_amd import _capi
    paths, base = _files(oracle, tmp_path)
    ixs = [oracle.Index.open(p) for p in paths]
    q_long = base[0]
    queries = [q_long[(11 * i) % 150:(11 * i) % 150 + 40 + (53 * i) % 300] for i in range(13)]
    want = np.stack([np.concatenate([ix.counts(q) for ix in ixs]) for q in queries])
    s = gpu_lib.Search(paths, device=0)
    for nsub in (1, 2, 3, 7):
        sb = gpu_lib.ShardedBatch(s, comm, nsub)
        sb.set_queries(queries)
        assert len(sb.subs) == nsub and sum(b.nq for b in sb.subs) == len(queries)
        for mode in (_capi.XCHG_ALLTOALL, _capi.XCHG_ALLGATHER, _capi.XCHG_REDUCE):
            for _ in range(3):
                sb.step(0.0, mode)
            sb.sync()
            for b in sb.subs:
                if b.nq == 0:
                    continue
                q0, qn, rows = b.global_counts_tensor()
                assert (q0, qn) == (0, b.nq)
                got = rows.cpu().numpy()          # (u8 / u16 scores by the sub-batch's longest query, as signed torch types)
                got = got.astype(np.int64) & ((1 << (8 * got.itemsize)) - 1)
                assert np.array_equal(got, want[b.q_begin:b.q_begin + b.nq]), (nsub, mode)
            t = sb.times()
            assert t["steps"] == 3 and t["scan_ms"] > 0 and t["exchange_ms"] > 0 and t["received_bytes"] == 0
            assert t["algorithmic_bytes"] == sum(b.stats()["algorithmic_bytes"] for b in sb.subs)
        bad = list(queries)
        bad[9] = bad[9][:5] + b"N" + bad[9][6:]
        sb.set_queries(bad)
        sb.step(0.0)
        with pytest.raises(gpu_lib.CobsGpuError) as e:
            sb.sync()
        assert e.value.status == _capi.ERR_INVALID_BASE and "(query 9)" in str(e.value), str(e.value)
        sb.close()
