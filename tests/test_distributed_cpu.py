"""world_size-2 gloo tests (CPU) of the multi-GPU exchange: count slices of
disjoint sub-index blocks are gathered and assembled into the global vector, hit
lists are merged into the reference's result order.  The per-rank count slices
are produced here by the oracle restricted to the rank's documents (on a GPU box
they come from libcobs_gpu with shard_rank/shard_count; see
tests/test_gpu_parity.py::test_sharded_counts_sum_to_whole)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, queries, out_dir, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cobs_amd import distributed as D
        from oracle import oracle as O
        ix = O.Index.open(path)
        # the engine's own shard layout (host-side planner of libcobs_gpu.so, no device needed):
        # work-balanced cuts (mode 0) may fall inside a sub-index
        import ctypes as C
        from cobs_amd import _capi
        b_, c_ = (C.c_uint64 * world)(), (C.c_uint64 * world)()
        _capi.check(_capi.load().cobs_gpu_plan_shards(path.encode(), world, mode, b_, c_, None))
        begin, count = int(b_[rank]), int(c_[rank])
        full = np.stack([ix.counts(q) for q in queries]).astype(np.uint16)
        local = torch.from_numpy(full[:, begin:begin + count].astype(np.int16))
        layouts = [None] * world
        dist.all_gather_object(layouts, [(begin, count, 0)])
        gathered = D.all_gather_counts(local)
        assert [g.shape[1] for g in gathered] == [l[0][1] for l in layouts]
        whole = D.assemble_counts(gathered, layouts, ix.counts_size)
        assert np.array_equal(whole.numpy().astype(np.uint16), full)
        # COBS_GPU_XCHG_REDUCE (comm.cpp): zero-padded rows of global length, summed as BYTES --
        # disjoint slices leave one non-zero addend per byte, so the u16 counters come out exact
        padded = np.zeros_like(full)
        padded[:, begin:begin + count] = full[:, begin:begin + count]
        as_bytes = torch.from_numpy(padded.view(np.uint8).copy())
        dist.all_reduce(as_bytes, op=dist.ReduceOp.SUM)
        assert np.array_equal(as_bytes.numpy().view(np.uint16), full)
        # the library's OWN exchange plans (cobs_gpu_exchange_plan / cobs_gpu_hit_exchange_plan, what comm.cpp executes
        # over RCCL) run across these processes with point-to-point transfers: all-to-all to query owners and
        # all-gather for the count rows, owner-routed hit records
        nq = len(queries)
        for xmode, name in ((_capi.XCHG_ALLTOALL, "alltoall"), (_capi.XCHG_ALLGATHER, "allgather")):
            q0, qn, rows = D.exchange_counts_by_plan(local, layouts, ix.counts_size, nq, xmode)
            want_q = (nq * rank // world, nq * (rank + 1) // world - nq * rank // world) if xmode == _capi.XCHG_ALLTOALL else (0, nq)
            assert (q0, qn) == want_q, name
            assert np.array_equal(rows.numpy().astype(np.uint16), full[q0:q0 + qn]), name
        thr = [int(np.ceil(0.3 * (len(q) - 30))) for q in queries]
        mine_hits = [(qi, 0, d, int(full[qi, d])) for qi in range(nq) for d in range(begin, min(begin + count, ix.num_docs))
                     if full[qi, d] >= thr[qi]]
        q0, qn, recs = D.exchange_hits_by_plan(mine_hits, nq)
        want_recs = sorted((qi, f, d, sc) for qi in range(q0, q0 + qn)
                           for (f, d, sc) in cases.oracle_results([ix], queries[qi], 0.3, 0))
        assert sorted(recs) == want_recs
        # hits mode: per-shard ranked lists -> global order
        for t, lim in ((0.0, 0), (0.3, 0), (0.3, 3), (0.9, 0)):
            for qi, q in enumerate(queries):
                T = len(q) - 30
                thr = int(np.ceil(t * T))
                mine = [(0, d, int(full[qi, d])) for d in range(begin, min(begin + count, ix.num_docs))
                        if full[qi, d] >= thr]
                mine.sort(key=lambda h: (-h[2], h[1]))
                if lim:
                    mine = mine[:lim]
                everyone = [None] * world
                dist.all_gather_object(everyone, mine)
                merged = D.merge_hits(everyone, lim, T * ix.num_hashes, ix.counts_size)
                assert merged == cases.oracle_results([ix], q, t, lim), (t, lim, qi)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, 0), (3, 0), (2, 1), (3, 1)])
def test_shard_exchange_gloo(oracle, tmp_path, world, mode):
    q_long = oracle.random_sequence(400, 21)
    planted = {3: 1.0, 200: 0.8, 777: 0.5, 1100: 0.95}
    path = cases.make_compact(cases.tmp(tmp_path, "d.cobs_compact"), 1200, 32, [600, 700, 800, 900, 1000], 1, 31, 1,
                              0.3, 4, planted=planted, query=q_long)
    queries = [q_long, q_long[:31], q_long[:200]]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, path, queries, str(tmp_path), mode), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(os.path.join(str(tmp_path), "ok%d" % r))


def test_merge_hits_order():
    from cobs_amd.distributed import merge_hits
    a = [(0, 5, 9), (0, 1, 3)]
    b = [(0, 9, 9), (1, 0, 9), (0, 7, 1)]
    assert merge_hits([a, b]) == [(0, 5, 9), (0, 9, 9), (1, 0, 9), (0, 1, 3), (0, 7, 1)]
    assert merge_hits([a, b], 2) == [(0, 5, 9), (0, 9, 9)]
    assert merge_hits([a, b], 0, total_hashes=1) == [(0, 1, 3), (0, 5, 9), (0, 7, 1), (0, 9, 9), (1, 0, 9)]


def test_split_search_segments_on_a_rank_that_does_not_own_query_0():
    """cobs_gpu_sharded_search_batch_split with ranks in separate processes: a rank writes hit_offsets[q + 1] of the
    queries it owns only, so the start of its first owned query is NOT in its array (the wrapper sliced
    hits[offs[q]:offs[q + 1]] and returned uninitialised memory in front of the real rows).  The offsets a rank 1 of
    2 sees for 4 queries x 5 documents: entries 3 and 4 only."""
    import numpy as np
    from cobs_amd.search import Search, _split_segments

    class Info:
        num_docs = 5

    class FakeSearch:
        total_counts = 8
        num_files = 1

        def info(self, f):
            return Info()

    hits = np.zeros(20, dtype=Search.HIT_DTYPE)
    hits["doc"] = 777                                   # what np.empty could hold where nothing was written
    for q in (2, 3):
        for i in range(5):
            hits[q * 5 + i] = (0, i, 10 * q + 5 - i)
    offs = np.array([0, 0, 0, 15, 20], dtype=np.uint64)
    got = _split_segments(hits, offs, 4, 0.0, 0, FakeSearch())
    assert got[0] == [] and got[1] == []
    assert got[2] == [(0, i, 25 - i) for i in range(5)] and got[3] == [(0, i, 35 - i) for i in range(5)]
    # one process holding all ranks: every entry is filled, same slicing
    offs_all = np.array([0, 5, 10, 15, 20], dtype=np.uint64)
    assert [len(x) for x in _split_segments(hits, offs_all, 4, 0.0, 0, FakeSearch())] == [5, 5, 5, 5]
    # with a threshold or a limit the call is not shared: complete offsets on every rank
    offs_t = np.array([0, 1, 1, 3, 3], dtype=np.uint64)
    assert [len(x) for x in _split_segments(hits, offs_t, 4, 0.5, 0, FakeSearch())] == [1, 0, 2, 0]
    assert [len(x) for x in _split_segments(hits, offs_t, 4, 0.0, 2, FakeSearch())] == [1, 0, 2, 0]
