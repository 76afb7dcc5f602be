"""GPU: top-k WITHOUT score rows -- K2's tile_topk epilogue selects the k best documents of every tile from the
bit-sliced counters, K3 merges tiles x k candidates (kernels.hip).  Same contract as counts_to_result's
partial_sort (reference cobs/query/classic_search.cpp:127-145): score descending, ties by document ascending,
threshold first.  Every result is compared with the oracle's; the geometry sweeps put ties on both sides of tile,
chunk, sub-index and lane-group boundaries."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


def _check(b, queries, ixs, t, k, lims=None):
    for i, q in enumerate(queries):
        for lim in (lims or sorted({1, min(k, 3), k})):
            assert b.hits_host(i, lim) == cases.oracle_results(ixs, q, t, lim), (t, k, lim, i)


def test_tile_topk_matches_partial_sort(gpu_lib, oracle, tmp_path):
    """two files, planted documents, short queries (many ties), thresholds; k from 1 to the largest the tile-level
    selection takes (128), then beyond (score rows + K3 again); no score rows are kept"""
    q_long = oracle.random_sequence(700, 13)
    D = 9000
    planted = {d: f for d, f in zip(range(5, D, 211), np.linspace(0.1, 1.0, 43))}
    pa = cases.make_compact(cases.tmp(tmp_path, "tk.cobs_compact"), D, 160, [900, 1100, 1300, 1500, 1700, 1900, 2100, 2300],
                            1, 31, 1, 0.3, 21, planted=planted, query=q_long)
    pb = cases.make_classic(cases.tmp(tmp_path, "tk.cobs_classic"), 333, 1201, 1, 31, 1, 0.3, 22,
                            planted={7: 1.0, 300: 0.97}, query=q_long)
    s = gpu_lib.Search([pa, pb])
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    queries = [q_long, q_long[:36], q_long[:50], q_long[100:400], q_long[:32]]
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    for t in (0.0, 0.31, 0.9):
        for k in (1, 7, 100, 128, 129, 5000):
            b.run_topk(t, k, keep_counts=False)
            b.sync()
            _check(b, queries, ixs, t, k)
            if k <= 128:
                with pytest.raises(gpu_lib.CobsGpuError):       # the pass kept no score rows
                    b.counts_host(0)
    # over ONE file a query with a single k-mer has a single hash in total and is returned in index order, which
    # only the score rows give (classic_search.cpp:136): such a batch keeps its rows
    s1 = gpu_lib.Search(pa)
    b1 = gpu_lib.Batch(s1)
    b1.set_queries(queries + [q_long[:31]])
    b1.run_topk(0.0, 5, keep_counts=False)
    b1.sync()
    _check(b1, queries + [q_long[:31]], ixs[:1], 0.0, 5)
    assert b1.counts_host(5).shape == (s1.total_counts,)


@pytest.mark.parametrize("tile_w", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("waves", [1, 2, 4])
def test_every_tile_width_and_wave_count(gpu_lib, oracle, tmp_path, tile_w, waves):
    """ties across tile boundaries: a compact index whose 40-byte pages put tile edges inside and between
    sub-indexes for every width, short reads (8-bit scores, both work-group variants) and 400-term queries (10 planes);
    documents planted in adjacent tiles with equal scores"""
    q = oracle.random_sequence(430, 5 + tile_w + waves)
    D = 7 * 8 * 40 - 11
    planted = {d: 1.0 for d in (0, 1, 127, 128, 319, 320, 639, 640, 1023, 1024, 1025, 2047, 2048, D - 1)}
    p = cases.make_compact(cases.tmp(tmp_path, "tw.cobs_compact"), D, 40, [700, 800, 900, 1000, 1100, 1200, 1300], 1, 31, 1,
                           0.3, 6, planted=planted, query=q)
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)
    s.set_tuning("tile_w", tile_w)
    s.set_tuning("waves", waves)
    for mq in (0, 1):
        s.set_tuning("mq", mq)
        for queries in ([q, q[:200], q[3:430]], [q[i:i + 31 + 3 * i] for i in range(23)]):
            b = gpu_lib.Batch(s)
            b.set_queries(queries)
            for t, k in ((0.0, 10), (0.0, 14), (0.5, 20), (0.0, 1), (0.28, 128)):
                b.run_topk(t, k, keep_counts=False)
                b.sync()
                _check(b, queries, [ix], t, k, lims=[k])
    # the same through the host API, tile-level selection on and off
    qs = [q, q[:90], q[:31 + 17]]
    want = [cases.oracle_results([ix], x, 0.0, 12) for x in qs]
    assert s.search_hits(qs, 0.0, 12) == want
    s.set_tuning("tile_topk", 0)
    assert s.search_hits(qs, 0.0, 12) == want


def test_long_queries_and_shards_and_streaming(gpu_lib, oracle, tmp_path):
    """12 / 16 / 20 score planes; a shard cut inside a sub-index (three chunks per file: tile numbers continue across
    launches); a file streamed under a budget (many chunks)"""
    k = 31
    D = 5 * 8 * 64 - 3
    q = oracle.random_sequence(70000 + k - 1, 77)
    p = cases.make_compact(cases.tmp(tmp_path, "lg.cobs_compact"), D, 64, [3001, 3301, 3701, 4001, 4507], 1, k, 1, 0.3, 9,
                           planted={9: 1.0, 1000: 0.9, 2500: 0.95}, query=q[:3000])
    ix = oracle.Index.open(p)
    queries = [q[:3000], q[:5000], q, q[:40]]
    s = gpu_lib.Search(p)
    for nq in (2, 3, 4):            # 12, 16 and 20 planes
        assert s.search_hits(queries[:nq], 0.0, 9) == [cases.oracle_results([ix], x, 0.0, 9) for x in queries[:nq]]
    for n, r in ((3, 1), (2, 0), (7, 3)):
        sh = gpu_lib.Search(p, shard_rank=r, shard_count=n)
        lo, cnt = int(sh.info(0).slot_begin), int(sh.info(0).slot_count)
        got = sh.search_hits(queries[:2], 0.0, 25)
        for g, x in zip(got, queries[:2]):
            full = cases.oracle_results([ix], x, 0.0, 0)
            assert g == [h for h in full if lo <= h[1] < lo + cnt][:25], (n, r)
    st = gpu_lib.Search(p, hbm_budget=400 * 1000)
    b = gpu_lib.Batch(st)
    b.set_queries(queries[:2])
    b.run_topk(0.0, 6, keep_counts=False)
    b.sync()
    assert b.stats()["scan_launches"] >= 3
    _check(b, queries[:2], [ix], 0.0, 6)


@pytest.mark.parametrize("H", [2, 3])
def test_tile_topk_with_several_hash_functions(gpu_lib, oracle, tmp_path, H):
    """the generic-H scan (aggregate_rows: the H rows of a term are ANDed before counting) ends a limited pass with the same
    tile-level selection; the reference's own query tests build their indexes with three hash functions
    (tests/compact_index_query.cpp:44-47)"""
    q = oracle.random_sequence(500, 90 + H)
    p = cases.make_compact(cases.tmp(tmp_path, "h%d.cobs_compact" % H), 4 * 8 * 40 - 3, 40, [2003, 2503, 3001, 3511], H, 31, 1,
                           0.3, 6, planted={0: 1.0, 127: 1.0, 128: 1.0, 640: 0.8, 1000: 0.6, 1276: 1.0}, query=q)
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)
    queries = [q, q[:200], q[3:90], q[:40]]
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    for t, k in ((0.0, 10), (0.5, 4), (0.0, 128)):
        b.run_topk(t, k, keep_counts=False)
        b.sync()
        _check(b, queries, [ix], t, k, lims=[k])
        with pytest.raises(gpu_lib.CobsGpuError):           # no score rows were kept: the tile-level path ran
            b.counts_host(0)
    assert s.search_hits(queries, 0.0, 7) == [cases.oracle_results([ix], x, 0.0, 7) for x in queries]
