"""GPU: counts_to_result over whole score rows ON THE DEVICE (rank_kernels.hip / rank.cpp) against the
oracle's ranking (reference cobs/query/classic_search.cpp:109-202: score descending, ties by (file,
document) ascending, the unsorted single-hash case) -- the reference's default call (threshold 0, no limit;
its own benchmark, src/cobs.cpp:618-626) at 100 000 documents, every score width and pass count of the radix
sort, limits beyond K3's reach, thresholds after a hit-pool overflow, several files, shards."""
import numpy as np
import pytest

import bench
from tests import cases

pytestmark = pytest.mark.gpu


def _same(got_offs, got_hits, i, want):
    oi, od, osc = want
    seg = got_hits[int(got_offs[i]):int(got_offs[i + 1])]
    return (len(seg) == len(od) and np.array_equal(seg["file_no"], oi) and np.array_equal(seg["doc"], od)
            and np.array_equal(seg["score"], osc))


def _c3_small(scale=0.02):
    cfg = bench.c3_config(scale)
    return cfg


def _open(gpu, cfg, **kw):
    return gpu.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"],
                                term_size=cfg["term_size"], canonicalize=cfg["canonicalize"],
                                num_hashes=cfg["num_hashes"], seed=cfg["seed"], **kw)


def _oracle_index(O, cfg):
    return O.Index.synthetic(1, cfg["term_size"], cfg["canonicalize"], cfg["num_hashes"], cfg["page_size"],
                             cfg["signature_sizes"], cfg["num_docs"], cfg["seed"])


def test_default_call_ranks_100k_documents_on_the_device(gpu_lib, oracle):
    """BASELINE configs[2]'s document count: 100 000 documents, 1000-k-mer queries (10-bit scores: one pass
    straight from the score rows), 24 queries per call (two 64 MiB windows) -- every document of every query in the
    reference's order.  Binomial(1000, 0.3) scores over 100 000 documents: ~60 distinct values, thousands of ties
    per value, all broken by document id.  Same call with the device ranking off (host threads): identical."""
    cfg = _c3_small()
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(24, 1000, seed=7) + [bench.make_queries(1, 1, seed=8)[0]]   # + one single-k-mer query
    offs, hits = s.search_arrays(queries, 0.0, 0)
    assert int(offs[-1]) == len(queries) * 100000
    for i, q in enumerate(queries):
        assert _same(offs, hits, i, oracle.search_arrays(ix, q, 0.0, 0)), i
    sc = hits[:100000]["score"].astype(np.int64)
    assert (np.diff(sc) <= 0).all() and len(np.unique(sc)) < 200           # many ties
    one = hits[int(offs[24]):int(offs[25])]
    assert np.array_equal(one["doc"], np.arange(100000))                   # single hash: index order, not by score
    s.set_tuning("device_rank", 0)
    offs2, hits2 = s.search_arrays(queries, 0.0, 0)
    assert np.array_equal(offs, offs2) and np.array_equal(hits, hits2)


@pytest.mark.parametrize("terms", [12, 40, 255, 1023, 4095, 4096, 70000])
def test_every_score_width_and_pass_count(gpu_lib, oracle, tmp_path, terms):
    """4 / 8 / 10 / 12 score planes (one radix pass), 16 / 20 (two passes); 8-, 16- and 32-bit scores; two files with
    different document counts (one classic with a ragged last byte, one compact), planted documents"""
    k = 31
    q = oracle.random_sequence(terms + k - 1, 1000 + terms)
    pa = cases.make_classic(cases.tmp(tmp_path, "a.cobs_classic"), 1237, 997, 1, k, 1, 0.3, 3,
                            planted={17: 1.0, 300: 0.5, 1236: 0.5}, query=q[:400 + k - 1])
    pb = cases.make_compact(cases.tmp(tmp_path, "b.cobs_compact"), 5 * 8 * 24 - 7, 24, [509, 401, 307, 600, 450], 1, k, 1,
                            0.3, 4, planted={0: 0.5, 900: 1.0}, query=q[:400 + k - 1])
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    s = gpu_lib.Search([pa, pb])
    queries = [q, q[:len(q) // 2 + k], q[5:], q[:k], q[1:k + 3], q[:k + 1]]
    offs, hits = s.search_arrays(queries, 0.0, 0)
    for i, qq in enumerate(queries):
        assert _same(offs, hits, i, oracle.search_arrays(ixs, qq, 0.0, 0)), (terms, i)


def test_three_radix_passes(gpu_lib, oracle, tmp_path):
    """a query of more than 2^24 terms: 32 score planes, three passes of 11 bits through (score, slot) pairs"""
    k = 31
    T = (1 << 24) + 77
    q = oracle.random_sequence(T + k - 1, 4242)
    p = cases.make_classic(cases.tmp(tmp_path, "w.cobs_classic"), 203, 1511, 1, k, 1, 0.3, 3)
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)
    queries = [q, q[:5000], q[:k], q[100:9000]]
    offs, hits = s.search_arrays(queries, 0.0, 0)
    for i, qq in enumerate(queries):
        assert _same(offs, hits, i, oracle.search_arrays(ix, qq, 0.0, 0)), i


def test_limit_beyond_k3_and_thresholds_from_the_rows(gpu_lib, oracle):
    """num_results = 70 000 of 100 000 documents (more than K3 selects): the first 70 000 of the full ranking;
    threshold 0.29 over 16 queries selects ~1.1 M documents, more than the hit pool holds: the pass is repeated
    with score rows and the passing documents are ranked on the device (per-query counts differ)"""
    cfg = _c3_small()
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(16, 1000, seed=9)
    offs, hits = s.search_arrays(queries, 0.0, 70000)
    for i in (0, 7, 15):
        assert _same(offs, hits, i, oracle.search_arrays(ix, queries[i], 0.0, 70000)), i
    offs, hits = s.search_arrays(queries, 0.29, 0)
    assert int(offs[-1]) > (1 << 20)
    for i in (0, 3, 15):
        assert _same(offs, hits, i, oracle.search_arrays(ix, queries[i], 0.29, 0)), i
    # a capacity that is too small is reported with the needed size, then the retry succeeds (search_arrays does that)
    offs2, hits2 = s.search_arrays(queries[:5], 0.31, 0)
    for i in range(5):
        assert _same(offs2, hits2, i, oracle.search_arrays(ix, queries[i], 0.31, 0)), i


def test_shards_rank_their_own_documents(gpu_lib, oracle):
    """a shard (cut inside a sub-index) ranks the documents whose slots it computed: the full ranking filtered"""
    cfg = _c3_small()
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(6, 1000, seed=11)
    wants = [oracle.search_arrays(ix, q, 0.0, 0) for q in queries]
    for n, r in ((3, 1), (8, 7), (8, 0)):
        sh = _open(gpu_lib, cfg, shard_rank=r, shard_count=n)
        lo, cnt = int(sh.info(0).slot_begin), int(sh.info(0).slot_count)
        offs, hits = sh.search_arrays(queries, 0.0, 0)
        for i, (oi, od, osc) in enumerate(wants):
            keep = (od >= lo) & (od < lo + cnt)
            assert _same(offs, hits, i, (oi[keep], od[keep], osc[keep])), (n, r, i)


def test_packed_and_pair_records_agree(gpu_lib, oracle):
    """the ordered records cross PCIe as ONE u32 (score << slot_bits | slot) where slot and score fit 32 bits together
    -- 100 000 documents x 10-bit scores: 17 + 10 -- and as (slot, score) pairs otherwise; tuning key rank_pack = 0
    forces the pairs.  Same results either way, and the oracle's."""
    cfg = _c3_small()
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(9, 1000, seed=21) + bench.make_queries(3, 40, seed=22)
    offs, hits = s.search_arrays(queries, 0.0, 0)
    s.set_tuning("rank_pack", 0)
    offs2, hits2 = s.search_arrays(queries, 0.0, 0)
    assert np.array_equal(offs, offs2) and np.array_equal(hits, hits2)
    for i in (0, 8, 9, 11):
        assert _same(offs, hits, i, oracle.search_arrays(ix, queries[i], 0.0, 0)), i


def test_more_than_2_pow_22_slots_fall_back_to_pairs(gpu_lib, oracle):
    """4.2 M documents (129 sub-indexes of 32 768): a slot needs 23 bits.  With 10-bit scores (1000-k-mer queries) a
    record does not fit one word -- 8-byte pairs --, with 4-bit scores (queries of up to 15 k-mers) it does: both against
    the oracle, every document of every query"""
    ndocs = (1 << 22) + 5000
    page = 4096
    npages = (ndocs + 8 * page - 1) // (8 * page)
    sigs = [61 + 2 * (i % 7) for i in range(npages)]
    s = gpu_lib.Search.synthetic("compact", sigs, ndocs, page_size=page, seed=5)
    assert s.total_counts > (1 << 22)
    ix = oracle.Index.synthetic(1, 31, 1, 1, page, sigs, ndocs, 5)
    for kmers, seed in ((1000, 31), (12, 32)):
        queries = bench.make_queries(4, kmers, seed=seed)
        offs, hits = s.search_arrays(queries, 0.0, 0)
        assert int(offs[-1]) == 4 * ndocs
        for i in (0, 3):
            assert _same(offs, hits, i, oracle.search_arrays(ix, queries[i], 0.0, 0)), (kmers, i)


def test_slot_streams_with_score_counts_agree_with_records(gpu_lib, oracle, tmp_path, monkeypatch, capfd):
    """FULL lists (threshold 0, no limit: the reference's default call) cross PCIe as a bit stream of slots -- 17 bits per
    result at 100 000 documents -- plus the number of records per score; tuning key rank_slim = 0 sends the 4-byte records.
    Same results either way and the oracle's: one file, mixed score widths, pieces that cut the span, two files with odd
    document counts, a shard; a batch that holds a single-k-mer query (index order: no score counts) takes the records."""
    monkeypatch.setenv("COBS_GPU_TRACE", "1")
    cfg = _c3_small()
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(9, 1000, seed=41) + bench.make_queries(3, 40, seed=42)
    capfd.readouterr()
    offs, hits = s.search_arrays(queries, 0.0, 0)
    assert "slot streams + score counts" in capfd.readouterr().err
    for i in (0, 5, 8, 9, 11):
        assert _same(offs, hits, i, oracle.search_arrays(ix, queries[i], 0.0, 0)), i
    s.set_tuning("rank_slim", 0)
    offs2, hits2 = s.search_arrays(queries, 0.0, 0)
    assert "4-byte records" in capfd.readouterr().err
    assert np.array_equal(offs, offs2) and np.array_equal(hits, hits2)
    s.set_tuning("rank_slim", 1)
    s.set_tuning("rank_window_kib", 256)                  # one query per piece
    offs3, hits3 = s.search_arrays(queries, 0.0, 0)
    assert np.array_equal(offs, offs3) and np.array_equal(hits, hits3)
    capfd.readouterr()
    mixed = queries[:5] + [bench.make_queries(1, 1, seed=43)[0]]
    offs4, hits4 = s.search_arrays(mixed, 0.0, 0)
    assert "4-byte records" in capfd.readouterr().err
    assert np.array_equal(hits4[:500000], hits[:500000])
    assert np.array_equal(hits4[500000:]["doc"], np.arange(100000))
    # thresholds and limits do not take the path (their lists are not full)
    offs5, hits5 = s.search_arrays(queries[:6], 0.0, 70000)
    for i in (0, 5):
        assert _same(offs5, hits5, i, oracle.search_arrays(ix, queries[i], 0.0, 70000)), i
    # a shard: its own documents
    sh = _open(gpu_lib, cfg, shard_rank=1, shard_count=3)
    lo, cnt = int(sh.info(0).slot_begin), int(sh.info(0).slot_count)
    capfd.readouterr()
    offs6, hits6 = sh.search_arrays(queries[:6], 0.0, 0)
    assert "slot streams + score counts" in capfd.readouterr().err
    for i in (0, 5):
        oi, od, osc = oracle.search_arrays(ix, queries[i], 0.0, 0)
        keep = (od >= lo) & (od < lo + cnt)
        assert _same(offs6, hits6, i, (oi[keep], od[keep], osc[keep])), i
    # two files, 569 and 333 documents (10-bit slots, 8- and 16-bit scores)
    q_long = oracle.random_sequence(700, 9)
    pa = cases.make_compact(cases.tmp(tmp_path, "s.cobs_compact"), 3 * 8 * 24 - 7, 24, [800, 1200, 1000], 1, 31, 1, 0.3, 2,
                            planted={3: 1.0}, query=q_long)
    pb = cases.make_classic(cases.tmp(tmp_path, "s.cobs_classic"), 333, 901, 2, 31, 1, 0.4, 3)
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    two = gpu_lib.Search([pa, pb], device=0)
    for qs in ([q_long[:100], q_long[40:200], q_long[:33], q_long[7:90]], [q_long, q_long[100:500], q_long[3:640], q_long[50:400]]):
        capfd.readouterr()
        got = two.search_hits(qs, 0.0, 0)
        assert "slot streams + score counts" in capfd.readouterr().err
        assert got == [cases.oracle_results(ixs, q, 0.0, 0) for q in qs]


def test_segmented_single_pass_ranking_agrees_with_one_work_group_per_query(gpu_lib, oracle):
    """a single-pass ranking of a long row is cut into segments, a work-group each (histograms, a prefix over all
    segments' waves, scatter); tuning key rank_segments: 1 = one work-group per query, 0 = by row length (8 at 100 000
    documents).  Same lists for 1 / 2 / 3 / 8 / 64 segments: full lists, a limit beyond K3, thresholds from the rows,
    a single-k-mer query (index order), a shard whose slice is no multiple of anything"""
    cfg = _c3_small()
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(7, 1000, seed=51) + bench.make_queries(2, 40, seed=52) + [bench.make_queries(1, 1, seed=53)[0]]
    s = _open(gpu_lib, cfg)
    s.set_tuning("rank_segments", 1)
    base = {}
    for thr, lim in ((0.0, 0), (0.0, 70000), (0.31, 0)):
        base[(thr, lim)] = s.search_arrays(queries, thr, lim)
    for i in (0, 6, 8, 9):
        assert _same(*base[(0.0, 0)], i, oracle.search_arrays(ix, queries[i], 0.0, 0)), i
        assert _same(*base[(0.0, 70000)], i, oracle.search_arrays(ix, queries[i], 0.0, 70000)), i
        assert _same(*base[(0.31, 0)], i, oracle.search_arrays(ix, queries[i], 0.31, 0)), i
    for nseg in (0, 2, 3, 8, 64):
        s.set_tuning("rank_segments", nseg)
        for key, (offs0, hits0) in base.items():
            offs, hits = s.search_arrays(queries, *key)
            assert np.array_equal(offs, offs0) and np.array_equal(hits, hits0), (nseg, key)
    sh = _open(gpu_lib, cfg, shard_rank=2, shard_count=7)
    sh.set_tuning("rank_segments", 1)
    offs0, hits0 = sh.search_arrays(queries[:6], 0.0, 0)
    for nseg in (0, 5):
        sh.set_tuning("rank_segments", nseg)
        offs, hits = sh.search_arrays(queries[:6], 0.0, 0)
        assert np.array_equal(offs, offs0) and np.array_equal(hits, hits0), nseg
