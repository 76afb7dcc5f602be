"""GPU parity: libcobs_gpu.so (HIP, through the C ABI) against the oracle, bit-exact.

Integer path: every per-document count and every ranked result must be
identical (no tolerance).  Edge cases follow the reference's own tests
(tests/classic_index_query.cpp, tests/compact_index_query.cpp) and SURVEY 8c.
"""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

Q50 = b"AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"


def _check_index(gpu, O, paths, queries, thresholds=(0.0, 0.3, 0.8, 1.0), limits=(0, 1, 5)):
    paths = paths if isinstance(paths, list) else [paths]
    s = gpu.Search(paths if len(paths) > 1 else paths[0])
    ixs = [O.Index.open(p) for p in paths]
    for q in queries:
        want = np.concatenate([ix.counts(q) for ix in ixs])
        got = s.counts(q)
        assert got.dtype == np.uint32 and got.shape == want.shape
        assert np.array_equal(got, want), "counts differ for query of %d chars" % len(q)
    for t in thresholds:
        for lim in limits:
            got = s.search_hits(queries, t, lim)
            for q, g in zip(queries, got):
                assert g == cases.oracle_results(ixs, q, t, lim), (t, lim, len(q))
    return s


def test_c1_fixture_known_answers(gpu_lib, oracle, golden_dir):
    """BASELINE config 1 / reference python/tests/test_cobs_index.py:22-61"""
    for name in ("c1.cobs_classic", "c1.cobs_compact"):
        s = gpu_lib.Search(os.path.join(golden_dir, name))
        r = s.search(Q50.decode())
        assert len(r) == 7
        assert r[0].doc_name == "sample1" and r[0].score == 20
        assert [(x.doc_name, x.score) for x in r] == [
            ("sample1", 20), ("sample7", 3), ("sample2", 1), ("sample4", 1), ("sample6", 1),
            ("sample3", 0), ("sample5", 0)]
        assert list(s.counts(Q50)[:8]) == [20, 1, 0, 1, 0, 1, 3, 0]
        # threshold = ceil(threshold * T) in double
        assert len(s.search(Q50, 0.05)) == 5
        assert len(s.search(Q50, 0.051)) == 2
        assert [(x.doc_name, x.score) for x in s.search(Q50, 0.15)] == [("sample1", 20), ("sample7", 3)]
        assert len(s.search(Q50, 0.1500001)) == 1
        # single k-mer, one hash: no sorting, document order
        assert [(x.doc_name, x.score) for x in s.search(Q50[5:36])] == [
            ("sample1", 1), ("sample2", 0), ("sample3", 0), ("sample4", 0), ("sample5", 0),
            ("sample6", 0), ("sample7", 1)]
    _check_index(gpu_lib, oracle, os.path.join(golden_dir, "c1.cobs_classic"), [Q50, Q50[:31], Q50[3:40]])
    _check_index(gpu_lib, oracle, os.path.join(golden_dir, "c1.cobs_compact"), [Q50, Q50[:31], Q50[3:40]])


def test_c1_two_indexes_ordering(gpu_lib, oracle, golden_dir):
    """several -i files: ties by (index number, document) (classic_search.cpp:158-201)"""
    paths = [os.path.join(golden_dir, "c1.cobs_compact"), os.path.join(golden_dir, "c1.cobs_classic")]
    s = _check_index(gpu_lib, oracle, paths, [Q50, Q50[:33]])
    assert [(x.doc_name, x.score) for x in s.search(Q50)] == [
        ("sample1", 20), ("sample1", 20), ("sample7", 3), ("sample7", 3), ("sample2", 1),
        ("sample4", 1), ("sample6", 1), ("sample2", 1), ("sample4", 1), ("sample6", 1),
        ("sample3", 0), ("sample5", 0), ("sample3", 0), ("sample5", 0)]


@pytest.mark.parametrize("num_hashes", [1, 3])
@pytest.mark.parametrize("canonicalize", [0, 1])
def test_classic_random_bits(gpu_lib, oracle, tmp_path, num_hashes, canonicalize):
    """hand-built classic indexes, D not a multiple of 8 / of 128, all plane widths"""
    for D, S, seed in ((7, 997, 1), (130, 5003, 2), (1001, 4099, 3), (10007, 2053, 4)):
        q_long = oracle.random_sequence(1030, 100 + seed)
        p = cases.make_classic(cases.tmp(tmp_path, "r%d.cobs_classic" % D), D, S, num_hashes, 31,
                               canonicalize, 0.3, seed,
                               planted={0: 1.0, D // 2: 0.85, D - 1: 0.5}, query=q_long)
        # T = 1, 8, 9, 20, 255, 256, 270, 1000 cross the u8/u16 boundary and the plane counts
        queries = [q_long[:31], q_long[:38], q_long[:39], q_long[:50], q_long[:285], q_long[:286],
                   q_long[:300], q_long]
        _check_index(gpu_lib, oracle, p, queries)


@pytest.mark.parametrize("page_size,pages", [(2, 5), (8, 3), (16, 4), (24, 3), (64, 2), (200, 3)])
def test_compact_random_bits(gpu_lib, oracle, tmp_path, page_size, pages):
    """variable signature size per sub-index, page sizes that are not multiples of 16"""
    rng = np.random.default_rng(page_size)
    sigs = [int(x) for x in rng.integers(300, 3000, size=pages)]
    D = (pages - 1) * 8 * page_size + max(1, 8 * page_size - 3)
    q_long = oracle.random_sequence(600, 7 + page_size)
    planted = {0: 1.0, D - 1: 0.9, D // 2: 0.4}
    for H in (1, 2):
        p = cases.make_compact(cases.tmp(tmp_path, "c%d_%d.cobs_compact" % (page_size, H)), D, page_size,
                               sigs, H, 31, 1, 0.3, page_size, planted=planted, query=q_long)
        _check_index(gpu_lib, oracle, p, [q_long[:31], q_long[:131], q_long[:400], q_long])


def test_other_term_sizes(gpu_lib, oracle, tmp_path):
    """k below 8, even k, k >= 32 (XXH64 stripe loop), k = 64 and beyond (term_size is a uint32 in the file format)"""
    for k in (3, 12, 20, 32, 33, 47, 64, 65, 100, 131):
        q = oracle.random_sequence(200, k)
        p = cases.make_classic(cases.tmp(tmp_path, "k%d.cobs_classic" % k), 77, 1009, 2, k, 1, 0.3, k,
                               planted={5: 1.0, 70: 0.7}, query=q)
        _check_index(gpu_lib, oracle, p, [q[:k], q[:k + 1], q[:k + 30], q], thresholds=(0.0, 0.5),
                     limits=(0, 3))


def test_reference_corpora(gpu_lib, oracle, construct, tmp_path):
    """the corpora of tests/classic_index_query.cpp:36-111 and
    tests/compact_index_query.cpp:36-181, built by the construction restatement"""
    query = oracle.random_sequence(21000, 1)
    docs = construct.generate_documents_all(query)
    pc = cases.tmp(tmp_path, "all.cobs_classic")
    construct.classic_construct(docs, pc, num_hashes=3, false_positive_rate=0.1)
    pk = cases.tmp(tmp_path, "all.cobs_compact")
    construct.compact_construct(docs, pk, num_hashes=3, false_positive_rate=0.1, page_size=2)
    for p in (pc, pk):
        s = _check_index(gpu_lib, oracle, p, [query, query[:160]], thresholds=(0.0, 0.5), limits=(0, 4))
        res = s.search(query.decode())
        assert len(res) == len(docs)
        for r in res:       # ASSERT_GE(r.score, documents[index].data().size())
            assert r.score >= docs[int(r.doc_name[-2:])].num_terms
    one = construct.generate_documents_one(query, 2000)
    po = cases.tmp(tmp_path, "one.cobs_classic")
    construct.classic_construct(one, po, num_hashes=3, false_positive_rate=0.1)
    pko = cases.tmp(tmp_path, "one.cobs_compact")
    construct.compact_construct(one, pko, num_hashes=3, false_positive_rate=0.1, page_size=2)
    for p in (po, pko):
        s = _check_index(gpu_lib, oracle, p, [query[:5000]], thresholds=(0.0,), limits=(0,))
        res = s.search(query.decode())
        assert len(res) == 2000 and all(r.score == 1 for r in res)


def test_multi_index_one_included(gpu_lib, oracle, construct, tmp_path):
    """tests/classic_index_query.cpp:156-197: 33 + 44 + 55 documents, all scores 1"""
    query = oracle.random_sequence(5000, 2)
    paths = []
    for i, n in enumerate((33, 44, 55)):
        p = cases.tmp(tmp_path, "m%d.cobs_classic" % i)
        construct.classic_construct(construct.generate_documents_one(query, n), p, num_hashes=3,
                                    false_positive_rate=0.1)
        paths.append(p)
    s = _check_index(gpu_lib, oracle, paths, [query], thresholds=(0.0, 0.0001), limits=(0, 40))
    res = s.search(query.decode())
    assert len(res) == 33 + 44 + 55 and all(r.score == 1 for r in res)


def test_ragged_batch_and_device_selection(gpu_lib, oracle, tmp_path):
    """one device pass over queries of different lengths; on-device threshold
    selection (count >= ceil(t*T)) equals the oracle's filter + ranking"""
    D, ps, sigs = 5000, 128, [1500, 2100, 2900, 4001, 5003]
    q_long = oracle.random_sequence(1500, 9)
    planted = {d: f for d, f in zip(range(0, D, 97), np.linspace(0.05, 1.0, 52))}
    p = cases.make_compact(cases.tmp(tmp_path, "rag.cobs_compact"), D, ps, sigs, 1, 31, 1, 0.3, 5,
                           planted=planted, query=q_long)
    s = gpu_lib.Search(p)
    ix = oracle.Index.open(p)
    queries = [q_long[:31], q_long[:1030], q_long[:100], q_long, q_long[200:531], q_long[:255 + 30],
               q_long[:256 + 30]] * 3
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    for t in (0.0, 0.2, 0.7):
        b.run(t)
        b.sync()
        for i, q in enumerate(queries):
            assert np.array_equal(b.counts_host(i), ix.counts(q))
            for lim in (0, 3):
                assert b.hits_host(i, lim) == cases.oracle_results([ix], q, t, lim)
    ms = b.kernel_ms()
    assert ms["scan_ms"] > 0 and ms["hash_ms"] > 0
    st = b.stats()
    T = sum(len(q) - 30 for q in queries)
    assert st["kmer_lookups"] == T
    assert st["algorithmic_bytes"] == T * len(sigs) * ps + len(queries) * 8 * ps * len(sigs) * 2


def test_errors_match_reference_conditions(gpu_lib, oracle, golden_dir):
    from cobs_amd import _capi
    s = gpu_lib.Search(os.path.join(golden_dir, "c1.cobs_classic"))
    with pytest.raises(gpu_lib.CobsGpuError) as e:      # classic_search.cpp:431-433
        s.search("ACGT" * 7)
    assert e.value.status == _capi.ERR_QUERY_TOO_SHORT
    with pytest.raises(gpu_lib.CobsGpuError) as e:      # classic_search.cpp:93-96
        s.search(Q50[:20].decode() + "N" + Q50[21:].decode())
    assert e.value.status == _capi.ERR_INVALID_BASE
    with pytest.raises(gpu_lib.CobsGpuError) as e:      # lower case is not ACGT (util/query.cpp:104-141)
        s.search(Q50.decode().lower())
    assert e.value.status == _capi.ERR_INVALID_BASE
    with pytest.raises(oracle.OracleError):
        oracle.Index.open(os.path.join(golden_dir, "c1.cobs_classic")).counts(Q50[:20] + b"N" + Q50[21:])
    with pytest.raises(gpu_lib.CobsGpuError) as e:
        gpu_lib.Search(os.path.join(golden_dir, "fasta", "sample1.fasta"))
    assert e.value.status == _capi.ERR_FORMAT
    # the handle stays usable after an error
    assert s.search(Q50.decode())[0].score == 20


def test_synthetic_index_matches_generator(gpu_lib, oracle):
    """procedural index: rows in HBM equal the checker's generator; counts equal"""
    sigs = [1201, 1789, 2503]
    ps, D = 112, 2 * 8 * 112 + 500
    s = gpu_lib.Search.synthetic("compact", sigs, D, page_size=ps, seed=77)
    ix = oracle.Index.synthetic(1, 31, 1, 1, ps, sigs, D, 77)
    for page, row in ((0, 0), (0, 1200), (1, 17), (2, 2502), (2, 1000)):
        assert np.array_equal(s.read_row(0, page, row, ps), oracle.synth_row(1, 77, ps, 3, D, page, row, ps))
    assert not s.read_row(0, 1, sigs[1], ps).any()          # the zero row
    got = s.read_row(0, 2, 5, ps)
    assert not got[(D - 2 * 8 * ps + 7) // 8:].any()        # padding documents have no bits
    for q in cases.queries_acgt(3, 1030, 40):
        assert np.array_equal(s.counts(q), ix.counts(q))
    s2 = gpu_lib.Search.synthetic("classic", [3001], 1000, seed=5)
    ix2 = oracle.Index.synthetic(0, 31, 1, 1, 0, [3001], 1000, 5)
    for q in cases.queries_acgt(2, 300, 50):
        assert np.array_equal(s2.counts(q), ix2.counts(q))


def test_sharded_counts_sum_to_whole(gpu_lib, oracle, tmp_path):
    """shards hold disjoint sub-index blocks / column ranges; the sum
    of the zero-padded shard vectors equals the unsharded result"""
    q = oracle.random_sequence(500, 3)
    pc = cases.make_compact(cases.tmp(tmp_path, "sh.cobs_compact"), 700, 16, [800, 900, 1000, 1100, 1200, 1300],
                            2, 31, 1, 0.3, 8)
    pk = cases.make_classic(cases.tmp(tmp_path, "sh.cobs_classic"), 3000, 1999, 1, 31, 1, 0.3, 9)
    for p in (pc, pk):
        want = oracle.Index.open(p).counts(q)
        for n in (2, 3, 4):
            total = np.zeros_like(want)
            for r in range(n):
                s = gpu_lib.Search(p, shard_rank=r, shard_count=n)
                c = s.counts(q)
                i = s.info(0)
                outside = np.ones(len(c), dtype=bool)
                outside[i.slot_begin:i.slot_begin + i.slot_count] = False
                assert not c[outside].any()
                total += c
            assert np.array_equal(total, want)


def test_device_topk_matches_partial_sort(gpu_lib, oracle, tmp_path):
    """K3: the k best documents per query (score desc, document asc) selected on the
    device equal the oracle's partial_sort, with many ties (short queries) and
    thresholds, over two files"""
    q_long = oracle.random_sequence(700, 13)
    D = 9000
    planted = {d: f for d, f in zip(range(5, D, 211), np.linspace(0.1, 1.0, 43))}
    pa = cases.make_compact(cases.tmp(tmp_path, "tk.cobs_compact"), D, 160, [900, 1100, 1300, 1500, 1700, 1900, 2100, 2300],
                            1, 31, 1, 0.3, 21, planted=planted, query=q_long)
    pb = cases.make_classic(cases.tmp(tmp_path, "tk.cobs_classic"), 333, 1201, 1, 31, 1, 0.3, 22,
                            planted={7: 1.0, 300: 0.97}, query=q_long)
    s = gpu_lib.Search([pa, pb])
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    queries = [q_long, q_long[:36], q_long[:50], q_long[100:400], q_long[:31]]
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    for t in (0.0, 0.31, 0.9):
        for k in (1, 7, 100, 5000, 8192, 8500):       # beyond 8192 survivors the host orders them
            b.run_topk(t, k)
            b.sync()
            for i, q in enumerate(queries):
                for lim in sorted({1, min(k, 3), k}):
                    assert b.hits_host(i, lim) == cases.oracle_results(ixs, q, t, lim), (t, k, lim, i)
                # num_results = 0 after a top-k pass still returns everything
                assert b.hits_host(i, 0) == cases.oracle_results(ixs, q, t, 0)


def test_long_queries_u32_scores(gpu_lib, oracle, tmp_path):
    """queries long enough for 12, 16, 20, 24 and 32 bit planes (u32 scores from T >= 65536,
    the reference's uint32_t Score path, classic_search.cpp:485-500)"""
    D = 70
    p = cases.make_classic(cases.tmp(tmp_path, "long.cobs_classic"), D, 1201, 1, 31, 1, 0.3, 3)
    pk = cases.make_compact(cases.tmp(tmp_path, "long.cobs_compact"), 150, 8, [257, 509, 1021], 2, 31, 1, 0.3, 4)
    s, sk = gpu_lib.Search(p), gpu_lib.Search(pk)
    ix, ixk = oracle.Index.open(p), oracle.Index.open(pk)
    for length in (2000 + 30, 40000 + 30, 70000 + 30, 1100000 + 30):
        q = oracle.random_sequence(length, length)
        got = s.counts(q)
        want, width = ix.counts(q, want_width=True)
        assert width == (2 if length < 65565 else 4)
        assert np.array_equal(got, want)
        assert np.array_equal(sk.counts(q), ixk.counts(q))
        assert s.search_hits([q], 0.29, 5)[0] == cases.oracle_results([ix], q, 0.29, 5)
        assert s.search_hits([q], 0.0, 0)[0] == cases.oracle_results([ix], q, 0.0, 0)
    q = oracle.random_sequence(17000000, 99)          # T > 2^24: 32 planes
    assert np.array_equal(s.counts(q), ix.counts(q))
    # K3 over 32-bit scores (three radix levels), survivors ordered on the device: num_results > 0
    # never ships score rows (reference: the uint32_t instantiation of counts_to_result, :134-145)
    for length in (70000 + 30, 1100000 + 30):
        qs = [oracle.random_sequence(length, length + j) for j in range(3)]
        for srch, idx in ((s, ix), (sk, ixk)):
            b = gpu_lib.Batch(srch)
            b.set_queries(qs)
            for t, k in ((0.0, 1), (0.0, 9), (0.3, 40), (0.0, 1000)):
                b.run_topk(t, k)
                b.sync()
                assert b.counts_device()[1] == 4
                for i, qq in enumerate(qs):
                    assert b.hits_host(i, k) == cases.oracle_results([idx], qq, t, k), (length, t, k, i)
            assert srch.search_hits(qs, 0.0, 6) == [cases.oracle_results([idx], qq, 0.0, 6) for qq in qs]


def test_empty_and_minimal_batches(gpu_lib, oracle, golden_dir):
    s = gpu_lib.Search(os.path.join(golden_dir, "c1.cobs_compact"))
    assert s.search_hits([], 0.0, 0) == []
    b = gpu_lib.Batch(s)
    b.set_queries([])
    b.run(0.5)
    b.sync()
    ix = oracle.Index.open(os.path.join(golden_dir, "c1.cobs_compact"))
    q = Q50[:31]                                         # exactly one term
    assert np.array_equal(s.counts(q), ix.counts(q))
    assert s.search_hits([q], 1.0, 0)[0] == cases.oracle_results([ix], q, 1.0, 0)


def test_large_host_batch_is_cut_into_passes(gpu_lib, oracle, tmp_path, monkeypatch):
    """cobs_gpu_search_batch splits a batch whose score rows / tables exceed the pass limit;
    results and the index of a bad query are those of one big batch"""
    from cobs_amd import _capi
    q_long = oracle.random_sequence(300, 44)
    p = cases.make_compact(cases.tmp(tmp_path, "pass.cobs_compact"), 900, 16, [401, 503, 601, 701, 809, 907, 1009, 1103],
                           1, 31, 1, 0.3, 5, planted={3: 1.0, 500: 0.8}, query=q_long)
    s = gpu_lib.Search(p)
    ix = oracle.Index.open(p)
    queries = [q_long[i:i + 60 + 7 * (i % 5)] for i in range(40)]
    s.set_tuning("pass_bytes", 5 * s.local_counts)           # 5 queries (one-byte scores) per pass
    for t, lim in ((0.0, 0), (0.3, 4), (0.0, 3)):
        got = s.search_hits(queries, t, lim)
        assert got == [cases.oracle_results([ix], q, t, lim) for q in queries]
    bad = list(queries)
    bad[23] = bad[23][:10] + b"N" + bad[23][11:]
    arr = (_capi.C.c_char_p * len(bad))(*bad)
    lens = (_capi.C.c_size_t * len(bad))(*[len(q) for q in bad])
    offs = (_capi.C.c_size_t * (len(bad) + 1))()
    hits = (_capi.Hit * (len(bad) * s.total_counts))()
    badq = _capi.C.c_size_t(999)
    st = s._lib.cobs_gpu_search_batch(s._h, arr, lens, len(bad), 0.0, 0, hits, len(hits), offs, _capi.C.byref(badq))
    assert st == _capi.ERR_INVALID_BASE and badq.value == 23
    # a query that is too short is reported with ITS index (not the first query of its pass)
    short = list(queries)
    short[31] = short[31][:30]
    arr = (_capi.C.c_char_p * len(short))(*short)
    lens = (_capi.C.c_size_t * len(short))(*[len(q) for q in short])
    badq = _capi.C.c_size_t(999)
    st = s._lib.cobs_gpu_search_batch(s._h, arr, lens, len(short), 0.0, 0, hits, len(hits), offs, _capi.C.byref(badq))
    assert st == _capi.ERR_QUERY_TOO_SHORT and badq.value == 31
    assert b"(query 31)" in s._lib.cobs_gpu_last_error()


def test_default_python_call_ranks_every_document(gpu_lib, oracle, tmp_path):
    """cobs_index.Search.search(query) with its defaults (threshold 0, no limit): every
    document, named, in the reference's order -- two files, several hundred documents, ties"""
    q = oracle.random_sequence(331, 5)
    pa = cases.make_classic(cases.tmp(tmp_path, "a.cobs_classic"), 611, 701, 1, 31, 1, 0.3, 3,
                            planted={17: 1.0, 300: 0.5, 610: 0.5}, query=q)
    pb = cases.make_compact(cases.tmp(tmp_path, "b.cobs_compact"), 333, 8, [509, 401, 307, 600, 450, 333], 1, 31, 1,
                            0.3, 4, planted={0: 0.5, 332: 1.0}, query=q)
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    s = gpu_lib.Search([pa, pb])
    for query in (q, q[:40], q[:31]):
        want = oracle.search(ixs, query, 0.0, 0)
        got = s.search(query)
        assert len(got) == 611 + 333
        assert [(r.doc_name, r.score) for r in got] == [(n, sc) for (_, _, n, sc) in want]
        got5 = s.search(query, 0.0, 5)
        assert [(r.doc_name, r.score) for r in got5] == [(n, sc) for (_, _, n, sc) in want[:5]]


def test_hit_pool_overflow_is_ranked_from_the_score_rows(gpu_lib, oracle, tmp_path):
    """a low threshold on many queries selects more hits than the device pool holds (1 Mi):
    the host API repeats the pass with score rows and ranks them; results unchanged"""
    q = oracle.random_sequence(2000, 77)
    D = 3100
    p = cases.make_classic(cases.tmp(tmp_path, "pool.cobs_classic"), D, 1511, 1, 31, 1, 0.3, 9,
                           planted={5: 1.0, 3000: 0.6}, query=q[:200])
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)
    queries = [q[i:i + 100 + (i % 50)] for i in range(420)]
    offs, hits = s.search_arrays(queries, 0.05, 0)
    assert int(offs[-1]) > (1 << 20)                 # more than the pool
    for i in (0, 1, 200, 419):
        want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, queries[i], 0.05, 0)]
        got = hits[int(offs[i]):int(offs[i + 1])].tolist()
        assert got == want, i
    # and the common case right after it on the same handle: few hits, no score rows needed
    offs2, hits2 = s.search_arrays(queries[:8], 0.9, 0)
    for i in range(8):
        want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, queries[i], 0.9, 0)]
        assert hits2[int(offs2[i]):int(offs2[i + 1])].tolist() == want


def test_packed_queries(gpu_lib, oracle, golden_dir):
    """search_packed: queries back to back in one buffer (bytes or uint8 array) + offsets"""
    import os
    p = os.path.join(golden_dir, "c1.cobs_compact")
    s = gpu_lib.Search(p)
    ix = oracle.Index.open(p)
    qs = [Q50, Q50[3:40], Q50[10:41], Q50[:31]]
    text = b"".join(qs)
    offs = np.cumsum([0] + [len(q) for q in qs])
    for buf in (text, np.frombuffer(text, dtype=np.uint8), bytearray(text)):
        o, h = s.search_packed(buf, offs, 0.0, 0)
        for i, q in enumerate(qs):
            want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, q, 0.0, 0)]
            assert h[int(o[i]):int(o[i + 1])].tolist() == want
    o, h = s.search_packed(b"", [0], 0.5, 0)
    assert len(h) == 0 and list(o) == [0]
    with pytest.raises(ValueError):
        s.search_packed(text, [0, len(text) + 1])
    with pytest.raises(ValueError):
        s.search_packed(text, [0, 40, 35])
    with pytest.raises(gpu_lib.CobsGpuError):        # a 30-character slice is shorter than k = 31
        s.search_packed(text, [0, 30])


def test_survey_probe_known_answer(gpu_lib, oracle, construct, tmp_path):
    """the scores the REAL reference returned during the survey (SURVEY 8c [probed]) through the
    HIP path: classic H = 3 and compact with one-byte pages"""
    pc, pk, q, names, want = cases.survey_probe_files(oracle, construct, tmp_path)
    for p in (pc, pk):
        s = gpu_lib.Search(p)
        assert s.counts(q)[:20].tolist() == want
        assert [(r.doc_name, r.score) for r in s.search(q)] == sorted(zip(names, want), key=lambda t: (-t[1], t[0]))
        assert [(r.doc_name, r.score) for r in s.search(q, 0.25)] == [(n, sc) for n, sc in zip(names, want) if sc >= 68]


def test_pipelined_passes_of_a_large_call(gpu_lib, oracle, tmp_path):
    """a call of >= 64 Ki queries is cut into >= 4 passes pipelined over three scratch batches:
    results are in caller order, and a bad query deep inside is reported by its caller index"""
    q_long = oracle.random_sequence(4000, 91)
    p = cases.make_classic(cases.tmp(tmp_path, "pipe.cobs_classic"), 203, 1201, 1, 31, 1, 0.3, 13,
                           planted={7: 1.0, 150: 0.7}, query=q_long[:120])
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)
    rng = np.random.default_rng(17)
    starts = rng.integers(0, 3900, size=70001)
    lens = rng.integers(31, 90, size=70001)
    queries = [q_long[int(a):int(a) + int(n)] for a, n in zip(starts, lens)]
    queries[0] = q_long[:120]
    for t, lim in ((0.5, 0), (0.0, 3)):
        offs, hits = s.search_arrays(queries, t, lim)
        assert len(offs) == 70002
        for i in [0, 1, 17499, 17500, 17501, 35000, 52501, 69999, 70000] + [int(x) for x in rng.integers(0, 70001, size=40)]:
            want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, queries[i], t, lim)]
            assert hits[int(offs[i]):int(offs[i + 1])].tolist() == want, (t, lim, i)
    bad = list(queries)
    bad[61234] = bad[61234][:5] + b"N" + bad[61234][6:]
    with pytest.raises(gpu_lib.CobsGpuError) as e:
        s.search_arrays(bad, 0.5, 0)
    assert "(query 61234)" in str(e.value)                # named by its index in the call, not in its pass
    # the handle is usable afterwards
    offs, hits = s.search_arrays(queries[:3], 0.5, 0)
    assert hits[int(offs[0]):int(offs[1])].tolist() == [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, queries[0], 0.5, 0)]


def test_two_handles_from_two_threads(gpu_lib, oracle, tmp_path):
    """INTEGRATION.md, threading: one handle = one in-flight call, different handles may be driven
    from different threads at the same time (ctypes releases the GIL inside the library)"""
    import threading
    q_long = oracle.random_sequence(3000, 23)
    pa = cases.make_classic(cases.tmp(tmp_path, "ta.cobs_classic"), 500, 1999, 1, 31, 1, 0.3, 21,
                            planted={3: 1.0}, query=q_long[:200])
    pb = cases.make_compact(cases.tmp(tmp_path, "tb.cobs_compact"), 700, 16, [801, 907, 1009, 1103, 1201, 1301], 2, 31, 1,
                            0.3, 22, planted={650: 0.9}, query=q_long[:200])
    rng = np.random.default_rng(3)
    queries = [q_long[int(a):int(a) + int(n)] for a, n in zip(rng.integers(0, 2800, 400), rng.integers(31, 200, 400))]
    want = {}
    for p in (pa, pb):
        ix = oracle.Index.open(p)
        want[p] = [[(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, q, 0.3, 5)] for q in queries]
    errors = []

    def worker(path):
        try:
            s = gpu_lib.Search(path)
            for _ in range(6):
                if s.search_hits(queries, 0.3, 5) != want[path]:
                    errors.append("mismatch " + path)
                    return
        except Exception as e:                      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(p,)) for p in (pa, pb, pa)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_no_device_memory_leak(gpu_lib, oracle, tmp_path):
    """open / search (every mode) / close in a loop: device memory returns to where it was"""
    import gc

    import torch
    q_long = oracle.random_sequence(600, 3)
    p = cases.make_compact(cases.tmp(tmp_path, "leak.cobs_compact"), 3000, 64, [5001, 7001, 9001, 11003, 13001, 6007], 1,
                           31, 1, 0.3, 8, planted={1: 1.0}, query=q_long[:200])
    queries = [q_long[i:i + 100 + i] for i in range(60)]

    def cycle(budget):
        s = gpu_lib.Search(p, hbm_budget=budget)
        s.search_hits(queries, 0.5, 0)
        s.search_hits(queries, 0.0, 3)
        s.search_hits(queries[:2], 0.0, 0)
        b = gpu_lib.Batch(s)
        b.set_queries(queries)
        b.run_topk(0.2, 4)
        b.sync()
        b.hits_host(0, 4)
        b.close()
        s.close()

    for budget in (0, 1 << 20):
        cycle(budget)
        gc.collect()
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for _ in range(40):
            cycle(budget)
        gc.collect()
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free0 - free1 < (32 << 20), (budget, free0, free1)


def test_indexes_with_different_term_sizes(gpu_lib, oracle, tmp_path):
    """several index files whose term sizes differ (reference classic_search.cpp:413-449): length
    check against the largest k, per-file thresholds ceil(t * (len - k_i + 1)), max_counts over
    all files; K1 runs once per file with that file's k (31 -> the unrolled kernel, 21 -> generic)"""
    from cobs_amd import _capi
    q = oracle.random_sequence(400, 91)
    pa = cases.make_classic(cases.tmp(tmp_path, "k31.cobs_classic"), 90, 997, 1, 31, 1, 0.3, 1,
                            planted={3: 1.0, 40: 0.6}, query=q)
    pb = cases.make_compact(cases.tmp(tmp_path, "k21.cobs_compact"), 50, 2, [1201, 1301, 1409, 1511], 2, 21, 1, 0.3, 2,
                            planted={7: 1.0, 11: 0.5}, query=q)
    pc = cases.make_classic(cases.tmp(tmp_path, "k25.cobs_classic"), 17, 733, 1, 25, 0, 0.3, 3)
    ixs = [oracle.Index.open(p) for p in (pa, pb, pc)]
    s = gpu_lib.Search([pa, pb, pc])
    assert [s.info(f).term_size for f in range(3)] == [31, 21, 25]
    queries = [q[:31], q[:100], q[:270], q[:275], q[50:321], q[:286], q]
    for t, lim in ((0.0, 0), (0.5, 0), (0.55, 3), (0.0, 5), (1.0, 0)):
        got = s.search_hits(queries, t, lim)
        assert got == [cases.oracle_results(ixs, qq, t, lim) for qq in queries], (t, lim)
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    b.run(0.0)
    b.sync()
    for i, qq in enumerate(queries):
        assert np.array_equal(b.counts_host(i), np.concatenate([ix.counts(qq) for ix in ixs]))
    # shorter than the largest term size: the reference exits with "query too short" (:431-433)
    with pytest.raises(gpu_lib.CobsGpuError) as e:
        s.search_hits([q[:100], q[:30]], 0.0, 0)
    assert e.value.status == _capi.ERR_QUERY_TOO_SHORT and "(query 1)" in str(e.value) and "31" in str(e.value)
    # Deviation, on purpose: for 255 <= len - k_min and len - k_max < 255 the reference picks 8-bit
    # scores from the LARGEST term size (:453) and then dies in the file with the smaller one
    # ("query too long", :323-327) although the query is legal.  The engine sizes the scores by
    # the file with the most terms and answers: per-file counts equal the oracle's.
    for ln in (276, 280, 285):
        with pytest.raises(oracle.OracleError):
            oracle.search(ixs, q[:ln])
        got = s.counts(q[:ln])
        assert np.array_equal(got, np.concatenate([ix.counts(q[:ln]) for ix in ixs]))


def test_topk_single_file_is_ordered_on_the_device(gpu_lib, oracle, tmp_path):
    """one file: K3's output is the result, in order -- every score width, ties at the cut,
    fewer passing documents than k, k larger than the index"""
    q_long = oracle.random_sequence(700, 13)
    D = 9000
    planted = {d: f for d, f in zip(range(5, D, 211), np.linspace(0.1, 1.0, 43))}
    p = cases.make_compact(cases.tmp(tmp_path, "tk1.cobs_compact"), D, 160, [900, 1100, 1300, 1500, 1700, 1900, 2100, 2300],
                           1, 31, 1, 0.3, 21, planted=planted, query=q_long)
    s = gpu_lib.Search(p)
    ix = oracle.Index.open(p)
    for queries in ([q_long[:36], q_long[:50], q_long[:33]],             # 8-bit scores, many ties
                    [q_long, q_long[100:400], q_long[:31]]):            # 16-bit scores
        b = gpu_lib.Batch(s)
        b.set_queries(queries)
        for t in (0.0, 0.31, 0.9, 1.0):
            for k in (1, 2, 7, 64, 100, 4097, 8192, 20000):
                b.run_topk(t, k)
                b.sync()
                for i, q in enumerate(queries):
                    assert b.hits_host(i, k) == cases.oracle_results([ix], q, t, k), (t, k, i)
                    assert b.hits_host(i, 1) == cases.oracle_results([ix], q, t, 1)


def test_small_calls_replay_a_captured_graph(gpu_lib, oracle, tmp_path):
    """single queries / small batches of one shape: from the third call on the pass is a hipGraph
    replay; results stay those of the oracle for DIFFERENT queries of the same lengths, and a
    change of shape, threshold or limit falls back to plain launches"""
    q_long = oracle.random_sequence(600, 77)
    p = cases.make_compact(cases.tmp(tmp_path, "g.cobs_compact"), 3000, 64, [900, 1000, 1100, 1200, 1300, 1400], 1, 31, 1,
                           0.3, 5, planted={5: 1.0, 700: 0.9, 2500: 0.6}, query=q_long)
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)
    before = s.graph_replays
    for t, lim in ((0.0, 5), (0.5, 0), (0.0, 0)):
        for i in range(6):
            qs = [q_long[i * 7:i * 7 + 331]]                       # same length, different text
            assert s.search_hits(qs, t, lim) == [cases.oracle_results([ix], qs[0], t, lim)], (t, lim, i)
    assert s.graph_replays - before >= 9                            # 3 shapes x (6 calls - candidate - capture - ...)
    mid = s.graph_replays
    # a batch of four of mixed lengths, repeated; then other lengths
    for i in range(4):
        qs = [q_long[i:i + 100], q_long[i + 5:i + 5 + 31], q_long[i:i + 400], q_long[i + 9:i + 9 + 64]]
        assert s.search_hits(qs, 0.3, 3) == [cases.oracle_results([ix], q, 0.3, 3) for q in qs]
    assert s.graph_replays > mid
    for n in (90, 91, 92, 93):
        qs = [q_long[:n]]
        assert s.search_hits(qs, 0.0, 2) == [cases.oracle_results([ix], qs[0], 0.0, 2)]
    # bad input still reports, also on a replay
    good = q_long[:200]
    for _ in range(3):
        s.search_hits([good], 0.0, 1)
    with pytest.raises(gpu_lib.CobsGpuError):
        s.search_hits([good[:50] + b"N" + good[51:]], 0.0, 1)
    s.set_tuning("graph", 0)
    r = s.graph_replays
    for _ in range(4):
        assert s.search_hits([good], 0.0, 1) == [cases.oracle_results([ix], good, 0.0, 1)]
    assert s.graph_replays == r


def test_graph_replay_between_other_shapes(gpu_lib, oracle, tmp_path):
    """a captured shape is replayed after calls of OTHER shapes ran in between (other score width,
    other result mode): the replay must not inherit anything from them"""
    q_long = oracle.random_sequence(900, 78)
    p = cases.make_compact(cases.tmp(tmp_path, "g2.cobs_compact"), 2000, 64, [900, 1000, 1100, 1200], 1, 31, 1,
                           0.3, 6, planted={5: 1.0, 700: 0.9, 1900: 0.6}, query=q_long)
    ix = oracle.Index.open(p)
    s = gpu_lib.Search(p)

    def check(qs, t, lim):
        assert s.search_hits(qs, t, lim) == [cases.oracle_results([ix], q, t, lim) for q in qs], (len(qs[0]), t, lim)

    a = lambda i: [q_long[i:i + 331]]            # 301 terms: 16-bit scores
    b = lambda i: [q_long[i:i + 100]]            # 70 terms: 8-bit scores
    c = lambda i: [q_long[i:i + 600], q_long[i + 3:i + 3 + 45]]
    for i in range(3):
        check(a(i), 0.0, 5)                      # captured on the third call at the latest
    r0 = s.graph_replays
    for i in range(3, 12):
        check(b(i), 0.4, 0)                      # hits only, other width
        check(a(i), 0.0, 5)                      # shape A again
        check(c(i), 0.0, 0)                      # every document ranked, two queries
        check(a(i + 20), 0.0, 5)
    # three shapes in rotation: each is captured at its second sighting and replayed from then on
    assert s.graph_replays - r0 >= 25


def test_graph_replay_of_a_shape_displaced_by_many_others(gpu_lib, oracle, tmp_path):
    """eight shapes of one handle captured one after another (the batch keeps the current graph and three older
    ones), then the kept ones again, oldest first: a graph that comes back after several other graphs were
    instantiated must still clear the pass's flag words.  With the flags cleared by a captured MEMSET node such a
    replay left 16 bytes of stale host data in them and the pass reported an invalid base in query 998395903 of a
    one-query call (two classic files, threshold 0.2; found by test_gpu_fuzz.py::test_random_ties under
    COBS_FUZZ_SEED=14) -- the flags are cleared by a kernel now (fetch_kernels.hip)."""
    q = oracle.random_sequence(200, 4009)[40:75]
    p1 = cases.make_classic(cases.tmp(tmp_path, "d0.cobs_classic"), 176, 615, 1, 31, 1, 0.3, 450)
    p2 = cases.make_classic(cases.tmp(tmp_path, "d1.cobs_classic"), 4255, 247, 2, 31, 1, 0.7, 451)
    ixs = [oracle.Index.open(p1), oracle.Index.open(p2)]
    s = gpu_lib.Search([p1, p2])
    shapes = [(0.0, 4431), (0.2, 4431), (0.0, 3), (0.2, 3), (0.2, 0), (0.0, 8), (0.2, 8), (0.0, 0)]
    r0 = s.graph_replays
    for t, lim in shapes:
        for _ in range(2):                      # second sighting: captured
            assert s.search_hits([q], t, lim) == [cases.oracle_results(ixs, q, t, lim)], (t, lim)
    for rnd in range(3):
        for t, lim in shapes[4:] + shapes[4:][::-1]:
            assert s.search_hits([q], t, lim) == [cases.oracle_results(ixs, q, t, lim)], (rnd, t, lim)
    assert s.graph_replays - r0 >= 12


def test_graph_replay_of_two_thresholded_shapes(gpu_lib, oracle, tmp_path):
    """two captured shapes that BOTH use a threshold, of different query lengths, in rotation (A,B,A,B,...), in
    hits-only and in top-k mode: the thresholds ceil(t*T) differ per shape and reach the device through one pinned
    staging buffer per batch, so a replay must stage its own (a replay that inherits the other shape's
    threshold -- 85 for a query of 70 terms -- silently returns nothing)"""
    q_long = oracle.random_sequence(600, 91)
    planted = {3: 1.0, 11: 0.8, 500: 0.62, 1700: 0.55, 1999: 0.45}
    p = cases.make_compact(cases.tmp(tmp_path, "g3.cobs_compact"), 2000, 64, [900, 1000, 1100, 1200], 1, 31, 1,
                           0.3, 8, planted=planted, query=q_long)
    ix = oracle.Index.open(p)
    for lim in (0, 4):                              # hits only / K3 with a threshold
        s = gpu_lib.Search(p)
        r0 = s.graph_replays
        for rnd in range(6):
            for L in (100, 200, 331):               # 70 / 170 terms: 8-bit scores, 301 terms: 16-bit
                qs = [q_long[:L]]                   # the SAME text: planted documents pass at 0.5 in every shape
                want = [cases.oracle_results([ix], q, 0.5, lim) for q in qs]
                assert len(want[0]) >= 3
                assert s.search_hits(qs, 0.5, lim) == want, (rnd, L, lim)
        assert s.graph_replays - r0 >= 6            # every shape was replayed (buffers that grow re-key a shape once)


def test_one_graph_serves_a_class_of_query_lengths(gpu_lib, oracle, tmp_path):
    """graphs are keyed by shape class (query count, score planes, launch geometry), not by exact lengths: single
    queries of twenty different lengths (131..150 bp: one class) are served by ONE captured graph -- lengths,
    block counts and thresholds ceil(t * T) reach the kernels through device tables, not through the graph"""
    q_long = oracle.random_sequence(400, 123)
    p = cases.make_compact(cases.tmp(tmp_path, "g4.cobs_compact"), 2000, 64, [900, 1000, 1100, 1200], 1, 31, 1,
                           0.3, 8, planted={3: 1.0, 11: 0.8, 500: 0.62, 1999: 0.45}, query=q_long)
    ix = oracle.Index.open(p)
    for t, lim in ((0.5, 0), (0.0, 7), (0.4, 3)):
        s = gpu_lib.Search(p)
        r0 = s.graph_replays
        for rnd in range(2):
            for L in range(131, 151):
                q = q_long[rnd:rnd + L]
                assert s.search_hits([q], t, lim) == [cases.oracle_results([ix], q, t, lim)], (t, lim, rnd, L)
        assert s.graph_replays - r0 >= 36, (t, lim, s.graph_replays - r0)      # 40 calls: one plain, one capture, the rest replays
    # two queries per call, lengths drawn independently
    s = gpu_lib.Search(p)
    r0 = s.graph_replays
    for i in range(24):
        qs = [q_long[i:i + 131 + (7 * i) % 20], q_long[2 * i:2 * i + 150 - (5 * i) % 20]]
        assert s.search_hits(qs, 0.3, 5) == [cases.oracle_results([ix], q, 0.3, 5) for q in qs], i
    assert s.graph_replays - r0 >= 18


def test_every_score_width_on_one_query(gpu_lib, oracle, tmp_path):
    """the reference's tests run ONE query under every Score width ("check all expansion tables",
    tests/compact_index_query.cpp:54-140, classic_search_disable_8bit / _16bit / _32bit): the same short queries with
    8-, 16- and 32-bit scores -- counts, thresholded hits, tile-level top-k and the full ranking are identical"""
    import torch
    q = oracle.random_sequence(230, 61)
    p = cases.make_compact(cases.tmp(tmp_path, "w.cobs_compact"), 3 * 8 * 24 - 5, 24, [509, 401, 307], 3, 31, 1, 0.1, 4,
                           planted={0: 0.5, 200: 1.0, 570: 0.8}, query=q)
    ix = oracle.Index.open(p)
    queries = [q, q[:80], q[10:45], q[:31], q[3:200]]
    s = gpu_lib.Search(p)
    for width, dtype in ((1, torch.uint8), (2, torch.int16), (4, torch.int32)):
        s.set_tuning("min_score_bytes", width)
        b = gpu_lib.Batch(s)
        b.set_queries(queries)
        b.run(0.0)
        b.sync()
        assert b.counts_tensor().element_size() == width
        for i, qq in enumerate(queries):
            assert np.array_equal(b.counts_host(i), ix.counts(qq)), (width, i)
        for t, lim in ((0.0, 0), (0.5, 0), (0.0, 4), (0.3, 9)):
            assert s.search_hits(queries, t, lim) == [cases.oracle_results([ix], qq, t, lim) for qq in queries], (width, t, lim)


def test_default_call_in_batches_is_ranked_by_host_threads(gpu_lib, oracle, tmp_path):
    """threshold 0, no limit (the reference's default arguments) for MANY queries per call: the
    passes' score rows are ranked by several host threads -- every document, in the reference's
    order incl. ties and the unsorted single-hash case, over two files, whole and sharded"""
    q = oracle.random_sequence(400, 5)
    pa = cases.make_classic(cases.tmp(tmp_path, "ra.cobs_classic"), 611, 701, 1, 31, 1, 0.3, 3, planted={7: 1.0, 600: 0.5}, query=q)
    pb = cases.make_compact(cases.tmp(tmp_path, "rb.cobs_compact"), 3 * 8 * 16 - 3, 16, [401, 503, 601], 1, 31, 1, 0.3, 4)
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    queries = [q[i:i + 31 + (13 * i) % 300] for i in range(37)] + [q[:31]] * 3
    s = gpu_lib.Search([pa, pb])
    got = s.search_hits(queries, 0.0, 0)
    assert got == [cases.oracle_results(ixs, qq, 0.0, 0) for qq in queries]
    assert all(len(g) == 611 + 381 for g in got)
    s.set_tuning("pass_bytes", 9 * s.local_counts * 2)                   # several passes, several windows
    assert s.search_hits(queries, 0.0, 0) == got
    for n, r in ((3, 1), (2, 0)):
        sh = gpu_lib.Search([pa, pb], shard_rank=r, shard_count=n)
        lim = [(sh.info(f).slot_begin, sh.info(f).slot_begin + sh.info(f).slot_count) for f in range(2)]
        part = sh.search_hits(queries, 0.0, 0)
        for g, full in zip(part, got):
            assert g == [h for h in full if lim[h[0]][0] <= h[1] < lim[h[0]][1]]


def test_hash_stream_pipelines_batches_without_changing_results(gpu_lib, oracle, tmp_path):
    """tuning key hash_stream: K1 of a device-resident batch runs on the batch's own stream, K2 on the caller's, tied by
    one event (the overlapped multi-GPU flow of bench.py).  Two files (K1 per file, the event re-recorded), three
    batches run back to back several times on one stream without a sync in between (the next run's K1 must wait for
    the previous K2 of the SAME batch only), thresholds and top-k included: every result equals the oracle's and the
    unpipelined run's."""
    import torch
    q = oracle.random_sequence(600, 5)
    pa = cases.make_compact(cases.tmp(tmp_path, "h.cobs_compact"), 3 * 8 * 40 - 3, 40, [811, 1201, 977], 1, 31, 1, 0.3, 21,
                            planted={0: 1.0, 500: 0.8}, query=q)
    pb = cases.make_classic(cases.tmp(tmp_path, "h.cobs_classic"), 777, 1511, 2, 31, 1, 0.3, 22, planted={7: 0.9}, query=q)
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    s = gpu_lib.Search([pa, pb])
    sets = [[q, q[:31], q[10:300]], [q[100:400], q[5:]], [q[:64], q[3:90], q[7:500], q[:33]]]
    want = [[np.concatenate([ix.counts(x) for ix in ixs]) for x in qs] for qs in sets]
    s.set_tuning("hash_stream", 1)
    batches = [gpu_lib.Batch(s) for _ in sets]
    for b, qs in zip(batches, sets):
        b.set_queries(qs)
    st = torch.cuda.Stream()
    for rnd in range(4):
        for b in batches:                       # no sync between the runs of different batches, nor between rounds
            b.run(0.0, st.cuda_stream)
    for b, qs, w in zip(batches, sets, want):
        b.sync(st.cuda_stream)
        for i in range(len(qs)):
            assert np.array_equal(b.counts_host(i), w[i])
        ms = b.kernel_ms()
        assert ms["scan_ms"] > 0 and ms["hash_ms"] > 0
    for b, qs in zip(batches, sets):
        b.run(0.35, st.cuda_stream)
        b.sync(st.cuda_stream)
        for i, x in enumerate(qs):
            assert b.hits_host(i) == cases.oracle_results(ixs, x, 0.35, 0)
        b.run_topk(0.0, 5, st.cuda_stream, keep_counts=False)
        b.sync(st.cuda_stream)
        for i, x in enumerate(qs):
            assert b.hits_host(i, 5) == cases.oracle_results(ixs, x, 0.0, 5)
    # the host-buffer calls (scratch batches on their own streams, graphs) are unaffected by the key
    assert s.search_hits(sets[0], 0.3, 4) == [cases.oracle_results(ixs, x, 0.3, 4) for x in sets[0]]
    s.set_tuning("hash_stream", 0)
    b = gpu_lib.Batch(s)
    b.set_queries(sets[2])
    b.run(0.0)
    b.sync()
    for i in range(len(sets[2])):
        assert np.array_equal(b.counts_host(i), want[2][i])


# ---- round 5 ----------------------------------------------------------------------------------------------------------
def test_results_in_the_library_arena_and_in_reused_buffers(gpu_lib, oracle, tmp_path):
    """cobs_gpu_search_batch_view: the results of a call stay in an arena the handle owns (same lists as the array form,
    for every kind of call; a later call reuses the memory).  And the Python mirror's own result buffers: a buffer nothing
    references any more is reused by the next call, one that is still referenced is not"""
    q_long = oracle.random_sequence(400, 3)
    p = cases.make_compact(cases.tmp(tmp_path, "v.cobs_compact"), 1200, 32, [1500, 2100, 900, 1700, 1300], 1, 31, 1, 0.3, 8,
                           planted={5: 1.0, 700: 0.9, 1100: 0.85}, query=q_long)
    s = gpu_lib.Search(p, device=0)
    queries = [q_long[:200], q_long[50:300], q_long[10:41], q_long[100:400]] * 9
    addr = None
    for thr, k in ((0.0, 0), (0.8, 0), (0.0, 7), (0.3, 3)):
        offs_a, hits_a = s.search_arrays(queries, thr, k)
        offs_v, hits_v = s.search_view(queries, thr, k)
        assert np.array_equal(offs_a, offs_v) and np.array_equal(hits_a, hits_v), (thr, k)
        if thr == 0.0 and k == 0:
            addr = hits_v.ctypes.data
            assert len(hits_v) == len(queries) * 1200
    offs_v, hits_v = s.search_view(queries[:3], 0.0, 0)           # a smaller call: the same arena
    assert hits_v.ctypes.data == addr
    want = [cases.oracle_results([oracle.Index.open(p)], q, 0.0, 0) for q in queries[:3]]
    assert [hits_v[int(offs_v[i]):int(offs_v[i + 1])].tolist() for i in range(3)] == want
    # the mirror's buffers (results of >= 1 MiB): dropped -> reused; kept -> left alone
    big = queries * 12                                            # 432 queries x 1200 documents x 12 bytes = 6 MB
    o1, h1 = s.search_arrays(big, 0.0, 0)
    a1 = h1.ctypes.data
    first = h1[:5].tolist()
    del o1, h1
    o2, h2 = s.search_arrays(big, 0.0, 0)
    assert h2.ctypes.data == a1 and h2[:5].tolist() == first      # the first call's buffer, used again
    seg = h2[100:200]                                             # a slice keeps the buffer alive ...
    del h2
    o3, h3 = s.search_arrays(big, 0.0, 0)
    assert h3.ctypes.data != a1                                   # ... so the next call does not write into it
    assert seg.tolist() == h3[100:200].tolist()
    res = s.search(q_long[:200])                                  # the lazy ResultList holds on to its buffer the same way
    keep = [(r.doc_name, r.score) for r in res[:3]]
    for _ in range(4):
        s.search_arrays(big, 0.0, 0)
    assert [(r.doc_name, r.score) for r in res[:3]] == keep


def test_score_histogram_is_the_distribution_of_the_rows(gpu_lib, oracle, tmp_path):
    """cobs_gpu_batch_score_histogram (`cobs benchmark-fpr --dist`, src/cobs.cpp:627-632) against the checker's rows: real
    documents only, 8- and 16-bit scores, two files, a shard"""
    import collections
    q_long = oracle.random_sequence(700, 9)
    pa = cases.make_compact(cases.tmp(tmp_path, "h.cobs_compact"), 3 * 8 * 24 - 7, 24, [800, 1200, 1000], 1, 31, 1, 0.3, 2,
                            planted={3: 1.0}, query=q_long)
    pb = cases.make_classic(cases.tmp(tmp_path, "h.cobs_classic"), 333, 901, 2, 31, 1, 0.4, 3)
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    for queries in ([q_long[:100], q_long[40:200], q_long[:31]], [q_long, q_long[100:500]]):
        nb = max(len(q) for q in queries) - 30 + 1
        want = collections.Counter()
        for q in queries:
            for ix in ixs:
                want.update(int(v) for v in ix.counts(q)[:ix.num_docs])
        s = gpu_lib.Search([pa, pb], device=0)
        b = gpu_lib.Batch(s)
        b.set_queries(queries)
        b.run(0.0)
        b.sync()
        h = b.score_histogram(nb)
        assert {i: int(c) for i, c in enumerate(h) if c} == dict(want)
        # the shards' distributions add up to the whole
        tot = np.zeros(nb, dtype=np.uint64)
        for r in range(3):
            sr = gpu_lib.Search([pa, pb], device=0, shard_rank=r, shard_count=3)
            br = gpu_lib.Batch(sr)
            br.set_queries(queries)
            br.run(0.0)
            br.sync()
            tot += br.score_histogram(nb)
        assert np.array_equal(tot, h)


def test_a_thresholded_call_with_many_hits_is_not_run_twice(gpu_lib, oracle, tmp_path):
    """the number of hits of a thresholded call is not known before it has run.  cobs_gpu_search_batch reports
    ERR_CAPACITY with the size, and the caller repeats the WHOLE search; the view call grows the arena it owns while the
    passes come home (cobs_gpu_host_passes counts device passes), and the Python mirror's array form copies out of it"""
    q_long = oracle.random_sequence(4000, 92)
    p = cases.make_classic(cases.tmp(tmp_path, "many.cobs_classic"), 203, 1201, 1, 31, 1, 0.3, 14,
                           planted={9: 1.0, 120: 0.7}, query=q_long[:120])
    ix = oracle.Index.open(p)
    rng = np.random.default_rng(23)
    starts = rng.integers(0, 3900, size=70001)
    lens = rng.integers(31, 90, size=70001)
    queries = [q_long[int(a):int(a) + int(n)] for a, n in zip(starts, lens)]
    s = gpu_lib.Search(p)
    n0 = s.host_passes
    s.search_arrays(queries, 0.0, 3)                      # a call whose result size is known beforehand
    one_run = s.host_passes - n0
    assert one_run >= 4
    n0 = s.host_passes
    offs_v, hits_v = s.search_view(queries, 0.3, 0)       # ~ 100 of 203 documents per query: far beyond the first guess
    assert s.host_passes - n0 == one_run
    assert len(hits_v) > 40 * len(queries)
    offs_v, hits_v = np.array(offs_v), np.array(hits_v)
    for i in [0, 1, 17500, 35000, 70000] + [int(x) for x in rng.integers(0, 70001, size=30)]:
        want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, queries[i], 0.3, 0)]
        assert hits_v[int(offs_v[i]):int(offs_v[i + 1])].tolist() == want, i
    # the array form of the Python mirror goes through the arena too (one copy of the finished lists)
    s2 = gpu_lib.Search(p)
    n0 = s2.host_passes
    offs_a, hits_a = s2.search_arrays(queries, 0.3, 0)
    assert s2.host_passes - n0 == one_run
    assert np.array_equal(offs_a, offs_v) and np.array_equal(hits_a, hits_v)
    del offs_a, hits_a
    # the C call into a caller's buffer: ERR_CAPACITY + the size, nothing written beyond the buffer; the caller repeats it
    import ctypes as C
    from cobs_amd import _capi
    qs = queries[:20000]
    arr = (C.c_char_p * len(qs))(*qs)
    lens = (C.c_size_t * len(qs))(*[len(q) for q in qs])
    offs_c = np.zeros(len(qs) + 1, dtype=np.uint64)
    small = np.zeros(1000 + 8, dtype=s2.HIT_DTYPE)
    small["score"][1000:] = 0xABCDEF
    bad = C.c_size_t(0)
    st = s2._lib.cobs_gpu_search_batch(s2._h, arr, lens, len(qs), 0.3, 0, C.cast(small.ctypes.data, C.POINTER(_capi.Hit)), 1000,
                                       C.cast(offs_c.ctypes.data, C.POINTER(C.c_size_t)), C.byref(bad))
    assert st == _capi.ERR_CAPACITY and int(offs_c[len(qs)]) == int(offs_v[len(qs)])
    assert (small["score"][1000:] == 0xABCDEF).all()
    full = np.zeros(int(offs_c[len(qs)]), dtype=s2.HIT_DTYPE)
    st = s2._lib.cobs_gpu_search_batch(s2._h, arr, lens, len(qs), 0.3, 0, C.cast(full.ctypes.data, C.POINTER(_capi.Hit)), len(full),
                                       C.cast(offs_c.ctypes.data, C.POINTER(C.c_size_t)), C.byref(bad))
    assert st == 0 and np.array_equal(offs_c, offs_v[:len(qs) + 1]) and np.array_equal(full, hits_v[:len(full)])
    # a smaller arena call afterwards, and one with fewer hits: same memory, right lists
    offs_w, hits_w = s.search_view(queries[:5], 0.9, 0)
    for i in range(5):
        want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, queries[i], 0.9, 0)]
        assert hits_w[int(offs_w[i]):int(offs_w[i + 1])].tolist() == want, i


def test_a_limit_too_large_for_k3_with_a_threshold_cuts_every_list(gpu_lib, oracle, comm_one_rank):
    """num_results beyond what K3 selects on the device (k > 65536) AND a threshold: the pass runs as a thresholded one
    (hits into the pool, ordered on the device) and every query's list is cut at the limit -- counts_to_result's
    `num_results = min(num_results, #passing)`, classic_search.cpp:133-147.  [Until round 6 the one-sweep hand-over of an
    ordered pool returned every hit of such a call: found while the sharded call was given the same sweep.]  A procedural
    index of 70 000 documents (70 sub-indexes), almost every document above a low threshold; the one-GPU call, the sharded
    call on a one-rank communicator and the checker agree, with and without the limit."""
    ps, P = 125, 70
    D = P * 8 * ps - 11
    sigs = [1009 + 2 * p for p in range(P)]
    s = gpu_lib.Search.synthetic("compact", sigs, D, page_size=ps, seed=31)
    ix = oracle.Index.synthetic(1, 31, 1, 1, ps, sigs, D, 31)
    qs = [oracle.random_sequence(130, 500 + i) for i in range(3)]
    for t, lim in ((0.05, 66000), (0.05, 0), (0.28, 66000), (0.0, 66000)):
        want = [cases.oracle_results([ix], q, t, lim) for q in qs]
        if (t, lim) == (0.05, 66000):
            assert all(len(w) == 66000 for w in want)
        assert s.search_hits(qs, t, lim) == want, (t, lim)
        assert s.sharded_search_hits(comm_one_rank, qs, t, lim) == want, (t, lim, "sharded")
