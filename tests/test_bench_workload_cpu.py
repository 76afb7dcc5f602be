"""CPU: the benchmark's planted workload (bench.planted_documents / planted_queries) against the checker alone -- the
synthetic index has true positives, the queries that are windows of planted sequences have hits on BOTH sides of the
CLI's default threshold, a random query has none, and the whole thing is a pure function of its seeds (every rank of an
N-rank run, and the checker, build the same documents and queries)."""
import numpy as np

import bench


def test_planted_workload_has_hits_on_both_sides_of_the_threshold(oracle):
    cfg = bench.c3_config(0.004)                       # the C3 geometry with 1000 ... 16 000 rows per sub-index
    plants = bench.planted_documents(cfg, 1000, n_seq=6, docs_per_seq=21)
    again = bench.planted_documents(cfg, 1000, n_seq=6, docs_per_seq=21)
    assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(plants, again))
    assert all(len(set(d.tolist())) == len(d) and d.max() < cfg["num_docs"] for (_t, d, _k) in plants)
    ix = bench.oracle_index(cfg, plants)
    plain = bench.oracle_index(cfg)
    hq = bench.planted_queries(plants, 24, 1000)
    assert hq == bench.planted_queries(plants, 24, 1000) and all(len(q) == 1030 and set(q) <= set(b"ACGT") for q in hq)
    T = 1000
    thr = int(np.ceil(0.8 * T))
    above = below = 0
    for i, q in enumerate(hq):
        row, row0 = ix.counts(q), plain.counts(q)
        docs = plants[i % len(plants)][1]
        others = np.setdiff1d(np.arange(cfg["num_docs"]), np.concatenate([p[1] for p in plants]))
        assert np.array_equal(row[others], row0[others])          # planting touches the planted documents only
        assert (row[docs] >= row0[docs]).all() and (row[docs] > row0[docs]).any()
        above += int((row[docs] >= thr).sum())
        below += int((row[docs] < thr).sum())
        hits = oracle.search(ix, q, 0.8)
        assert {d for (_f, d, _n, _s) in hits} >= {int(d) for d in docs if row[d] >= thr}
        assert [s for (_f, _d, _n, s) in hits] == sorted((s for (_f, _d, _n, s) in hits), reverse=True)
    assert above >= 4 * len(hq) and below >= 2 * len(hq)
    # an unmutated window of a planted sequence scores T in every document that holds all of its terms
    full = [int(d) for d, k in zip(plants[0][1], plants[0][2]) if k == 1000]
    assert full and all(ix.counts(hq[0])[d] == T for d in full)
    # a random query reaches the threshold nowhere (this tiny geometry has only a few hundred rows per sub-index: allow noise)
    rq = bench.make_queries(4, 1000, seed=5)
    assert all(len(oracle.search(plain, q, 0.8)) == 0 for q in rq)
