"""GPU: randomised geometry sweep (fixed seeds) against the oracle -- random document counts,
page sizes, sub-index counts, signature sizes, hash counts, k, canonicalisation, ragged query
batches, thresholds and limits, under every tile width / wave count of the scan kernel."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


def _random_case(rng, oracle, tmp_path, idx):
    k = int(rng.choice([15, 21, 31, 31, 31, 33]))
    H = int(rng.choice([1, 1, 2, 3]))
    canon = int(rng.integers(0, 2))
    qlen = int(rng.choice([k, k + 7, 100, 300, 1030, 2100]))
    q_long = oracle.random_sequence(max(qlen, 400) + 50, 1000 + idx)
    if rng.random() < 0.5:
        D = int(rng.integers(1, 3000))
        S = int(rng.integers(50, 4000))
        planted = {int(d): float(rng.random()) for d in rng.integers(0, D, size=min(D, 6))}
        path = cases.make_classic(cases.tmp(tmp_path, "f%d.cobs_classic" % idx), D, S, H, k, canon, 0.3, idx,
                                  planted=planted, query=q_long[:qlen])
    else:
        ps = int(rng.choice([1, 2, 4, 8, 16, 24, 40, 64, 128, 136, 256]))   # what the reference's batching accepts (App. A)
        P = int(rng.integers(1, 7))
        D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
        sigs = [int(x) for x in rng.integers(40, 3000, size=P)]
        planted = {int(d): float(rng.random()) for d in rng.integers(0, D, size=min(D, 6))}
        path = cases.make_compact(cases.tmp(tmp_path, "f%d.cobs_compact" % idx), D, ps, sigs, H, k, canon, 0.3, idx,
                                  planted=planted, query=q_long[:qlen])
    nq = int(rng.integers(1, 20))
    queries = []
    for _ in range(nq):
        ln = int(rng.integers(k, len(q_long)))
        st = int(rng.integers(0, len(q_long) - ln + 1))
        queries.append(q_long[st:st + ln])
    queries.append(q_long[:qlen])
    return path, queries


@pytest.mark.parametrize("tile_w,waves,mq", [
    (None, None, "0"), ("4", "1", "0"), ("8", "4", "0"), ("16", "2", "0"), ("32", "1", "0"), ("64", "4", "0"),
    ("8", "1", "0"), ("64", "2", "0"),
    # multi-query work-groups (lane groups = different queries of the batch)
    (None, None, "1"), ("4", "1", "1"), ("8", "4", "1"), ("16", "2", "1"), ("32", "1", "1"), ("8", "1", "1"),
    ("16", "4", "1")])
def test_random_geometries(gpu_lib, oracle, tmp_path, monkeypatch, tile_w, waves, mq):
    monkeypatch.setenv("COBS_GPU_MQ", mq)
    if tile_w:
        monkeypatch.setenv("COBS_GPU_TILE_W", tile_w)
        monkeypatch.setenv("COBS_GPU_WAVES", waves)
    # COBS_FUZZ_SEED shifts the whole sweep (soak runs); the default is the committed, fixed sweep
    rng = np.random.default_rng(20260928 + (int(tile_w) if tile_w else 0) + (int(waves) if waves else 0) + 977 * int(mq)
                                + 100003 * int(os.environ.get("COBS_FUZZ_SEED", "0")))
    for idx in range(40):
        path, queries = _random_case(rng, oracle, tmp_path, idx)
        ix = oracle.Index.open(path)
        s = gpu_lib.Search(path)
        b = gpu_lib.Batch(s)
        b.set_queries(queries)
        t = float(rng.choice([0.0, 0.0, 0.25, 0.5, 0.9]))
        lim = int(rng.choice([0, 0, 1, 3, 50]))
        if lim:
            b.run_topk(t, lim)
        else:
            b.run(t)
        b.sync()
        for i, q in enumerate(queries):
            assert np.array_equal(b.counts_host(i), ix.counts(q)), (path, i, len(q))
            assert b.hits_host(i, lim) == cases.oracle_results([ix], q, t, lim), (path, i, t, lim)
        # host-buffer API with a threshold and no limit: the scan runs hits-only (bit-sliced
        # threshold compare, no score rows)
        tt = t if t > 0 else 0.3
        assert s.search_hits(queries, tt, 0) == [cases.oracle_results([ix], q, tt, 0) for q in queries], (path, tt)
        # a limit without score rows: the scan selects every tile's best, K3 merges the candidates (tile_topk) --
        # where that applies (one hash function, no single-k-mer query in the batch), else rows + K3 as above
        k2 = int(rng.choice([1, 2, 5, 17, 128]))
        b.run_topk(t, k2, keep_counts=False)
        b.sync()
        for i, q in enumerate(queries):
            assert b.hits_host(i, k2) == cases.oracle_results([ix], q, t, k2), (path, i, t, k2)
        assert s.search_hits(queries, t, k2) == [cases.oracle_results([ix], q, t, k2) for q in queries], (path, t, k2)
        # the reference's default call: every document of every query, ordered on the device from four queries on
        if idx % 4 == 0:
            assert s.search_hits(queries, 0.0, 0) == [cases.oracle_results([ix], q, 0.0, 0) for q in queries], path


def test_lds_staged_variant_is_bit_exact(gpu_lib, oracle):
    """the measured LDS-staged variant of the scan (rows HBM -> LDS via global_load_lds -> VGPR,
    hand-counted vmcnt pipeline; tuning key lds_staged, profiles/r02_lds_staged_ab.txt) computes
    the same counts as the VGPR-direct kernel and the oracle: every trip-count parity (1..6 trips
    per wave), 2 and 4 waves, ragged lengths inside the 10-plane range"""
    import bench
    cfg = bench.c3_config(0.01)
    s = gpu_lib.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    ix = oracle.Index.synthetic(1, 31, 1, 1, cfg["page_size"], cfg["signature_sizes"], cfg["num_docs"], 1)
    base = bench.make_queries(48, 993, seed=3)
    qs = [q[:286 + 15 * i] for i, q in enumerate(base)]            # 256 .. 961 terms: 32 .. 121 blocks
    b = gpu_lib.Batch(s)
    b.set_queries(qs)
    want = [ix.counts(q) for q in qs]
    for waves in (2, 4):
        for tile_w in (0, 8, 16, 64):
            s.set_tuning("lds_staged", 1)
            s.set_tuning("waves", waves)
            s.set_tuning("tile_w", tile_w)
            b.run(0.0)
            b.sync()
            for i in range(len(qs)):
                assert np.array_equal(b.counts_host(i), want[i]), (waves, tile_w, i)
            b.run(0.3)                                             # selection epilogue after the staged loop
            b.sync()
            for i in (0, 17, 47):
                assert b.hits_host(i, 0) == cases.oracle_results([ix], qs[i], 0.3, 0)


def test_random_shards(gpu_lib, oracle, tmp_path):
    """random geometries x random shard counts / shard modes / HBM budgets: every shard computes
    exactly its slot range, the shards' hits are its documents' hits, and the slices -- moved
    according to the library's exchange plan -- assemble to the oracle's rows"""
    import ctypes as C
    from cobs_amd import _capi
    lib = _capi.load()
    rng = np.random.default_rng(777 + 100003 * int(os.environ.get("COBS_FUZZ_SEED", "0")))
    for idx in range(24):
        path, queries = _random_case(rng, oracle, tmp_path, idx)
        ix = oracle.Index.open(path)
        N = int(rng.integers(2, 7))
        mode = int(rng.integers(0, 2))
        want = np.stack([ix.counts(q) for q in queries])
        t = float(rng.choice([0.25, 0.5]))
        shards, local = [], []
        budget = 0
        if rng.random() < 0.4:
            budget = int(os.path.getsize(path) * float(rng.uniform(0.6, 1.5)) / N) + 70000
        for r in range(N):
            try:
                s = gpu_lib.Search(path, shard_rank=r, shard_count=N, shard_mode=mode, hbm_budget=budget)
            except gpu_lib.CobsGpuError as e:
                assert e.status == _capi.ERR_CAPACITY and budget, (path, e)       # budget below one column slice
                s = gpu_lib.Search(path, shard_rank=r, shard_count=N, shard_mode=mode)
            i = s.info(0)
            b = gpu_lib.Batch(s)
            b.set_queries(queries)
            b.run(0.0)
            b.sync()
            rows = np.stack([b.counts_host(q) for q in range(len(queries))])
            outside = np.ones(rows.shape[1], dtype=bool)
            outside[i.slot_begin:i.slot_begin + i.slot_count] = False
            assert not rows[:, outside].any(), (path, N, mode, r)
            assert np.array_equal(rows[:, ~outside], want[:, ~outside]), (path, N, mode, r)
            if i.slot_count:
                got = s.search_hits(queries, t, 0)
                for q, g in zip(queries, got):
                    ref = [h for h in cases.oracle_results([ix], q, t, 0) if i.slot_begin <= h[1] < i.slot_begin + i.slot_count]
                    assert g == ref, (path, N, mode, r)
            shards.append((int(i.slot_begin), int(i.slot_count)))
            local.append(np.ascontiguousarray(rows[:, ~outside].astype(np.uint32)).view(np.uint8).reshape(-1))
        assert sum(c for _, c in shards) == want.shape[1]
        # the exchange plan on these layouts (all-to-all), played with memcpy
        nq, total = len(queries), want.shape[1]
        bs = (C.c_uint64 * N)(*[s_[0] for s_ in shards])
        cs = (C.c_uint64 * N)(*[s_[1] for s_ in shards])
        d0 = (C.c_uint64 * 1)(0)
        plans = []
        for r in range(N):
            xf, cp = (_capi.Xfer * N)(), (_capi.Copy2D * N)()
            ncp, out = C.c_size_t(N), (C.c_uint64 * 6)()
            _capi.check(lib.cobs_gpu_exchange_plan(bs, cs, d0, N, 1, total, nq, 4, 1, r, xf, cp, C.byref(ncp), out))
            plans.append((list(xf), list(cp)[:ncp.value], list(out)))
        for r in range(N):
            xf, cps, out = plans[r]
            staging = np.zeros(out[2], dtype=np.uint8)
            for j in range(N):
                if j != r and xf[j].recv_bytes:
                    peer = plans[j][0][r]
                    assert peer.send_bytes == xf[j].recv_bytes
                    staging[xf[j].recv_offset:xf[j].recv_offset + xf[j].recv_bytes] = \
                        local[j][peer.send_offset:peer.send_offset + peer.send_bytes]
            got = np.zeros(out[3], dtype=np.uint8)
            for c in cps:
                src = local[r] if c.src_is_local else staging
                for h in range(c.height):
                    got[c.dst_offset + h * c.dst_pitch:c.dst_offset + h * c.dst_pitch + c.width] = \
                        src[c.src_offset + h * c.src_pitch:c.src_offset + h * c.src_pitch + c.width]
            ref = np.ascontiguousarray(want[out[0]:out[0] + out[1]].astype(np.uint32)).view(np.uint8).reshape(-1)
            assert np.array_equal(got, ref), (path, N, mode, r)


@pytest.mark.parametrize("streamed", [False, True])
def test_random_ties(gpu_lib, oracle, tmp_path, monkeypatch, streamed):
    """Ties at the cut are where an ordering mistake hides (the reference orders by (score desc, document asc) and
    cuts there, classic_search.cpp:134-145): queries of 1..9 terms against dense filters score 0..9 over thousands of
    documents, so every limit cuts through a run of equal scores.  One or two index files per handle, resident or
    streamed under a random budget (whole chunks, row ranges, column slices, rows fetched), limits from 1 to beyond
    the document count, thresholds incl. the all-documents default call, batch and single-query calls -- against the
    oracle."""
    from cobs_amd import _capi
    monkeypatch.setenv("COBS_GPU_ROW_RANGE_MIN", "48")
    rng = np.random.default_rng(31337 + int(streamed) + 100003 * int(os.environ.get("COBS_FUZZ_SEED", "0")))
    done = 0
    for idx in range(24):
        k = int(rng.choice([15, 31, 31]))
        paths = []
        for f in range(int(rng.choice([1, 1, 2]))):
            H = int(rng.choice([1, 1, 2]))
            dens = float(rng.choice([0.3, 0.5, 0.7]))
            if rng.random() < 0.4:
                D, S = int(rng.integers(100, 5000)), int(rng.integers(200, 2500))
                paths.append(cases.make_classic(cases.tmp(tmp_path, "t%d_%d.cobs_classic" % (idx, f)), D, S, H, k, 1, dens, 50 * idx + f))
                size = S * ((D + 7) // 8)
            else:
                ps = int(rng.choice([8, 16, 48, 64, 136, 256]))
                P = int(rng.integers(2, 7))
                D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
                sigs = [int(x) for x in rng.integers(150, 2500, size=P)]
                paths.append(cases.make_compact(cases.tmp(tmp_path, "t%d_%d.cobs_compact" % (idx, f)), D, ps, sigs, H, k, 1, dens, 50 * idx + f))
                size = sum(sigs) * ps
        q_long = oracle.random_sequence(200, 4000 + idx)
        queries = [q_long[o:o + k - 1 + int(rng.integers(1, 10))] for o in rng.integers(0, 150, size=int(rng.integers(1, 12)))]
        budget = 0
        if streamed:
            budget = int(sum(os.path.getsize(p) for p in paths) * float(rng.choice([0.2, 0.4, 0.7])))
            monkeypatch.setenv("COBS_GPU_STREAM_PACKED", str(idx % 2))
        try:
            s = gpu_lib.Search(paths if len(paths) > 1 else paths[0], hbm_budget=budget)
        except gpu_lib.CobsGpuError as e:
            assert streamed and e.status == _capi.ERR_CAPACITY, (paths, budget, e)
            continue
        if streamed:
            mode = int(rng.integers(0, 3))
            if mode == 1:
                s.set_tuning("row_fetch_alpha", 0)
            elif mode == 2:
                s.set_tuning("row_fetch", 0)
        ixs = [oracle.Index.open(p) for p in paths]
        total = sum(ix.num_docs for ix in ixs)
        for lim in [int(x) for x in rng.choice([1, 2, 3, 5, 8, 13, 64, 100, 1000, total, total + 5], size=4, replace=False)] + [0]:
            for t in (0.0, float(rng.choice([0.2, 0.5, 1.0]))):
                want = [cases.oracle_results(ixs, q, t, lim) for q in queries]
                assert s.search_hits(queries, t, lim) == want, (paths, budget, t, lim)
                qi = int(rng.integers(0, len(queries)))
                assert s.search_hits([queries[qi]], t, lim)[0] == want[qi], (paths, budget, t, lim, qi)
        done += 1
    assert done >= 12


def per_pass_hint(idx):
    """every third index of test_random_pass_cuts is a large batch (queries x documents beyond the 1 Mi-entry hit pool)"""
    return idx % 3 != 2


@pytest.mark.parametrize("streamed", [False, True])
def test_random_pass_cuts(gpu_lib, oracle, tmp_path, monkeypatch, streamed):
    """One call of the host-buffer API is cut into device passes (workspace limit `pass_bytes`, pipelining over three
    scratch batches from `pipe_chars` of query text on) whose results are concatenated: random limits that put 1..40
    queries into a pass, batches of 20..300 ragged queries, thresholds from "everything passes" (the hit pool of a
    pass overflows and the pass is repeated with score rows) to 1.0, limits, the all-documents default call (ordered
    on the device from 4 queries per pass on), and now and then a query with a character outside ACGT somewhere in
    the batch (the call reports the FIRST such query by its index in the call) -- against the oracle."""
    from cobs_amd import _capi
    monkeypatch.setenv("COBS_GPU_ROW_RANGE_MIN", "48")
    rng = np.random.default_rng(8086 + int(streamed) + 100003 * int(os.environ.get("COBS_FUZZ_SEED", "0")))
    done = 0
    for idx in range(10):
        k = int(rng.choice([15, 31]))
        H = int(rng.choice([1, 1, 2]))
        ps = int(rng.choice([16, 64, 136]))
        P = int(rng.integers(1, 6))
        D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
        sigs = [int(x) for x in rng.integers(150, 2000, size=P)]
        q_long = oracle.random_sequence(700, 7000 + idx)
        path = cases.make_compact(cases.tmp(tmp_path, "p%d.cobs_compact" % idx), D, ps, sigs, H, k, 1, 0.4, 90 + idx,
                                  planted={0: 1.0, D - 1: 0.6, D // 2: 0.8}, query=q_long[:300])
        budget = int(os.path.getsize(path) * float(rng.choice([0.3, 0.6]))) if streamed else 0
        try:
            s = gpu_lib.Search(path, hbm_budget=budget)
        except gpu_lib.CobsGpuError as e:
            assert streamed and e.status == _capi.ERR_CAPACITY, (path, budget, e)
            continue
        ix = oracle.Index.open(path)
        nq = int(rng.integers(20, 300)) if per_pass_hint(idx) else int(rng.integers(250, 420))
        lens = rng.choice([k, k + 3, 60, 100, 300, 650], size=nq)
        queries = [q_long[o:o + int(n)] for o, n in zip(rng.integers(0, 40, size=nq), lens)]
        per_pass = int(rng.choice([0, 1, 2, 3, 7, 16, 40]))        # 0: the default limit, the whole batch in one pass (pool overflow)
        s.set_tuning("pass_bytes", per_pass * s.local_counts * int(rng.choice([1, 2])))
        s.set_tuning("pipe_chars", int(rng.choice([0, 1, 4096])))
        for t, lim in ((0.0, 0), (float(rng.choice([0.01, 0.3])), 0), (float(rng.choice([0.0, 0.5, 1.0])), int(rng.choice([1, 4, 50, D]))),
                       (0.8, 0)):
            want = [cases.oracle_results([ix], q, t, lim) for q in queries]
            assert s.search_hits(queries, t, lim) == want, (path, budget, per_pass, t, lim)
        if idx % 2 == 0:
            bad_at = sorted(int(x) for x in rng.choice(nq, size=2, replace=False))
            broken = list(queries)
            for i in bad_at:
                pos = int(rng.integers(0, len(broken[i])))
                broken[i] = broken[i][:pos] + b"N" + broken[i][pos + 1:]
            with pytest.raises(gpu_lib.CobsGpuError) as ei:
                s.search_hits(broken, 0.3, 0)
            assert ei.value.status == _capi.ERR_INVALID_BASE and ("(query %d)" % bad_at[0]) in str(ei.value), (bad_at, str(ei.value))
            assert s.search_hits(queries[:5], 0.3, 0) == [cases.oracle_results([ix], q, 0.3, 0) for q in queries[:5]]   # and goes on
        done += 1
    assert done >= 5
