"""GPU parity at BASELINE.json's full sizes (configs[1] and configs[2]) through
sampled exact equality against the oracle (which recomputes the procedural index
rows it needs) and size-independent properties: term additivity of counts,
batch-position independence, count bounds, and a checksum of checksums between
two differently shaped passes over the same queries."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

_ORACLE_ROWS = {}


def _oracle_rows(ix, key, queries):
    """the checker's count rows of a query set, computed once per session (C4: 245 000 regenerated rows per query)"""
    if key not in _ORACLE_ROWS:
        _ORACLE_ROWS[key] = [ix.counts(q) for q in queries]
    return _ORACLE_ROWS[key]


def _open(gpu, cfg):
    return gpu.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"],
                                page_size=cfg["page_size"], term_size=cfg["term_size"],
                                canonicalize=cfg["canonicalize"], num_hashes=cfg["num_hashes"],
                                seed=cfg["seed"])


def _oracle_index(O, cfg):
    return O.Index.synthetic(1 if cfg["kind"] == "compact" else 0, cfg["term_size"], cfg["canonicalize"],
                             cfg["num_hashes"], cfg["page_size"], cfg["signature_sizes"], cfg["num_docs"],
                             cfg["seed"])


@pytest.mark.parametrize("name", ["c3", "c2", "c4"])
def test_full_size_config(gpu_lib, oracle, name):
    cfg = {"c3": bench.c3_config, "c2": bench.c2_config, "c4": bench.c4_config}[name]()
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    # true positives (round 5): 64 sequences, each a part of 48 documents that hold 100 % ... 50 % of its terms --
    # planted into the GPU's matrix by cobs_gpu_plant and into the checker's definition by oracle_plant
    plants = bench.planted_documents(cfg)
    bench.apply_plants(s, plants)
    bench.apply_plants(ix, plants)
    nq = 256 if name != "c4" else 32              # C4: 1M documents, 245 sub-indexes, 68 GB (the checker regenerates 245 000 rows per query)
    queries = bench.make_queries(nq, 1000)
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    b.run(0.0)
    b.sync()
    T = 1000
    # exact equality of EVERY query's score row with the oracle (which regenerates the ~8000
    # procedural rows a query touches), plus device-side row checksums against oracle checksums
    import torch
    t_all = b.counts_tensor().to(torch.int64).bitwise_and(0xFFFF)
    w = (torch.arange(t_all.shape[1], device=t_all.device, dtype=torch.int64) % 1021) + 1
    dev_sum = t_all.sum(dim=1).cpu().numpy()
    dev_wsum = (t_all * w).sum(dim=1).cpu().numpy()
    wn = (np.arange(t_all.shape[1], dtype=np.int64) % 1021) + 1
    wants = _oracle_rows(ix, (name, nq, "with planted documents"), queries)
    for i in range(nq):
        want = wants[i]
        assert np.array_equal(b.counts_host(i), want), (name, i)
        assert int(dev_sum[i]) == int(want.sum()) and int(dev_wsum[i]) == int((want.astype(np.int64) * wn).sum())
    # a few rows of the index itself, incl. the last row of the largest sub-index
    P = len(cfg["signature_sizes"])
    width = cfg["page_size"] if cfg["kind"] == "compact" else (cfg["num_docs"] + 7) // 8
    for page, row in ((0, 0), (P - 1, cfg["signature_sizes"][-1] - 1), (P // 2, 12345)):
        want = oracle.synth_row(1 if cfg["kind"] == "compact" else 0, cfg["seed"], cfg["page_size"], P,
                                cfg["num_docs"], page, row, width)
        assert np.array_equal(s.read_row(0, page, row, width), want)
    c0 = b.counts_host(0)
    total = s.total_counts
    assert c0.shape == (total,) and c0.max() <= T
    assert not c0[cfg["num_docs"]:].any()                    # padding documents never score
    assert 0.25 * T < c0[:cfg["num_docs"]].mean() < 0.35 * T    # bit density ~0.297
    # property: additivity over terms -- counts(q) = counts(q[:m+k-1]) + counts(q[m:])
    q = queries[3]
    k = cfg["term_size"]
    parts = [q[:400 + k - 1], q[400:]]
    b2 = gpu_lib.Batch(s)
    b2.set_queries(parts + [q, queries[0]])
    b2.run(0.0)
    b2.sync()
    assert np.array_equal(b2.counts_host(0) + b2.counts_host(1), b2.counts_host(2))
    assert np.array_equal(b2.counts_host(2), b.counts_host(3))
    # property: position in the batch / batch shape does not matter
    assert np.array_equal(b2.counts_host(3), c0)
    # checksum of checksums over the whole batch, device-side, vs a second pass in reversed order
    import torch
    t = b.counts_tensor()
    sums1 = t.to(torch.int64).bitwise_and(0xFFFF).sum(dim=1).cpu().numpy()
    b3 = gpu_lib.Batch(s)
    b3.set_queries(queries[::-1])
    b3.run(0.0)
    b3.sync()
    sums3 = b3.counts_tensor().to(torch.int64).bitwise_and(0xFFFF).sum(dim=1).cpu().numpy()
    assert np.array_equal(sums1, sums3[::-1])
    assert int(sums1[0]) == int(c0.sum())
    # on-device selection at full size: a RANDOM query reaches 0.8 nowhere (like the
    # reference's own benchmark), everything passes a threshold of one k-mer
    b.run(0.8)
    b.sync()
    assert b.hits_host(0) == []
    # ... and queries that are mutated windows of the planted sequences DO have hits at the CLI's default threshold
    # (src/cobs.cpp:486-489): the hit lists and their order (score desc, document asc: classic_search.cpp:127-156)
    # against the checker's, through the score-writing pass, the hits-only pass, the limited pass and the host API
    nh = 128 if name != "c4" else 32
    hq = bench.planted_queries(plants, nh)
    want = [[(d, sc) for (_, d, _n, sc) in oracle.search(ix, q, 0.8)] for q in hq]
    assert sum(len(w_) for w_ in want) >= 10 * nh and min(len(w_) for w_ in want) >= 1      # the workload has hits ...
    planted_below = 0
    for i, q in enumerate(hq[:8]):                   # ... and planted documents on BOTH sides of the threshold
        row = ix.counts(q)
        docs = plants[i % len(plants)][1]
        planted_below += int((row[docs] < int(np.ceil(0.8 * T))).sum())
    assert planted_below > 0
    bh = gpu_lib.Batch(s)
    bh.set_queries(hq)
    bh.run(0.8)
    bh.sync()
    for i in range(nh):
        assert [(d, sc) for (_, d, sc) in bh.hits_host(i)] == want[i], (name, i)
    bh.run_hits(0.8)
    bh.sync()
    for i in range(nh):
        assert [(d, sc) for (_, d, sc) in bh.hits_host(i)] == want[i], (name, i, "hits only")
    bh.run_topk(0.8, 5, keep_counts=False)
    bh.sync()
    for i in range(nh):
        assert [(d, sc) for (_, d, sc) in bh.hits_host(i, 5)] == want[i][:5], (name, i, "top 5")
    got = s.search_batch([q.decode() for q in hq[:16]], 0.8)
    for i in range(16):
        assert [(r.doc_name, r.score) for r in got[i]] == [("file_%06u" % d, sc) for (d, sc) in want[i]], (name, i, "search")
    b.run(0.25)
    b.sync()
    hits = b.hits_host(0, 50)
    thr = int(np.ceil(0.25 * T))
    order = sorted([(int(sc), d) for d, sc in enumerate(c0[:cfg["num_docs"]]) if sc >= thr],
                   key=lambda x: (-x[0], x[1]))[:50]
    assert [(sc, d) for (_, d, sc) in hits] == order


@pytest.mark.parametrize("bp", [50, 100, 150, 250])
def test_full_size_short_reads(gpu_lib, oracle, bp):
    """sequencing reads against the full-size C3 index: 8-bit scores (T <= 255), the multi-query
    work-groups (<= 10 blocks) and a ragged batch; sampled exact equality + properties"""
    import torch
    cfg = bench.c3_config()
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    k = cfg["term_size"]
    nq = 4099                                       # not a multiple of the 8 queries of a work-group
    queries = bench.make_queries(nq, bp - k + 1, seed=bp)
    ragged = [q[:k + (i * 7) % (bp - k + 1)] for i, q in enumerate(queries)]     # 1 .. T terms
    for qs in (queries, ragged):
        b = gpu_lib.Batch(s)
        b.set_queries(qs)
        b.run(0.0)
        b.sync()
        t = b.counts_tensor()
        assert t.dtype == torch.uint8 and tuple(t.shape) == (nq, s.local_counts)
        # every read of the batch: device-side row checksums against the oracle's, exact rows for 64
        tt = t.to(torch.int64)
        w = (torch.arange(tt.shape[1], device=tt.device, dtype=torch.int64) % 1021) + 1
        dev_sum, dev_wsum = tt.sum(dim=1).cpu().numpy(), (tt * w).sum(dim=1).cpu().numpy()
        wn = (np.arange(tt.shape[1], dtype=np.int64) % 1021) + 1
        step = max(1, nq // 512)
        for i in list(range(0, nq, step)) + [1, 7, 8, nq - 2, nq - 1]:
            want = ix.counts(qs[i])
            assert int(dev_sum[i]) == int(want.sum()) and int(dev_wsum[i]) == int((want.astype(np.int64) * wn).sum()), (bp, i)
        for i in list(range(0, nq, nq // 57)) + [1, 7, 8, nq // 2, nq - 2, nq - 1]:
            assert np.array_equal(b.counts_host(i), ix.counts(qs[i])), (bp, i)
        terms = torch.tensor([len(q) - k + 1 for q in qs], device=t.device)
        assert bool((t.max(dim=1).values.to(torch.int64) <= terms).all())
        assert not bool(t[:, cfg["num_docs"]:].any())            # padding documents never score
        # batch-position independence: the same reads in reversed order, device-side row sums
        b2 = gpu_lib.Batch(s)
        b2.set_queries(qs[::-1])
        b2.run(0.0)
        b2.sync()
        s1 = t.to(torch.int64).sum(dim=1).cpu().numpy()
        s2 = b2.counts_tensor().to(torch.int64).sum(dim=1).cpu().numpy()
        assert np.array_equal(s1, s2[::-1])
        # exact top-5 on the device (K3 over 8-bit scores) vs the oracle's ranking
        b.run_topk(0.0, 5)
        b.sync()
        for i in (0, 9, nq - 1):
            want = [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, qs[i], 0.0, 5)]
            assert b.hits_host(i, 5) == want, (bp, i)


@pytest.mark.parametrize("kind", ["classic", "compact"])
def test_signature_size_beyond_32_bits(gpu_lib, oracle, kind):
    """sub-indexes with more than 2^32 rows (signature sizes of human-sized documents are ~10^10;
    the reference's signature_size is a uint64): 64-bit row-index table, 69 GB of rows in HBM"""
    S = (1 << 32) + 12345
    if kind == "classic":
        cfg = dict(kind="classic", signature_sizes=[S], num_docs=61, page_size=0, term_size=31,
                   canonicalize=1, num_hashes=1, seed=11)
    else:       # a small and a huge sub-index side by side share one table format
        cfg = dict(kind="compact", signature_sizes=[1000003, S], num_docs=100, page_size=8, term_size=31,
                   canonicalize=1, num_hashes=2, seed=12)
    s = _open(gpu_lib, cfg)
    ix = _oracle_index(oracle, cfg)
    assert s.signature_size(0, len(cfg["signature_sizes"]) - 1) == S
    queries = bench.make_queries(40, 300, seed=5) + bench.make_queries(3, 1, seed=6)
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    b.run(0.0)
    b.sync()
    rows_beyond = 0
    for i, q in enumerate(queries):
        assert np.array_equal(b.counts_host(i), ix.counts(q)), (kind, i)
    # the lookups really land beyond row 2^32 - 1 for a share of the terms
    P = len(cfg["signature_sizes"])
    width = cfg["page_size"] if kind == "compact" else (cfg["num_docs"] + 7) // 8
    for row in (S - 1, (1 << 32) + 1, (1 << 32) - 1):
        want = oracle.synth_row(1 if kind == "compact" else 0, cfg["seed"], cfg["page_size"], P,
                                cfg["num_docs"], P - 1, row, width)
        assert np.array_equal(s.read_row(0, P - 1, row, width), want)
        rows_beyond += int(want.any())
    assert rows_beyond > 0
    got = s.search_hits(queries[:5], 0.3, 7)
    assert got == [[(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, q, 0.3, 7)] for q in queries[:5]]


def test_large_files_cross_staging_boundaries(gpu_lib, oracle, construct, tmp_path):
    """file-backed indexes big enough to cross the 64 MiB re-pitch chunks and the 1 GiB
    straight-copy steps of the upload path"""
    rng = np.random.default_rng(5)

    def random_rows(rows, width):
        raw = rng.integers(0, 2 ** 63, size=(rows * width + 7) // 8, dtype=np.int64)
        a = raw & rng.integers(0, 2 ** 63, size=raw.shape, dtype=np.int64)       # density 0.25
        return a.view(np.uint8)[:rows * width].reshape(rows, width)

    queries = [oracle.random_sequence(1030, 60 + i) for i in range(3)]
    # classic: 10 000 docs -> 1250-byte rows (re-pitched to 1280), 230 MB
    D, S = 10000, 184321
    pc = str(tmp_path / "big.cobs_classic")
    construct.write_classic(pc, 31, 1, ["d%05d" % i for i in range(D)], S, 1, random_rows(S, 1250))
    # compact: 512-byte pages (copied straight), 1.28 GB -> crosses the 1 GiB copy step
    ps, sigs = 512, [1300003, 1200007]
    pk = str(tmp_path / "big.cobs_compact")
    mats = [random_rows(s, ps) for s in sigs]
    construct.write_compact(pk, 31, 1, ps, [(s, 1) for s in sigs], ["d%05d" % i for i in range(2 * 8 * ps)], mats)
    for p in (pc, pk):
        s = gpu_lib.Search(p)
        ix = oracle.Index.open(p)
        for q in queries:
            assert np.array_equal(s.counts(q), ix.counts(q))
        assert s.search_hits(queries, 0.26, 20) == [
            [(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, q, 0.26, 20)] for q in queries]
        del s


def test_streamed_file_at_scale_is_bit_exact(gpu_lib, oracle, tmp_path):
    """BASELINE configs[4] in small: the C3 geometry at 1/4 scale as a 4.6 GB .cobs_compact FILE
    (written by the generator), opened under a 1.5 GB HBM budget (whole sub-indexes and column
    slices of the large ones stream through the two buffers), every query of a 64-query batch equal
    to the oracle, which regenerates the procedural rows itself.  (The same run at the full
    configs[4] size -- 183.7 GB file, 64 GB budget -- is profiles/r02_c5_184GB_bench.json.)"""
    import cobs_amd
    cfg = bench.c3_config(0.25)
    path = str(tmp_path / "c5_quarter.cobs_compact")
    cobs_amd.write_synthetic(path, "compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    s = gpu_lib.Search(path, hbm_budget=1500 * 1000 * 1000)
    assert s.info(0).hbm_bytes <= 1500 * 1000 * 1000
    ix = _oracle_index(oracle, cfg)
    queries = bench.make_queries(64, 1000)
    b = gpu_lib.Batch(s)
    b.set_queries(queries)
    b.run(0.0)
    b.sync()
    assert b.stats()["scan_launches"] >= 4                      # really streamed in several chunks
    for i, q in enumerate(queries):
        assert np.array_equal(b.counts_host(i), ix.counts(q)), i
    got = s.search_hits(queries[:8], 0.0, 5)
    assert got == [[(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, q, 0.0, 5)] for q in queries[:8]]


@pytest.mark.parametrize("shape", ["c3", "c4"])
def test_full_geometries_from_20_gb_files(gpu_lib, oracle, tmp_path, shape):
    """The headline geometry (BASELINE configs[2]: 100 000 documents, 8 sub-indexes of 1568-byte pages, S_p geometric)
    and the geometry of configs[3] (1 M documents, 245 sub-indexes of 512-byte pages), each as a FILE of more than
    20 GB, opened RESIDENT: the path a user's index takes -- parse_index_header, the slab upload through pinned
    staging, re-pitching 1568 -> 1664 bytes (c3) or the straight copy of rows that already have the device pitch (c4)
    -- instead of the in-HBM generator the other full-size tests share with their checker.  The checker here reads
    the same FILE through its own header parser and mmap (oracle.Index.open) and counts on the CPU: every score row
    of the queries element by element, plus rows of the matrix read back from HBM against the file's bytes at the
    offsets the checker computes."""
    import os
    import shutil
    import cobs_amd
    import psutil
    where = str(tmp_path)
    cfg = bench.c3_config(1.1) if shape == "c3" else bench.c4_config(0.3)
    if shape == "c4":
        # configs[3] at the size BASELINE names (68 GB) where a tmpfs and the host's memory take it (the MI355X boxes
        # of this project: 1.5 TB of /dev/shm over 3 TB of RAM, /tmp a 79 GB overlay), else 0.3 of its rows: 20.4 GB
        full = bench.c4_config(1.0)
        need = sum(full["signature_sizes"]) * full["page_size"]
        if (os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * need
                and psutil.virtual_memory().available > 4 * need):
            cfg, where = full, "/dev/shm"
    npages = len(cfg["signature_sizes"])
    width = cfg["page_size"]
    size = sum(cfg["signature_sizes"]) * width
    assert size > 20 * 10 ** 9
    if shutil.disk_usage(where).free <= size + (2 << 30) and os.path.isdir("/dev/shm"):
        where = "/dev/shm"                                       # (pytest's tmp directory is short of space: a tmpfs)
    free = shutil.disk_usage(where).free
    assert free > size + (2 << 30), "needs %.1f GB of scratch space under %s" % (size / 1e9, where)
    path = os.path.join(where, "cobs_test_%d_%s.cobs_compact" % (os.getpid(), shape))
    cobs_amd.write_synthetic(path, "compact", cfg["signature_sizes"], cfg["num_docs"], page_size=width, seed=7)
    try:
        assert os.path.getsize(path) > size
        s = gpu_lib.Search(path)                                 # no budget: resident, uploaded from the file
        info = s.info(0)
        assert info.kind == 1 and info.num_pages == npages and info.page_size == width and info.num_docs == cfg["num_docs"]
        assert info.hbm_bytes > size                             # all of it in HBM (rows at their device pitch)
        ix = oracle.Index.open(path)                             # the checker's own reader of the same file
        assert [ix.signature_size(p) for p in range(npages)] == cfg["signature_sizes"]
        queries = bench.make_queries(64 if shape == "c3" else 24, 1000, seed=77)
        b = gpu_lib.Batch(s)
        b.set_queries(queries)
        b.run(0.0)
        b.sync()
        for i, q in enumerate(queries):
            assert np.array_equal(b.counts_host(i), ix.counts(q)), i
        # ranking from the same rows
        got = s.search_hits(queries[:4], 0.0, 7)
        assert got == [[(f, d, sc) for (f, d, _n, sc) in oracle.search(ix, q, 0.0, 7)] for q in queries[:4]]
        # rows of the matrix as HBM holds them against the FILE's bytes: first / last row of the first and last
        # sub-index, rows on both sides of a 256 MiB upload slab (171 196 rows of 1568 bytes)
        raw = np.memmap(path, dtype=np.uint8, mode="r")
        data0 = os.path.getsize(path) - size                     # the matrix is the file's tail
        off = [0]
        for sp in cfg["signature_sizes"]:
            off.append(off[-1] + sp * width)
        last = npages - 1
        rows = [(0, 0), (0, cfg["signature_sizes"][0] - 1), (last, 0), (last, cfg["signature_sizes"][last] - 1)]
        if shape == "c3":
            rows += [(7, 171195), (7, 171196), (7, 171197), (3, 2 * 171196 - 1), (3, 2 * 171196)]
        else:                                                    # (a sub-index is smaller than a slab here)
            rows += [(1, 0), (122, 12345), (243, cfg["signature_sizes"][243] - 1), (244, 1)]
            slab = (256 << 20) // width                          # rows of one upload slab
            rows += [(244, r) for r in (slab - 1, slab, 2 * slab) if r < cfg["signature_sizes"][244]]
        for page, row in rows:
            o = data0 + off[page] + row * width
            assert np.array_equal(s.read_row(0, page, row, width), raw[o:o + width]), (page, row)
        del raw, b, s
    finally:
        os.unlink(path)


def _shards_in_turn(gpu, opener, nshards, mode, queries, budget=0):
    """every shard of an N-way sharded index opened one after another on this one GPU: -> (slot layout per rank,
    local count rows per rank as the scan leaves them in HBM, score width)"""
    import torch
    begins, counts, local, eb, hbm = [], [], [], None, []
    for r in range(nshards):
        s = opener(shard_rank=r, shard_count=nshards, shard_mode=mode, hbm_budget=budget)
        info = s.info(0)
        begins.append([int(info.slot_begin)])
        counts.append([int(info.slot_count)])
        hbm.append(int(info.hbm_bytes))
        if budget:
            assert info.hbm_bytes <= budget
        b = gpu.Batch(s)
        b.set_queries(queries)
        b.run(0.0)
        b.sync()
        t = b.counts_tensor()
        assert t.shape[1] == info.slot_count == s.local_counts
        eb = t.element_size()
        local.append(t.cpu().numpy().view(np.uint8).reshape(-1).copy())
        b.close()
        s.close()
        del t, b, s
        torch.cuda.empty_cache()
    return begins, counts, local, eb, hbm


def _check_shard_union(lib, begins, counts, local, eb, total, wants, nshards):
    from tests import xchg
    nq = len(wants)
    # the slot ranges of the shards partition counts_size
    pos = 0
    for r in range(nshards):
        if counts[r][0]:
            assert begins[r][0] == pos and begins[r][0] % 8 == 0 and counts[r][0] % 8 == 0
            pos += counts[r][0]
    assert pos == total
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[eb]
    want_rows = np.stack(wants).astype(dt)
    # union of the local rows = the oracle
    for r in range(nshards):
        rows = local[r].view(dt).reshape(nq, counts[r][0])
        assert np.array_equal(rows, want_rows[:, begins[r][0]:begins[r][0] + counts[r][0]]), r
    # ... and the N-rank exchange of exactly these rows, played on the CPU from the library's own plan:
    # all-to-all (every rank ends with the full rows of the queries it owns) and all-gather
    for mode in (1, 0):
        owned = 0
        for q0, qn, got in xchg.emulate(lib, local, begins, counts, [0], total, nq, eb, mode):
            assert np.array_equal(got, np.ascontiguousarray(want_rows[q0:q0 + qn]).view(np.uint8).reshape(-1)), mode
            owned += qn
        assert owned == (nq if mode == 1 else nq * nshards)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_c4_eight_shards_in_turn(gpu_lib, oracle, mode):
    """BASELINE configs[3] at its own geometry -- 1 M documents, 245 sub-indexes, 68 GB -- cut into the 8 shards an
    8-GPU node holds (mode 0: equal scan time ~ equal columns, lines of cache-resident columns discounted; mode 1: whole sub-indexes; mode 2: equal bytes, 8.5 GB each,
    cuts inside sub-indexes), every shard on this GPU in turn, 32 queries: the slot ranges partition counts_size, the union of the shards' rows is the
    oracle's, and the all-to-all / all-gather plans of the 8 ranks assemble those rows into the oracle's
    (everything of the N = 8 run but the xGMI transfers themselves).
    Shard boundary: reference cobs/query/compact_index/mmap_search_file.cpp:22-27, search_file.cpp:30-32."""
    from cobs_amd import _capi
    cfg = bench.c4_config()
    ix = _oracle_index(oracle, cfg)
    nq = 32
    queries = bench.make_queries(nq, 1000)
    wants = _oracle_rows(ix, ("c4", nq), queries)

    def opener(**kw):
        return gpu_lib.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"],
                                        term_size=cfg["term_size"], canonicalize=cfg["canonicalize"],
                                        num_hashes=cfg["num_hashes"], seed=cfg["seed"], **kw)
    begins, counts, local, eb, hbm = _shards_in_turn(gpu_lib, opener, 8, mode, queries)
    assert eb == 2
    if mode == 2:       # byte-balanced: every shard holds an eighth of the 68 GB (documents per shard differ 8x)
        assert max(hbm) < 1.1 * min(hbm) and 8.0e9 < min(hbm) and max(hbm) < 9.5e9
    elif mode == 1:     # whole sub-indexes: every cut on a sub-index boundary
        assert all(b[0] % (8 * cfg["page_size"]) == 0 for b in begins)
    else:               # time-balanced: a gather's cost is its columns (128-byte lines), and a line of a sub-index whose tile
        # column stays in the Infinity Cache costs 0.77-0.91 of one that does not (plan.cpp: time_balanced_cuts) -- the
        # shards of the small sub-indexes hold up to 1.3x the score slots of the shard of the largest, never less
        slots = [c[0] for c in counts]
        assert max(slots) < 1.35 * min(slots) and sum(slots) == ix.counts_size
        assert slots[0] >= slots[-1] and all(a >= b - 1024 for a, b in zip(slots, slots[1:]))
    _check_shard_union(_capi.load(), begins, counts, local, eb, ix.counts_size, wants, 8)


def test_c5_quarter_scale_eight_streamed_shards_in_turn(gpu_lib, oracle, tmp_path):
    """BASELINE configs[4]'s N = 8 arithmetic at quarter scale: the 4.6 GB .cobs_compact FILE cut into 8 byte-balanced
    shards (shard_mode 2: what a streamed shard costs is the bytes that cross its PCIe link), every shard opened under an
    HBM budget smaller than itself (so each rank streams its slices), in turn on
    this GPU; union and exchange plans as above."""
    import cobs_amd
    from cobs_amd import _capi
    cfg = bench.c3_config(0.25)
    path = str(tmp_path / "c5q.cobs_compact")
    cobs_amd.write_synthetic(path, "compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    ix = _oracle_index(oracle, cfg)
    nq = 48
    queries = bench.make_queries(nq, 1000)
    wants = [ix.counts(q) for q in queries]
    budget = 200 * 1000 * 1000          # a shard is ~575 MB
    begins, counts, local, eb, _ = _shards_in_turn(gpu_lib, lambda **kw: gpu_lib.Search(path, **kw), 8, 2, queries, budget)
    _check_shard_union(_capi.load(), begins, counts, local, eb, ix.counts_size, wants, 8)
