"""CPU: the shard planner of libcobs_gpu.so (cobs_gpu_plan_shards, host only) and the header
parser's defences.  A compact index is a concatenation of sub-indexes over disjoint document
ranges (reference cobs/query/compact_index/mmap_search_file.cpp:22-27, search_file.cpp:30-32);
shards are contiguous score-slot ranges that are disjoint, ascending and cover counts_size --
whatever the shard count and mode."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tests import cases, xchg


def _plan(path, n, mode=0):
    from cobs_amd import _capi
    lib = _capi.load()
    b = (C.c_uint64 * n)()
    c = (C.c_uint64 * n)()
    by = (C.c_uint64 * n)()
    _capi.check(lib.cobs_gpu_plan_shards(os.fsencode(path), n, mode, b, c, by))
    return list(b), list(c), list(by)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_shards_are_a_partition_of_the_score_slots(oracle, tmp_path, golden_dir, mode):
    pc = cases.make_compact(cases.tmp(tmp_path, "p.cobs_compact"), 700, 16, [800, 900, 1000, 1100, 1200, 1300], 2)
    pk = cases.make_classic(cases.tmp(tmp_path, "p.cobs_classic"), 3000, 1999, 1)
    pw = cases.make_compact(cases.tmp(tmp_path, "w.cobs_compact"), 5000, 200, [300, 5000, 700, 9000], 1)
    # rows of 400 bytes (25 chunks: cuts of the time-balanced mode fall on whole 128-byte lines, 8 chunks) and sub-indexes
    # whose lines are priced differently (by rows x 128 B against half the Infinity Cache: decided by the header alone)
    pr = cases.make_compact(cases.tmp(tmp_path, "r.cobs_compact"), 3 * 8 * 400 - 1, 400, [64, 2000, 128], 1)
    for p in (pc, pk, pw, pr, os.path.join(golden_dir, "c1.cobs_compact"), os.path.join(golden_dir, "c1.cobs_classic")):
        ix = oracle.Index.open(p)
        for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 61):
            begin, count, _ = _plan(p, n, mode)
            pos = 0
            for r in range(n):
                if count[r] == 0:
                    continue
                assert begin[r] == pos and begin[r] % 8 == 0 and count[r] % 8 == 0, (p, n, r)
                pos += count[r]
            assert pos == ix.counts_size, (p, n)


def test_default_mode_balances_work_not_bytes(oracle, tmp_path):
    """sub-indexes whose sizes differ 16x (the shape of BASELINE configs[2]).  The scan is a gather -- a term looks up
    one row in every sub-index -- so a shard's time follows the COLUMNS (score slots) it holds: mode 0 gives every
    shard about the same scan time (whole 128-byte lines, a line of a sub-index whose tile column stays in the Infinity
    Cache counts 0.77-0.91 of one that does not: plan.cpp time_balanced_cuts), whatever that does to its bytes; mode 2 equalises the bytes in HBM (rounds 1-3's default), which
    leaves the shard of the small sub-indexes with several times the slots of the others"""
    ratio = 16.0 ** (1.0 / 7.0)
    sigs = [int(300 * ratio ** p) for p in range(8)]
    p = cases.make_compact(cases.tmp(tmp_path, "b.cobs_compact"), 8 * 8 * 160 - 5, 160, sigs, 1)
    total = sum(s * 160 for s in sigs)
    slots = 8 * 8 * 160
    for n in (2, 3, 4, 5, 8):
        _, c0, _ = _plan(p, n, 0)
        _, c1, by1 = _plan(p, n, 1)
        _, c2, by2 = _plan(p, n, 2)
        assert max(c0) <= 1.1 * slots / n + 8 * 128 and min(c0) >= 0.9 * slots / n - 8 * 128      # equal work
        assert max(by2) <= 1.25 * total / n + 4096 * n            # equal bytes (pitch padding and zero rows aside)
        if n == 8:
            assert max(by1) >= 2.5 * total / n and len(set(c1)) == 1
            assert max(c2) >= 2.5 * slots / n                     # ... and with them several times the work
    # a cut inside a sub-index of rows of 256 bytes and more falls on whole 128-byte lines (8 chunks of 16 bytes = 1024 slots)
    pw = cases.make_compact(cases.tmp(tmp_path, "w.cobs_compact"), 4 * 8 * 384 - 9, 384, [400, 500, 600, 700], 1)
    for n in (3, 5, 7):
        bw, cw, _ = _plan(pw, n, 0)
        assert all(b % 1024 == 0 for b in bw) and sum(cw) == 4 * 8 * 384
    # many small sub-indexes: cuts snap to sub-index boundaries (whole sub-indexes only)
    p2 = cases.make_compact(cases.tmp(tmp_path, "m.cobs_compact"), 100 * 8 * 16 - 3, 16, [50 + 3 * i for i in range(100)], 1)
    begin, count, _ = _plan(p2, 4, 0)
    assert all(b % (8 * 16) == 0 for b in begin)


def _line_weight(sig):
    """plan.cpp time_balanced_cuts: what a 128-byte line of a sub-index of `sig` rows costs"""
    col = sig * 128
    return 1.0 if col > (128 << 20) else min(0.905, 0.765 + 0.0025 * col / 1e6)


def _best_maximum(costs, n):
    """smallest maximum of a contiguous partition of `costs` into n groups (binary search + greedy, as the planner)"""
    lo, hi = max(costs), sum(costs)
    for _ in range(60):
        mid, groups, acc = (lo + hi) / 2, 1, 0.0
        for c in costs:
            if acc + c > mid * (1 + 1e-12):
                groups, acc = groups + 1, 0.0
            acc += c
        lo, hi = (lo, mid) if groups <= n else (mid, hi)
    return hi


def test_default_mode_reaches_the_smallest_maximum_over_whole_lines(tmp_path):
    """BASELINE configs[2]'s geometry as a SPARSE file (header + 18.4 GB of holes: the planner reads the header only):
    mode 0 cuts on whole 128-byte lines, and the dearest shard -- lines priced by where their tile column lives, a row's
    partial last line a whole one -- costs what the best contiguous partition of the 104 lines costs.  The 8-way split is
    15 / 14 / 14 / 13 / 12 / 12 / 12 / 12 lines (scripts/shard_times.py on it: mean / max scan time 0.959)."""
    import bench
    cfg = bench.c3_config()
    sigs, ps = cfg["signature_sizes"], cfg["page_size"]
    hdr = _compact_header(cfg["num_docs"], ps, [(s, 1) for s in sigs])
    path = str(tmp_path / "c3_sparse.cobs_compact")
    with open(path, "wb") as f:
        f.write(hdr)
        f.truncate(len(hdr) + sum(sigs) * ps)
    lines = (ps + 127) // 128                                        # 12 full lines + 32 bytes
    costs = [_line_weight(s) for s in sigs for _ in range(lines)]
    slots_of_line = [min(1024, 8 * ps - 1024 * i) for _ in sigs for i in range(lines)]
    for n in (2, 3, 4, 5, 8, 16):
        begin, count, _ = _plan(path, n, 0)
        assert sum(count) == 8 * ps * len(sigs)
        got, pos, li = [], 0, 0
        for r in range(n):                                         # every shard is a run of whole lines
            assert begin[r] == pos
            k, acc = 0, 0
            while acc < count[r]:
                acc += slots_of_line[li + k]
                k += 1
            assert acc == count[r], (n, r)
            got.append(sum(costs[li:li + k]))
            li, pos = li + k, pos + count[r]
        assert max(got) <= _best_maximum(costs, n) * (1 + 1e-9), (n, got)
        if n == 8:
            assert count == [14592, 13568, 13568, 12544, 11520, 11520, 11520, 11520]


def _classic_header(ndocs, sig, nh=1, names=None):
    h = b"COBS:CLASSIC_INDEX" + struct.pack("<IIBIQQ", 1, 31, 1, ndocs, sig, nh)
    for i in range(ndocs if names is None else names):
        h += b"d%d\n" % i
    return h + b"CLASSIC_INDEX"


def _compact_header(ndocs, page_size, params, names=None):
    h = b"COBS:COMPACT_INDEX" + struct.pack("<IIBIIQ", 1, 31, 1, len(params), ndocs, page_size)
    for s, nh in params:
        h += struct.pack("<QQ", s, nh)
    for i in range(ndocs if names is None else names):
        h += b"d%d\n" % i
    pad = (page_size - ((len(h) + 13) % page_size)) % page_size if page_size < (1 << 20) else 0
    return h + b"\0" * pad + b"COMPACT_INDEX"


def test_crafted_headers_are_rejected_not_trusted(tmp_path):
    """signature_size / page_size / document counts that wrap 64-bit arithmetic or promise more
    than the file holds must be a format error at open, on the host, before any allocation
    (reference headers are trusted PODs: cobs/file/compact_index_header.cpp:45-65)"""
    import cobs_amd
    from cobs_amd import _capi
    bad = {
        "wrap_mul": _compact_header(8, 1 << 20, [(1 << 44, 1)]),                  # page_size * signature_size = 2^64
        "wrap_sum": _compact_header(8, 16, [((1 << 60) + 1, 1)] * 16),
        "huge_sig": _classic_header(8, (1 << 63) + 5),
        "ndocs_lie": _classic_header(0xFFFFFFF0, 100, names=3),
        "nparams_lie": b"COBS:COMPACT_INDEX" + struct.pack("<IIBIIQ", 1, 31, 1, 0xFFFFFFF0, 8, 16),
        "page_size_huge": _compact_header(8, 1 << 40, [(100, 1)]),
    }
    for name, raw in bad.items():
        p = tmp_path / (name + ".cobs")
        p.write_bytes(raw + b"\0" * 64)
        with pytest.raises(cobs_amd.CobsGpuError) as e:
            cobs_amd.Search(str(p))
        assert e.value.status == _capi.ERR_FORMAT, (name, e.value)
        b = (C.c_uint64 * 2)()
        c = (C.c_uint64 * 2)()
        assert _capi.load().cobs_gpu_plan_shards(os.fsencode(str(p)), 2, 0, b, c, None) == _capi.ERR_FORMAT, name


@pytest.mark.parametrize("mode", [0, 1])                   # all-gather, all-to-all
def test_exchange_plan_moves_every_count_exactly_once(oracle, tmp_path, mode):
    """The multi-GPU exchange of libcobs_gpu.so is a host-computed plan executed over RCCL
    (comm.cpp: plan_exchange).  Here N ranks are emulated on the CPU: every rank's local count rows
    are the oracle's counts cut to the engine's own shard layout, the plans' sends / receives are
    played with memcpy, the assembly copies are applied -- every rank must end with the oracle's
    full rows of the queries it owns.  Also: send sizes match the peer's receive sizes (a mismatch
    would hang or corrupt a real ncclSend / ncclRecv pair) and all ranks pick the same collective."""
    from cobs_amd import _capi
    lib = _capi.load()
    q_long = oracle.random_sequence(400, 21)
    ratio = 16.0 ** (1.0 / 7.0)
    pa = cases.make_compact(cases.tmp(tmp_path, "x.cobs_compact"), 8 * 8 * 48 - 9, 48,
                            [int(150 * ratio ** p) for p in range(8)], 1, 31, 1, 0.3, 8)
    pb = cases.make_classic(cases.tmp(tmp_path, "x.cobs_classic"), 1000, 701, 1, 31, 1, 0.3, 9)
    ixs = [oracle.Index.open(pa), oracle.Index.open(pb)]
    doc_off = [0, ixs[0].counts_size]
    total = ixs[0].counts_size + ixs[1].counts_size
    for nq in (1, 3, 8, 13):
        queries = [q_long[i:i + 40 + 9 * i] for i in range(nq)]
        full = np.stack([np.concatenate([ix.counts(q) for ix in ixs]) for q in queries])
        for eb, dt in ((1, np.uint8), (2, np.uint16), (4, np.uint32)):
            rows = full.astype(dt)
            for N in (1, 2, 3, 5, 8):
                for shard_mode in (0, 1):
                    lay = [_plan(p, N, shard_mode) for p in (pa, pb)]
                    begins = [[lay[f][0][r] for f in range(2)] for r in range(N)]
                    counts = [[lay[f][1][r] for f in range(2)] for r in range(N)]
                    # what rank r's scan leaves in HBM: [nq][local slots] (its files' slices back to back)
                    local = []
                    for r in range(N):
                        parts = [rows[:, doc_off[f] + begins[r][f]: doc_off[f] + begins[r][f] + counts[r][f]] for f in range(2)]
                        local.append(np.ascontiguousarray(np.concatenate(parts, axis=1)).view(np.uint8).reshape(-1))
                    owned = []
                    for q0, qn, got in xchg.emulate(lib, local, begins, counts, doc_off, total, nq, eb, mode):
                        owned.append((q0, qn))
                        want = np.ascontiguousarray(rows[q0:q0 + qn]).view(np.uint8).reshape(-1)
                        assert np.array_equal(got, want), (N, mode, eb, nq)
                    if mode == 1:                                                  # the owners partition the batch
                        assert [o[0] for o in owned] == [nq * j // N for j in range(N)]
                        assert sum(o[1] for o in owned) == nq
                    else:
                        assert all(o == (0, nq) for o in owned)


def test_hit_exchange_plan_routes_every_record_to_its_query_owner(oracle, tmp_path):
    """The owner-routed hit exchange (comm.cpp: cobs_gpu_batch_exchange_hits_owned) as a plan, N ranks emulated on the
    CPU: every rank's hit pool = the oracle's hits of ITS documents (the engine's own shard layout), bucketed by the
    rank that owns each record's query (rank j owns [nq*j/N, nq*(j+1)/N)); the plans' sends / receives are played with
    slices.  Every rank must end with exactly the oracle's hits of its own queries, every send size must equal the
    peer's receive size, and every record must cross exactly once."""
    from cobs_amd import _capi
    lib = _capi.load()
    q_long = oracle.random_sequence(400, 33)
    ratio = 16.0 ** (1.0 / 7.0)
    p = cases.make_compact(cases.tmp(tmp_path, "h.cobs_compact"), 8 * 8 * 48 - 9, 48, [int(150 * ratio ** i) for i in range(8)],
                           1, 31, 1, 0.3, 8, planted={3: 1.0, 700: 0.9, 1500: 0.7, 3000: 0.8}, query=q_long)
    ix = oracle.Index.open(p)
    for nq in (1, 3, 8, 13):
        queries = [q_long[i:i + 60 + 9 * i] for i in range(nq)]
        hits = [(q, f, d, sc) for q, qq in enumerate(queries) for (f, d, _n, sc) in oracle.search(ix, qq, 0.33, 0)]
        assert len(hits) > nq
        for N in (1, 2, 3, 5, 8):
            for shard_mode in (0, 1):
                begin, count, _ = _plan(p, N, shard_mode)
                owner = lambda q: max(j for j in range(N) if nq * j // N <= q)
                local = [[h for h in hits if begin[r] <= h[2] < begin[r] + count[r]] for r in range(N)]
                assert sum(len(x) for x in local) == len(hits)
                bucketed = [sorted(x, key=lambda h: owner(h[0])) for x in local]
                counts = [[sum(1 for h in local[r] if owner(h[0]) == j) for j in range(N)] for r in range(N)]
                flat = (C.c_uint64 * (N * N))(*[c for row in counts for c in row])
                plans = []
                for r in range(N):
                    xf = (_capi.Xfer * N)()
                    out = (C.c_uint64 * 2)()
                    _capi.check(lib.cobs_gpu_hit_exchange_plan(flat, N, r, xf, out))
                    plans.append((list(xf), list(out)))
                moved = 0
                for i in range(N):
                    xf, out = plans[i]
                    assert out[1] == 16 * len(local[i])
                    got = [None] * (out[0] // 16)
                    for j in range(N):
                        peer = plans[j][0][i]                    # what j sends to i
                        assert peer.send_bytes == xf[j].recv_bytes and peer.send_bytes % 16 == 0, (N, i, j)
                        src = bucketed[j][peer.send_offset // 16:(peer.send_offset + peer.send_bytes) // 16]
                        got[xf[j].recv_offset // 16:(xf[j].recv_offset + xf[j].recv_bytes) // 16] = src
                        if j != i:
                            moved += len(src)
                    assert None not in got
                    q0, q1 = nq * i // N, nq * (i + 1) // N
                    assert sorted(got) == sorted(h for h in hits if q0 <= h[0] < q1), (N, i, nq)
                assert moved == sum(1 for r in range(N) for h in local[r] if owner(h[0]) != r)


# ---- round 5: the out-of-core plan (residency per slice) as host arithmetic ----------------------------------------------
def _plan_stream(path, budget, rank=0, count=1, mode=0):
    import ctypes as C
    from cobs_amd import _capi
    out = (C.c_uint64 * 4)()
    keep = (C.c_uint8 * 512)()
    n = C.c_size_t(0)
    _capi.check(_capi.load().cobs_gpu_plan_stream(path.encode(), budget, rank, count, mode, C.byref(out), keep, 512, C.byref(n)))
    return [int(v) for v in out], [int(keep[i]) for i in range(n.value)]


def test_budget_is_spent_per_slice(tmp_path, monkeypatch):
    """plan.cpp: chunk_part / plan_index without a device.  The stream buffers are bounded (COBS_GPU_STREAM_BUF_KIB), what the
    budget leaves beyond them keeps whole slices of the streamed file resident -- the smallest first, then whatever else
    fits; a file cut by columns (two hash functions) keeps the wide buffers; everything stays inside the budget"""
    from tests import cases
    monkeypatch.setenv("COBS_GPU_ROW_RANGE_MIN", "48")
    ps = 64
    sigs = [300, 450, 700, 1100, 1700, 2600, 4000, 6200]           # the C3 shape in small: ratio ~1.5
    D = 8 * 8 * ps - 9
    p1 = cases.make_compact(cases.tmp(tmp_path, "pl.cobs_compact"), D, ps, sigs, 1, 31, 1, 0.3, 4)
    file_bytes = sum(sigs) * ps
    pitch = 128                                                     # 64-byte rows at the device pitch ... or packed: either way
    # everything fits: resident, no buffers
    out, keep = _plan_stream(p1, 0)
    assert out[0] == 0 and out[3] == 0 and keep == [1] * 8
    # a third of the file as budget, buffers of 1/12 of it each: the rest keeps the small sub-indexes
    budget = file_bytes // 3
    monkeypatch.setenv("COBS_GPU_STREAM_BUF_KIB", str(max(1, budget // 12 // 1024)))
    out, keep = _plan_stream(p1, budget)
    buf, kept, per_pass, nchunks = out
    assert 0 < buf <= budget // 12 + 1024 and 2 * buf + kept <= budget
    assert kept > 0 and per_pass < file_bytes and nchunks >= 3
    first_streamed = keep.index(0)
    assert first_streamed >= 2 and keep[:first_streamed] == [1] * first_streamed        # the smallest sub-indexes stay
    assert per_pass == sum(s * ps for s, k in zip(sigs, keep) if not k)                   # what is not resident crosses the link
    # round 4's plan on request: two buffers of half the budget, nothing of the file resident
    monkeypatch.setenv("COBS_GPU_STREAM_BUF_KIB", "0")
    out0, keep0 = _plan_stream(p1, budget)
    assert keep0 == [0] * 8 and out0[1] == 0 and out0[2] == file_bytes and out0[0] * 2 <= budget
    # a larger budget keeps more, never less
    monkeypatch.setenv("COBS_GPU_STREAM_BUF_KIB", str(max(1, budget // 12 // 1024)))
    out2, keep2 = _plan_stream(p1, 2 * budget)
    assert out2[1] >= kept and sum(keep2) >= sum(keep) and out2[2] <= per_pass
    # a shard plans its own slices under its own budget
    outs, keeps = _plan_stream(p1, budget // 4, rank=1, count=2, mode=2)
    assert 2 * outs[0] + outs[1] <= budget // 4 and len(keeps) >= 1
    # two hash functions: cut by columns -> the wide buffers, nothing else resident
    p2 = cases.make_compact(cases.tmp(tmp_path, "pl2.cobs_compact"), D, ps, sigs, 2, 31, 1, 0.3, 5)
    outc, keepc = _plan_stream(p2, budget)
    assert keepc == [0] * 8 and outc[1] == 0 and 2 * outc[0] <= budget and outc[0] > budget // 12 + 1024
