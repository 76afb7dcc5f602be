#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X COBS query engine.

Metric (BASELINE.json): k-mer queries/s + achieved HBM GB/s on the synthetic
100k-document compact index (8 sub-indexes, page_size 1568 B, S_p 250k..4M rows,
18.4 GB in HBM), 1000-k-mer (1030-character) queries, batch of 10k queries --
BASELINE configs[2], the configuration the metric is quoted on.

A "step" is one pass of the hot path over the resident batch:
  K1 canonicalise + XXH64 + row index per sub-index, K2 row gather + AND +
  bit-sliced per-document count (+ on-device threshold selection if
  --threshold > 0), counts written to HBM.  Query text and index are resident in
  HBM before the timed region; nothing is copied to the host inside it.

Usage:  python bench.py --gpus N --steps K --warmup W
N > 1 = one rank per GPU over RCCL.  Launched by torch.distributed.run (WORLD_SIZE / RANK in the
environment) the script is one rank of N; started plainly (`python bench.py --gpus 8`) it launches
the N ranks itself (`launch_plan`): the same torch.distributed.run command, rendezvous on 127.0.0.1.
Either way the printed line carries n_gpus == --gpus == rccl_ranks, or the run fails.
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cobs_amd  # noqa: E402
from cobs_amd.launch import (Watchdog, _StorePeers, bind_to_numa_node, gpu_numa_node, launch_plan,  # noqa: E402,F401
                             run_preflight, visible_devices)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PCIE_PEAK_GBS = 63.0       # PCIe Gen5 x16 (MI355X_MICROARCH.md: 63 GB/s spec), the bound of the out-of-core configuration


def c3_config(scale=1.0):
    """BASELINE configs[2] / SURVEY 8d: compact, D=100000, page_size 1568 -> P=8,
    S_p geometric 250k..4M rows (ratio 16^(1/7))."""
    ratio = 16.0 ** (1.0 / 7.0)
    sigs = [max(64, int(round(250000 * scale * ratio ** p))) for p in range(8)]
    return {"kind": "compact", "num_docs": 100000, "page_size": 1568, "signature_sizes": sigs,
            "term_size": 31, "canonicalize": 1, "num_hashes": 1, "seed": 1}


def c2_config(scale=1.0):
    """BASELINE configs[1]: classic, 10k docs x 1M-row signatures"""
    return {"kind": "classic", "num_docs": 10000, "page_size": 0,
            "signature_sizes": [max(64, int(1000000 * scale))],
            "term_size": 31, "canonicalize": 1, "num_hashes": 1, "seed": 1}


def c4_config(scale=1.0):
    """BASELINE configs[3]: compact, 1M docs, default page size 512 B -> 245 sub-indexes (the index
    that is sharded by sub-index block over 8 GPUs; 68 GB, it also fits one MI355X)"""
    r = (1600000 / 100000) ** (1.0 / 244)
    return {"kind": "compact", "num_docs": 1000000, "page_size": 512,
            "signature_sizes": [max(64, int(100000 * r ** i * scale)) for i in range(245)],
            "term_size": 31, "canonicalize": 1, "num_hashes": 1, "seed": 1}


def make_queries(n, kmers, seed=42):
    """benchmark-fpr queries (reference src/cobs.cpp:709-720): one std::mt19937(seed),
    rng() % 4 -> ACGT, kmers+30 characters each.  numpy's legacy RandomState uses the
    same init_genrand seeding and emits the same raw 32-bit stream."""
    rs = np.random.RandomState(seed)
    raw = rs.randint(0, 2 ** 32, size=n * (kmers + 30), dtype=np.uint64)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[(raw % 4).astype(np.int64)]
    text = text.reshape(n, kmers + 30)
    return [text[i].tobytes() for i in range(n)]


PLANT_SALT = 0x5EED0FC0B5
PLANT_KEEP = (1000, 950, 900, 850, 800, 700, 500)


def planted_documents(cfg, kmers=1000, n_seq=64, docs_per_seq=48, seed=7):
    """TRUE POSITIVES of the synthetic index ("for parity plant true positives"): random
    bits alone score every document ~ Binomial(T, 0.3), so no query ever reaches the CLI's default threshold 0.8
    (src/cobs.cpp:486-489) and the thresholded paths would be measured and tested empty-handed.  `n_seq` random
    sequences of kmers + 230 bases; each is a part of `docs_per_seq` documents spread over the whole index, which hold
    100 % ... 50 % of its terms (PLANT_KEEP in rotation; the rule: cobs_gpu_plant, include/cobs_gpu_batch.h).
    -> list of (text, docs uint32[], keep_permille uint32[]); the same list for the GPU index and for the checker's"""
    rs = np.random.RandomState(seed)
    out = []
    nd = int(cfg["num_docs"])
    for _ in range(n_seq):
        text = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=kmers + 230)].tobytes()
        docs = rs.choice(nd, size=min(docs_per_seq, nd), replace=False).astype(np.uint32)
        keep = np.array([PLANT_KEEP[i % len(PLANT_KEEP)] for i in range(len(docs))], dtype=np.uint32)
        out.append((text, docs, keep))
    return out


def planted_queries(plants, n, kmers=1000, seed=11):
    """queries WITH hits: query q is a window of kmers + 30 bases of planted sequence q % n_seq with 0 / 0.2 / 0.5 / 1 %
    of its bases changed (by q // n_seq % 4), so that (1 - m)^31 = 100 / 94 / 86 / 73 % of its terms are intact: together
    with the documents' own shares the scores of the planted documents lie on both sides of 0.8 T"""
    rs = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for q in range(n):
        text = np.frombuffer(plants[q % len(plants)][0], dtype=np.uint8)
        off = int(rs.randint(0, len(text) - (kmers + 30) + 1))
        w = text[off:off + kmers + 30].copy()
        m = (0.0, 0.002, 0.005, 0.01)[(q // len(plants)) % 4]
        hit = np.nonzero(rs.random_sample(len(w)) < m)[0]
        for i in hit:
            w[i] = acgt[(int(np.searchsorted(acgt, w[i])) + 1 + int(rs.randint(0, 3))) % 4]
        out.append(w.tobytes())
    return out


def apply_plants(index, plants):
    """index: cobs_amd.Search (a shard plants the documents it holds) or the checker's oracle.Index"""
    for text, docs, keep in plants:
        index.plant(text, docs, keep, salt=PLANT_SALT)


def oracle_index(cfg, plants=None):
    """the checker's own instance of the procedural index (it regenerates rows from the definition; nothing is read back
    from the GPU), with the same planted documents"""
    from oracle import oracle as O
    gen = O.Index.synthetic(1 if cfg["kind"] == "compact" else 0, cfg["term_size"], cfg["canonicalize"], cfg["num_hashes"],
                            cfg["page_size"], cfg["signature_sizes"], cfg["num_docs"], cfg["seed"])
    if plants:
        apply_plants(gen, plants)
    return gen


def cpu_baseline(search, cfg, queries, seconds_target=12.0, check_queries=4, batch=None):
    """Time the oracle (plain-C port of the reference algorithm: per-batch row
    gather -> AND -> LUT/SSE2 expand-add -> threshold -> partial sort) on this
    host's cores, on a bounded sample of the same workload, and compare its
    counts with the GPU's on the same queries."""
    from oracle import oracle as O
    O.build(native=True, force=True)
    info = search.info(0)
    sigs = cfg["signature_sizes"]
    index_bytes = sum(s * (info.page_size if info.kind else info.row_size) for s in sigs)
    avail = 0
    with open("/proc/meminfo") as f:
        for ln in f:
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) * 1024
    kind = 1 if info.kind else 0
    width = info.page_size if info.kind else info.row_size
    sample = "same index, same queries"
    if avail > 2.5 * index_bytes:
        # host copy of the very index the GPU scans (the reference's --load-complete)
        pages = [search.read_rows(0, p, 0, sigs[p]) for p in range(len(sigs))]
        ix = O.Index.from_memory(kind, cfg["term_size"], cfg["canonicalize"], cfg["num_hashes"],
                                 cfg["page_size"], sigs, cfg["num_docs"], pages)
        exact_index = True
    else:
        # not enough host RAM: same geometry with every S_p divided by 16 (row gathers
        # still miss all caches), bits from the same generator
        small = [max(64, s // 16) for s in sigs]
        pages = []
        for p, s in enumerate(small):
            m = np.empty((s, width), dtype=np.uint8)
            for r in range(s):
                m[r] = O.synth_row(kind, cfg["seed"], cfg["page_size"], len(sigs), cfg["num_docs"], p, r, width)
            pages.append(m)
        ix = O.Index.from_memory(kind, cfg["term_size"], cfg["canonicalize"], cfg["num_hashes"],
                                 cfg["page_size"], small, cfg["num_docs"], pages)
        exact_index = False
        sample = "same geometry with S_p/16 (host RAM too small for the full index), same queries"
    # bit-exactness of the GPU counts against the port on the same inputs
    # (the checker regenerates the procedural rows itself: nothing is read back from the GPU for it)
    bit_exact, checked = None, {}
    if batch is not None:
        gen = oracle_index(cfg, cfg.get("plants"))
        n_sum = min(64, len(queries))
        eb = batch.counts_device()[1]
        t_all = batch.counts_tensor()[:n_sum].to(torch.int64)
        if eb > 1:
            t_all = t_all.bitwise_and((1 << (8 * eb)) - 1)      # int16 / int32 views of u16 / u32 scores
        w = (torch.arange(t_all.shape[1], device=t_all.device, dtype=torch.int64) % 1021) + 1
        dev_sum = t_all.sum(dim=1).cpu().numpy()
        dev_wsum = (t_all * w).sum(dim=1).cpu().numpy()
        wn = (np.arange(t_all.shape[1], dtype=np.int64) % 1021) + 1
        bit_exact = True
        n_rows = min(64, len(queries))
        for i in range(n_sum):
            want = gen.counts(queries[i])
            bit_exact = bit_exact and int(dev_sum[i]) == int(want.sum()) \
                and int(dev_wsum[i]) == int((want.astype(np.int64) * wn).sum())
            if i < n_rows:
                bit_exact = bit_exact and bool(np.array_equal(batch.counts_host(i), want))
        checked = {"exact_rows": n_rows, "row_checksums": n_sum}
    # The timed GPU step ends with the per-document scores (the reference's score_list,
    # classic_search.cpp:456-467) in memory, so the CPU figure times the same step:
    # hashes + row gather + AND + expand-add, no threshold filter / ranking.  The rate of
    # the full ClassicSearch::search (which at threshold 0 also partial_sorts all
    # documents) is reported next to it.
    out = {}
    ncores = os.cpu_count() or 1
    share = seconds_target / 4
    for threads in (1, min(ncores, 8)):
        O.timers(reset=True)
        n, dt = O.search_many(ix, queries, -1.0, 0, threads=threads, seconds=share)
        out[threads] = (n / dt, n, dt, O.timers())
    n_full, dt_full = O.search_many(ix, queries, 0.0, 0, threads=1, seconds=share)
    # all cores, the way a throughput-oriented caller would use the reference's algorithm:
    # independent queries on independent threads (each query single-threaded)
    n_all, dt_all = O.search_many_parallel(ix, queries, ncores, -1.0, 0, seconds=share)
    qps_all = n_all / dt_all
    qps1, n1, dt1, tm1 = out[1]
    tmax = max(out)
    T = len(queries[0]) - cfg["term_size"] + 1
    gathered = T * cfg["num_hashes"] * width * (len(sigs) if kind else 1)
    # `value` is the faster of the two: the reference parallelises a query over its document
    # batches (8 sub-index batches at C3, classic_search.cpp:338-341), so its own default is the
    # threaded run; the single-thread figure is kept next to it.
    qpsm, nm, dtm, _ = out[tmax]
    best_threads = tmax if qpsm >= qps1 else 1
    qpsb, nb, dtb = (qpsm, nm, dtm) if best_threads == tmax else (qps1, n1, dt1)
    how = "%d thread(s) over the document batches of one query, as the reference parallelises" % best_threads
    if qps_all > qpsb:
        qpsb, nb, dtb, best_threads = qps_all, n_all, dt_all, min(ncores, len(queries))
        how = "%d threads, one query per thread (all host cores)" % best_threads
    calib = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_cpu_calibration.json")) as f:
            cj = json.load(f)
        calib = {"is": "queries/s of this port / of the REAL reference (BASELINE.md section 2, survey container) on that section's two "
                       "shapes, full search at threshold 0.8, same CPU model (8 vCPU Xeon @ 2.1 GHz), measured in the build container: "
                       "profiles/r06_cpu_calibration.json; BASELINE.md section 4 promised +-15 %",
                 "within_15_percent_everywhere": cj["within_15_percent_everywhere"],
                 "port_over_reference": {k: {t: v["port_over_reference"] for t, v in e.items()} for k, e in cj["shapes"].items()}}
    except Exception:                                               # noqa: BLE001
        calib = {"missing": "profiles/r06_cpu_calibration.json"}
    res = {"value": round(qpsb, 2), "unit": "queries/s", "cores": best_threads, "kind": "port",
           # the port is bit-exact against the reference's known answers; its SPEED against the real reference's: see
           # `calibration` (scripts/calibrate_cpu_baseline.py, run in the build container: the survey's CPU model)
           "calibrated_vs_reference": bool(calib.get("within_15_percent_everywhere", False)),
           "calibration": calib,
           "sample": "%d of the batch's queries, per-document counts (same step as the GPU: hash + "
                     "gather + AND + expand-add), %s, %.1f s; %s; index resident in host RAM"
                     % (nb, how, dtb, sample),
           "kmer_lookups_per_s": round(qpsb * T, 1),
           "gathered_GBps": round(qpsb * gathered / 1e9, 3),
           "phase_seconds_1thread": {k: round(v, 3) for k, v in tm1.items()},
           "host_cores": ncores,
           "threads_1": {"value": round(qps1, 2), "queries": n1},
           "threads_%d_over_document_batches" % tmax: {"value": round(qpsm, 2), "queries": nm},
           "all_cores_one_query_per_thread": {"value": round(qps_all, 2), "queries": n_all, "threads": min(ncores, len(queries))},
           "full_search_with_ranking_1thread": {"value": round(n_full / dt_full, 2), "queries": n_full},
           "bit_exact_vs_gpu": bit_exact, "bit_exact_checked": checked}
    return res


def pack_queries(qs):
    """the query text as one host buffer + offsets (search_packed / sharded_search_arrays)"""
    t = np.frombuffer(b"".join(qs), dtype=np.uint8)
    o = np.zeros(len(qs) + 1, dtype=np.uint64)
    np.cumsum([len(q) for q in qs], out=o[1:])
    return t, o


def end_to_end(search, batch, queries, hit_queries=None):
    """PCIe-inclusive rates of the host-buffer API on the same batch (never `value`):
    query text H2D + K1 + K2 (+ selection) + D2H of the results + host ordering.
    hit_queries: a batch of the same shape whose queries HAVE hits at threshold 0.8 (planted_queries): the
    thresholded call then carries records through compaction, D2H and the host-side ordering."""
    res = {}
    nq = len(queries)

    packed = pack_queries
    text, offsets = packed(queries)
    cases_ = [("threshold_0.8_random_queries", 0.8, 0, text, offsets), ("threshold_0_top10", 0.0, 10, text, offsets)]
    if hit_queries:
        ht, ho = packed(hit_queries)
        cases_.insert(0, ("threshold_0.8_all_hits", 0.8, 0, ht, ho))
    for name, thr, k, tx, of in cases_:
        search.search_packed(tx, of, thr, k)              # sizes the scratch workspaces
        best, tm_best = None, None
        for _ in range(3):
            search.timers(reset=True)
            t0 = time.perf_counter()
            offs, hits = search.search_packed(tx, of, thr, k)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, tm_best = dt, search.timers()
        res[name] = {"queries_per_s": round(nq / best, 1), "seconds": round(best, 4), "hits": int(len(hits)),
                     "hit_bytes_to_host": int(len(hits)) * 12,
                     "phase_seconds": {k_: round(v, 5) for k_, v in tm_best.items()}}
        if name == "threshold_0.8_all_hits":
            per = np.diff(np.asarray(offs, dtype=np.int64))
            res[name]["queries_with_hits"] = int((per > 0).sum())
            res[name]["hits_per_query_max"] = int(per.max()) if len(per) else 0
            res[name]["is"] = ("queries that are mutated windows of the planted sequences: hit records selected in the scan, compacted "
                               "into the pool, ordered on the device, copied home into the library's arena as the passes finish "
                               "(grown there: the search runs ONCE) and handed over as one array.  The windows of one sequence share "
                               "k-mers, i.e. index rows: the scan of this batch is cache-friendlier than the random batch's "
                               "(compare threshold_0.8_random_queries, same call, 0 hits); phase_seconds: the library's own timers")
    # the reference's default call (threshold 0, no limit; what its own benchmark times, src/cobs.cpp:618-626):
    # EVERY document of every query in rank order.  The rows are ordered on the device (rank_kernels.hip) and
    # the finished 12-byte records cross PCIe; 256 queries per call into a result array the caller keeps.
    nd = min(256, nq)
    sub_text = np.frombuffer(b"".join(queries[:nd]), dtype=np.uint8)
    sub_offs = np.ascontiguousarray(offsets[:nd + 1])
    keep = np.zeros(nd * search.total_counts, dtype=search.HIT_DTYPE)
    search.search_packed(sub_text, sub_offs, 0.0, 0, out=keep)
    best = None
    for _ in range(7):
        t0 = time.perf_counter()
        offs, hits = search.search_packed(sub_text, sub_offs, 0.0, 0, out=keep)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # what crosses PCIe per result: one u32 (score << slot_bits | slot) where both fit 32 bits, else an 8-byte pair (rank.cpp)
    terms = max(len(q) for q in queries[:nd]) - 30
    planes = next(p for p in (4, 8, 10, 12, 16, 20, 24, 32) if p >= max(terms, 1).bit_length())
    slot_bits = max((max(search.total_counts - 1, 1)).bit_length(), 1)
    rec = 4 if slot_bits + planes <= 32 else 8
    if rec == 4 and planes <= 12:
        # full lists: a bit stream of slots + the records per score of every query (rank.cpp: expand_slim)
        rec = ((search.total_counts * slot_bits + 31) // 32 + 2 + (1 << planes)) * 4 / search.total_counts
    # the same call into a FRESH array per call (first-touch page faults of 307 MB, which the library asks to be huge
    # pages), and with the results left in the arena the library keeps on the handle (cobs_gpu_search_batch_view)
    fresh = None
    for _ in range(5):
        buf = np.empty(nd * search.total_counts, dtype=search.HIT_DTYPE)
        t0 = time.perf_counter()
        search.search_packed(sub_text, sub_offs, 0.0, 0, out=buf)
        dt = time.perf_counter() - t0
        fresh = dt if fresh is None else min(fresh, dt)
        del buf
    qlist = [bytes(q) for q in queries[:nd]]
    search.search_view(qlist, 0.0, 0)
    view = None
    for _ in range(7):
        t0 = time.perf_counter()
        vo, vh = search.search_view(qlist, 0.0, 0)
        dt = time.perf_counter() - t0
        view = dt if view is None else min(view, dt)
    same = bool(np.array_equal(vh[:1000], hits[:1000]) and np.array_equal(vh[-1000:], hits[-1000:]) and len(vh) == len(hits))
    del vo, vh
    res["default_call_all_ranked"] = {"queries_per_s": round(nd / best, 1), "seconds": round(best, 5), "queries": nd,
                                      "result_array": "kept by the caller",
                                      "fresh_result_array": {"queries_per_s": round(nd / fresh, 1), "seconds": round(fresh, 5)},
                                      "library_arena_view": {"queries_per_s": round(nd / view, 1), "seconds": round(view, 5),
                                                             "same_results": same},
                                      "results": int(len(hits)), "pcie_record_bytes": round(rec, 3),
                                      "pcie_record_GBps": round(len(hits) * rec / best / 1e9, 2),
                                      "host_result_GBps": round(len(hits) * 12 / best / 1e9, 2),
                                      "ranked": "on the device; compare cpu_baseline.full_search_with_ranking_1thread"}
    del keep
    # threshold 0, every document scored: the scores themselves have to cross PCIe
    t = batch.counts_tensor()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.run(0.0, 0)
    host.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    batch.sync()
    res["threshold_0_all_scores_to_pinned_host"] = {
        "queries_per_s": round(nq / dt, 1), "seconds": round(dt, 4),
        "d2h_GB": round(t.numel() * t.element_size() / 1e9, 3)}
    return res


def unique_row_bytes(cfg, nq, kmers, row_bytes):
    """Expected bytes of DISTINCT index rows a batch looks up: sum over sub-indexes of
    row_bytes * S_p * (1 - exp(-Q*T*H / S_p)) (uniform hashes); every further look-up of a row can be served by a
    cache, these bytes have to come from HBM at least once.  -> (unique bytes, looked-up bytes)"""
    import math
    n = float(nq) * kmers * cfg["num_hashes"]
    uniq = sum(row_bytes * sp * (1.0 - math.exp(-n / sp)) for sp in cfg["signature_sizes"])
    return uniq, n * row_bytes * len(cfg["signature_sizes"])


def cache_cold_probe(search, cfg, queries, kmers, row_bytes, nq=256, steps=10):
    """The scan at a batch size where the caches cannot help: `nq` queries look up rows of which
    < 15 % repeat inside the batch, and one launch touches 3 GB at C3 -- twelve times the Infinity Cache -- so a line
    is gone before the next launch asks for it again.  Its algorithmic bandwidth IS HBM-side bandwidth (up to that
    repeat fraction).  The grid of such a batch does not fill the device for long: a lower bound of what the pins give."""
    nq = min(nq, len(queries))
    b = cobs_amd.Batch(search)
    b.set_queries(queries[:nq])
    for _ in range(2):
        b.run(0.0, 0)
    b.sync()
    b.kernel_ms()
    for _ in range(steps):
        b.run(0.0, 0)
    b.sync()
    ms = b.kernel_ms()["scan_ms"]
    algo = b.stats()["algorithmic_bytes"]
    uniq, looked = unique_row_bytes(cfg, nq, kmers, row_bytes)
    score_bytes = algo - looked if algo > looked else 0
    res = {"queries": nq, "scan_ms": round(ms, 4), "algorithmic_bytes": algo,
           "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "repeat_frac": round(1.0 - uniq / looked, 4) if looked else None,
           "hbm_side_lower_bound": round((uniq + score_bytes) / (ms * 1e-3) / 1e9, 1),
           "hbm_side_lower_bound_frac": round((uniq + score_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "note": "rows looked up once per launch and evicted (one launch touches %.1f GB) before the next: achieved x (1 - repeat_frac) "
                   "left the HBM pins at least" % (looked / 1e9)}
    del b
    return res


def hbm_side_probe(args, queries, dev, scale=8.0, exact_rows=16):
    """What leaves the HBM pins at FULL occupancy.  The headline batch -- the same 10 000 queries x
    1000 k-mers, the same kernel and launch geometry -- against the C3 geometry with every sub-index `scale` times as
    long: 147 GB resident in the 288 GB of one MI355X at scale 8.  A row is then looked up 0.3-5 times per batch instead
    of 2.5-40 times, the working set of a 128-byte tile column (S_p x 128 B: 256 MB ... 4 GB) no longer fits the 256 MB
    Infinity Cache, and 56 % of the algorithmic bytes are DISTINCT rows that have to cross the pins: `unique_frac` is a
    floor of the pin-side fraction from a grid that fills the device for 21 ms (cache_cold's 256 queries fill it for half
    a millisecond).  A quarter / a tenth of the batch on the same index repeat even less (unique / algorithmic 0.82 / 0.92)
    and still fill the device for 5 / 2 ms.
    Parity: `exact_rows` rows of the batch element by element against rows the oracle regenerates for THAT index."""
    from oracle import oracle as O
    cfg = c3_config(scale)
    cfg["num_hashes"] = args.num_hashes
    if not args.no_plants:
        cfg["plants"] = planted_documents(cfg, args.kmers)
    s = make_index(cfg, dev)
    info0 = s.info(0)
    row_bytes = int(info0.page_size)
    out = {"index_bytes": int(sum(cfg["signature_sizes"]) * row_bytes), "scale": scale,
           "workload": "the headline batch on the C3 geometry with S_p x %g (%.0f GB resident)"
                       % (scale, sum(cfg["signature_sizes"]) * row_bytes / 1e9)}
    b = cobs_amd.Batch(s)
    for name, nq in (("batch", len(queries)), ("quarter_batch", max(1, len(queries) // 4)), ("tenth_batch", max(1, len(queries) // 10))):
        b.set_queries(queries[:nq])
        for _ in range(2):
            b.run(0.0, 0)
        b.sync()
        b.kernel_ms()
        for _ in range(5):
            b.run(0.0, 0)
        b.sync()
        ms = b.kernel_ms()["scan_ms"]
        algo = b.stats()["algorithmic_bytes"]
        uniq, looked = unique_row_bytes(cfg, nq, args.kmers, row_bytes)
        score_bytes = max(0, algo - looked)
        ent = {"queries": nq, "scan_ms": round(ms, 4), "algorithmic_bytes": algo, "unique_bytes": int(uniq + score_bytes),
               "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "unique_GBps": round((uniq + score_bytes) / (ms * 1e-3) / 1e9, 1),
               "unique_frac": round((uniq + score_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "unique_over_algorithmic": round((uniq + score_bytes) / algo, 4)}
        if name == "batch":
            out.update(ent)
            gen = oracle_index(cfg, cfg.get("plants"))
            ok = True
            step = max(1, nq // exact_rows)
            rows = list(range(0, nq, step))[:exact_rows]
            for i in rows:
                ok = ok and bool(np.array_equal(b.counts_host(i), gen.counts(queries[i])))
            out["bit_exact_vs_oracle"] = ok
            out["exact_rows_checked"] = len(rows)
        else:
            out[name] = ent
    out["is"] = ("frac: algorithmic bytes / scan time / 8 TB/s on this index; unique_frac: the distinct rows of the batch (+ the scores "
                 "written) / scan time / 8 TB/s -- bytes that cannot come from a cache, a floor of what left the HBM pins")
    del b
    s.close()
    return out


def kernels_hash():
    """identifies the kernel source a profile belongs to (the GPU box has no .git)"""
    import hashlib
    h = hashlib.sha256()
    # (the scan kernels and what chooses their launch geometry: what the replayed PMC traffic depends on)
    for fn in ("kernels.hip", "geometry.cpp"):
        with open(os.path.join(ROOT, "cobs_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def make_index(cfg, dev, rank=0, world=1, hbm_budget=0, path=None):
    # Which balance a shard split wants: a RESIDENT shard's time is the work of its gather -- the columns it holds
    # (shard_mode 0); a shard STREAMED under an HBM budget is bound by the bytes that cross its PCIe link per pass --
    # rows x columns (shard_mode 2: equal bytes per rank).
    if path:
        return cobs_amd.Search(path, device=dev, shard_rank=rank, shard_count=world, hbm_budget=hbm_budget,
                               shard_mode=2 if hbm_budget else 0)
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"],
                                  page_size=cfg["page_size"], term_size=cfg["term_size"],
                                  canonicalize=cfg["canonicalize"], num_hashes=cfg["num_hashes"],
                                  seed=cfg["seed"], device=dev, shard_rank=rank, shard_count=world,
                                  hbm_budget=hbm_budget)
    if cfg.get("plants") and not hbm_budget:
        apply_plants(s, cfg["plants"])          # true positives (a shard plants the documents it holds)
    return s


class ShardedRun:
    """north_star's multi-GPU layout: the index sharded by sub-index block over the ranks (equal work per rank: a cut may
    fall inside a sub-index), ONE query batch shared by all ranks, every rank scans its slice for the whole batch, then
    one exchange of the per-document counts over RCCL / xGMI.  The step is the LIBRARY's (cobs_gpu_sharded_batch_step,
    sharded.cpp): the batch cut into nsub sub-batches whose hashing, scan and exchange overlap on three streams tied by
    events, also across steps -- this class only builds the index and calls it.  mode ALLTOALL: rank j ends up with the
    complete count rows (global document order) of the queries [nq*j/N, nq*(j+1)/N); ALLGATHER: every rank with every row.
    comm None (--dist-backend gloo: several ranks on ONE GPU, or the preflight's last resort) executes the library's own
    exchange plan (cobs_gpu_exchange_plan) with torch.distributed transfers through host memory instead."""

    def __init__(self, cfg, queries, world, rank, dev, comm, nsub=1, threshold=0.0, mode=None, hbm_budget=0,
                 path=None, backend="nccl"):
        from cobs_amd import _capi
        self.world, self.rank, self.comm, self.backend = world, rank, comm, backend
        self.mode = _capi.XCHG_ALLTOALL if mode is None else mode
        self.threshold = threshold
        self.s = make_index(cfg, dev, rank, world, hbm_budget, path)
        nsub = max(1, min(nsub, len(queries)))
        self.steps_seen = 0
        if comm is not None:
            self.sb = cobs_amd.ShardedBatch(self.s, comm, nsub)
            self.sb.set_queries(queries)
            self.sub = self.sb.subs
            self.sub_queries = [queries[b.q_begin:b.q_begin + b.nq] for b in self.sub]
            return
        from cobs_amd.distributed import shard_slots
        self.sub, self.sub_queries = [], []
        for i in range(nsub):
            bi = cobs_amd.Batch(self.s)
            qi = queries[i * len(queries) // nsub:(i + 1) * len(queries) // nsub]
            bi.set_queries(qi)
            self.sub.append(bi)
            self.sub_queries.append(qi)
        self.layouts = shard_slots(self.s, None)
        self.x_host_s = 0.0                 # host seconds inside the exchange
        self.rows = [None] * nsub           # (q_begin, q_count, assembled rows) of the last step
        self.moved = 0

    def step(self):
        self.steps_seen += 1
        if self.comm is not None:
            self.sb.step(self.threshold, self.mode)
            return
        from cobs_amd.distributed import exchange_counts_by_plan
        self.moved = 0
        for i, bi in enumerate(self.sub):
            bi.run(self.threshold, 0)
            bi.sync()
            t0 = time.perf_counter()
            self.rows[i] = exchange_counts_by_plan(bi.counts_tensor(), self.layouts, self.s.total_counts, len(self.sub_queries[i]),
                                                   self.mode, None)
            torch.cuda.synchronize()
            self.x_host_s += time.perf_counter() - t0
            q0, qn, rows = self.rows[i]
            self.moved += qn * (self.s.total_counts - sum(c for (_, c, _) in self.layouts[self.rank])) * rows.element_size()

    def finish(self):
        """-> per step on this rank, summed over the sub-batches:
        (scan ms, hash ms, exchange ms, algorithmic bytes, bytes received)"""
        if self.comm is not None:
            self.sb.sync()
            t = self.sb.times()
            return t["scan_ms"], t["hash_ms"], t["exchange_ms"], t["algorithmic_bytes"], t["received_bytes"]
        scan = hsh = algo = 0
        for bi in self.sub:
            bi.sync()
            ms = bi.kernel_ms()
            scan, hsh, algo = scan + ms["scan_ms"], hsh + ms["hash_ms"], algo + bi.stats()["algorithmic_bytes"]
        return scan, hsh, self.x_host_s * 1e3 / max(self.steps_seen, 1), algo, self.moved

    def drop_warmup_events(self):
        self.finish()
        self.steps_seen = 0
        if self.comm is None:
            self.x_host_s = 0.0

    def owned_rows(self, i):
        """after the last step: (first query, query count, rows [count, total_counts]) this rank holds of sub-batch i,
        in GLOBAL document order -- what the exchange assembled"""
        return self.rows[i] if self.comm is None else self.sub[i].global_counts_tensor()

    def local_rows(self, i):
        return self.sub[i].counts_tensor()


def _as_int64(t):
    """count rows as int64 (the int16 / int32 views of u16 / u32 scores carry their bit patterns)"""
    eb = t.element_size()
    v = t.to(torch.int64)
    return v.bitwise_and((1 << (8 * eb)) - 1) if eb > 1 else v


def _row_sums(rows, weights, chunk=512):
    """-> (sum, weighted sum) per row, int64 on the device; rows [n, m] any count width, weights int64 [m]"""
    n = rows.shape[0]
    a = torch.zeros(n, dtype=torch.int64, device=rows.device)
    b = torch.zeros(n, dtype=torch.int64, device=rows.device)
    for r0 in range(0, n, chunk):
        v = _as_int64(rows[r0:r0 + chunk])
        a[r0:r0 + chunk] = v.sum(dim=1)
        b[r0:r0 + chunk] = (v * weights).sum(dim=1)
    return a, b


def verify_sharded(run, cfg, world, rank, backend, corrupt=False, seconds=20.0, exact_rows=16, checksum_rows=64,
                   exchanged=True):
    """The N > 1 line proves itself.  After the timed region, on the rows of the LAST step:
    (a) EVERY row of the batch: each rank sums its local count slices, weighted by the GLOBAL slot of every count
        (plain sum and position-weighted sum), the partial sums of all ranks are added (one all-reduce of the checker,
        not of the data path), and the owner of every assembled row compares the sums of what the exchange delivered --
        a count lost, duplicated, routed to the wrong query or assembled at the wrong document changes them;
    (b) a sample of the rows this rank owns against rows the ORACLE regenerates from the index's procedural
        definition (nothing is read back from the GPU for them): `exact_rows` compared element by element,
        `checksum_rows` by the same two sums, as many as `seconds` of host time allow.
    The flag is the AND over all ranks.  corrupt: test hook -- one count of one assembled row (beyond the sampled ones)
    is changed on the last rank before the check, which must then fail."""
    from oracle import oracle as O
    if rank == 0:
        O.build()                       # (the checker's library: one rank compiles it if it is missing, the others wait)
    if world > 1:
        dist.barrier()
    s = run.s
    total = s.total_counts
    dev = torch.device("cuda", torch.cuda.current_device())
    w_glob = (torch.arange(total, device=dev, dtype=torch.int64) % 1021) + 1
    # the global slot of every local count of this rank's rows (files' held slots back to back)
    gidx = []
    for f in range(s.num_files):
        i = s.info(f)
        gidx.append(torch.arange(int(i.slot_count), device=dev, dtype=torch.int64) + int(i.doc_offset) + int(i.slot_begin))
    gidx = torch.cat(gidx) if gidx else torch.zeros(0, dtype=torch.int64, device=dev)
    w_loc = (gidx % 1021) + 1
    gen = oracle_index(cfg, cfg.get("plants"))
    wn = (np.arange(total, dtype=np.int64) % 1021) + 1
    ok_exchange, ok_oracle = True, True
    n_all = n_exact = n_sum = 0
    t_end = time.perf_counter() + seconds
    nsub = len(run.sub)
    for i in range(nsub):
        qi = run.sub_queries[i]
        q0, qn, rows = run.owned_rows(i)
        if corrupt and rank == world - 1 and i == nsub - 1 and qn > 0:
            r = qn - 1                                    # the last owned row: outside every oracle sample
            rows[r, total // 2] = rows[r, total // 2] ^ 1
        # (a) expected sums of every row of the sub-batch from the shards' own slices
        pa, pb = _row_sums(run.local_rows(i), w_loc)
        part = torch.stack([pa, pb])
        if world > 1 and exchanged:
            part = part.cpu()
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            part = part.to(dev)
        ga, gb = _row_sums(rows, w_glob)
        if qn:
            ok_exchange = ok_exchange and bool(torch.equal(ga, part[0][q0:q0 + qn])) and bool(torch.equal(gb, part[1][q0:q0 + qn]))
        n_all += qn
        # (b) the oracle on a sample of the owned rows
        want_exact = (exact_rows * (i + 1)) // nsub - (exact_rows * i) // nsub
        want_sum = (checksum_rows * (i + 1)) // nsub - (checksum_rows * i) // nsub
        ga_h, gb_h = ga.cpu().numpy(), gb.cpu().numpy()
        for j in range(min(qn, max(want_exact, want_sum))):
            if j >= want_exact and time.perf_counter() > t_end:
                break
            want = gen.counts(qi[q0 + j])
            good = int(ga_h[j]) == int(want.sum()) and int(gb_h[j]) == int((want.astype(np.int64) * wn).sum())
            n_sum += 1
            if j < want_exact:
                got = _as_int64(rows[j]).cpu().numpy()
                good = good and bool(np.array_equal(got, want.astype(np.int64)))
                n_exact += 1
            ok_oracle = ok_oracle and good
    flags = torch.tensor([1 if ok_exchange else 0, 1 if ok_oracle else 0, n_all, n_exact, n_sum], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    fl = [int(v) for v in flags.tolist()]
    return {"bit_exact_vs_oracle": fl[0] == 1 and fl[1] == 1,
            "exchange_consistent_all_rows": fl[0] == 1, "oracle_sample_exact": fl[1] == 1,
            "checked_per_rank_at_least": {"rows_exchange_sums": fl[2], "rows_exact_vs_oracle": fl[3],
                                          "rows_checksummed_vs_oracle": fl[4]},
            "how": "every assembled row: (sum, slot-weighted sum) against the sums of the shards' local slices; sampled owned rows "
                   "against rows the oracle regenerates (element by element / by the same sums); AND over ranks"}


def timed(step, steps, warmup, world, backend, after_warmup=None):
    """the bench contract's timing: warm-up, barrier + synchronize on both sides, max over ranks"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if after_warmup:
        after_warmup()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)      # (the control plane is a gloo group: CPU tensors)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def all_ranks_ok(ok, backend):
    """every rank must agree that its set-up succeeded before the next collective, so that a
    local failure (out of memory, ...) cannot strand the others inside RCCL"""
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def side_measurements(args, cfg, queries, world, rank, dev, comm):
    """N > 1, after the headline: the same job in its other forms, a few steps each (never `value`)."""
    from cobs_amd import _capi
    out = {}
    steps = max(1, min(args.steps, 3))
    backend = args.dist_backend

    def measure(name, build, describe):
        run, err = None, ""
        try:
            run = build()
        except Exception as e:                                      # noqa: BLE001
            err = repr(e)
        if not all_ranks_ok(run is not None, backend):
            out[name] = {"skipped": err or "set-up failed on another rank"}
            return
        dt = timed(run.step, steps, 1, world, backend, run.drop_warmup_events)
        scan, _, _, _, moved = run.finish()
        out[name] = dict(describe, queries_per_s=round(run.nq_total * steps / dt, 1),
                         ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
                         scan_ms_per_step_rank0=round(scan, 4) if scan else None,
                         received_bytes_per_step_rank0=moved)
        del run
        torch.cuda.empty_cache()

    def sharded(nsub, mode, cfg_=cfg, queries_=queries):
        r = ShardedRun(cfg_, queries_, world, rank, dev, comm, nsub, args.threshold, mode, backend=backend)
        r.nq_total = len(queries_)
        return r

    measure("sharded_one_sub_batch", lambda: sharded(1, _capi.XCHG_ALLTOALL),
            {"parallelism": "as the headline without the cut into sub-batches: hash, scan and exchange one after the other"})
    measure("sharded_4_sub_batches", lambda: sharded(4, _capi.XCHG_ALLTOALL),
            {"parallelism": "as the headline, batch cut in 4"})
    measure("sharded_allgather", lambda: sharded(1, _capi.XCHG_ALLGATHER),
            {"parallelism": "as the headline, but every rank receives every count row (N-1 times the traffic)"})

    class Replicated:
        """round 1's headline: index replicated, one batch per rank, no data-path collective (weak scaling)"""
        def __init__(self):
            self.s = make_index(cfg, dev)
            self.b = cobs_amd.Batch(self.s)
            self.b.set_queries(make_queries(args.queries, args.kmers, seed=42 + rank))
            self.nq_total = args.queries * world
        def step(self):
            self.b.run(args.threshold, 0)
        def drop_warmup_events(self):
            self.b.sync()
            self.b.kernel_ms()
        def finish(self):
            self.b.sync()
            return self.b.kernel_ms()["scan_ms"], 0, 0, 0, 0
    measure("index_replicated_weak", Replicated,
            {"parallelism": "index replicated on every GPU, one %d-query batch per GPU, no collective" % args.queries,
             "scaling": "weak"})

    class HitsMode:
        """threshold 0.8: the scan keeps no score rows, only hit records leave a shard (sizes first)"""
        def __init__(self):
            self.s = make_index(cfg, dev, rank, world)
            self.b = cobs_amd.Batch(self.s)
            self.b.set_queries(queries)
            self.nq_total = len(queries)
        def step(self):
            self.b.run_hits(0.8, 0)
            self.b.sync()
            if comm is not None:
                self.b.exchange_hits_owned(comm, 0)
        def drop_warmup_events(self):
            self.b.kernel_ms()
        def finish(self):
            return self.b.kernel_ms()["scan_ms"], 0, 0, 0, self.b.exchange_bytes() if comm is not None else 0
    measure("sharded_hits_threshold_0.8", HitsMode,
            {"parallelism": "sharded as the headline; hits-only scan + sizes-first exchange of hit records, each to the rank that owns its query"})

    if args.config == "c3" and args.scale == 1.0:
        c4 = c4_config(args.scale)
        q4 = make_queries(1000, args.kmers)
        measure("configs3_1M_docs_sharded", lambda: sharded(1, _capi.XCHG_ALLTOALL, c4, q4),
                {"workload": "BASELINE configs[3]: compact index, 1000000 docs, 245 sub-indexes (68 GB) sharded over "
                             "%d ranks, batch of 1000 queries x %d k-mers" % (world, args.kmers)})
    return out


def sharded_product_calls(s, comm, cfg, args, world):
    """N > 1 (and --one-rank-sharded), never `value`: what a CALLER of the multi-GPU layout gets -- host buffers in,
    finished result lists out -- from cobs_gpu_sharded_search_batch (the call behind cobs_gpu_multi_search_batch,
    cobs_gpu::ShardedClassicSearch and `cobs_gpu_query -d`): passes pipelined inside the library, one all-gathered
    status record per pass, hit records exchanged over the communicator and ordered on the device (sharded.cpp).
    Measured AFTER the headline and its self-check; whatever goes wrong here is recorded here and leaves the line alone
    (every call is followed by an agreement of the ranks over the control plane, so nobody waits for a rank that left)."""
    hit_q = planted_queries(cfg["plants"], args.queries, args.kmers)
    packed = pack_queries(hit_q)
    best, n_hits, err = None, 0, ""
    for i in range(4):                       # the first call sizes the workspaces and the result buffer
        ok, dt = True, 0.0
        try:
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            _, h = s.sharded_search_arrays(comm, packed, 0.8, 0)
            dt = time.perf_counter() - t0
            n_hits = int(len(h))
        except Exception as e:                                      # noqa: BLE001
            ok, err = False, repr(e)[:300]
        if not all_ranks_ok(ok, args.dist_backend):
            return {"sharded_search_batch_threshold_0.8": {"skipped": err or "the call failed on another rank"}}
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if i:
            best = dt if best is None else min(best, dt)
    return {"sharded_search_batch_threshold_0.8": {
        "queries_per_s": round(len(hit_q) / best, 1), "seconds": round(best, 4), "hits": n_hits, "ranks": world,
        "is": "cobs_gpu_sharded_search_batch on every rank, the planted batch (queries with hits) at the CLI's default threshold: "
              "upload + K1 of pass i+1 | K2 of pass i | agreement, hit exchange and ordering of pass i-1, inside the library; "
              "every rank returns every query's list; max over ranks, best of 3"}}


def workload_text(args, cfg):
    if args.config == "c2":
        return ("BASELINE configs[1]: synthetic classic index, %d docs x %d rows, batch of %d queries x %d k-mers"
                % (cfg["num_docs"], cfg["signature_sizes"][0], args.queries, args.kmers))
    t = ("BASELINE configs[%d]: synthetic compact index, %d docs, %d sub-indexes, page_size %d B, "
         "S_p %d..%d rows (%.1f GB in HBM), batch of %d queries x %d k-mers, H=%d, threshold %g"
         % ({"c3": 2, "c4": 3, "c5": 4}[args.config], cfg["num_docs"], len(cfg["signature_sizes"]), cfg["page_size"],
            cfg["signature_sizes"][0], cfg["signature_sizes"][-1],
            sum(cfg["signature_sizes"]) * cfg["page_size"] / 1e9,
            args.queries, args.kmers, cfg["num_hashes"], args.threshold))
    if args.num_results:
        t += ", top-%d selected on device (%s)" % (args.num_results, "from score rows" if args.topk_with_rows
                                                   else "per tile in the scan, no score rows")
    if args.hits_only:
        t += ", hits only (no score rows)"
    if args.hbm_budget_gb:
        t += ", streamed under an HBM budget of %g GB per GPU" % args.hbm_budget_gb
    return t


METRIC = "k-mer queries/sec + achieved HBM GB/s, 100k-doc compact index, 1000-kmer query"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries", type=int, default=10000, help="queries per batch (whole job)")
    ap.add_argument("--kmers", type=int, default=1000)
    ap.add_argument("--config", choices=["c3", "c2", "c4", "c5"], default="c3",
                    help="c5 = BASELINE configs[4]: the c3 index as a FILE, streamed under --hbm-budget-gb")
    ap.add_argument("--scale", type=float, default=1.0, help="scale signature sizes (smoke runs)")
    ap.add_argument("--num-hashes", type=int, default=1)
    ap.add_argument("--threshold", type=float, default=0.0,
                    help="0 = benchmark-fpr semantics (all documents scored)")
    ap.add_argument("--hits-only", action="store_true",
                    help="with --threshold > 0: the scan keeps no score rows, only hit records")
    ap.add_argument("--shard-mode", choices=["index", "queries"], default="index",
                    help="N>1: shard the index by sub-index block and exchange the counts over RCCL (north_star's "
                         "layout, the default), or replicate the index and split the work (no collective)")
    ap.add_argument("--num-results", type=int, default=0,
                    help="k > 0: the step also selects the k best documents per query on the device (K3)")
    ap.add_argument("--topk-with-rows", action="store_true",
                    help="with --num-results: keep the score rows (K3 selects from them) instead of selecting per tile in K2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plants", action="store_true",
                    help="the purely random index of rounds 1-4 (no planted documents: no query reaches threshold 0.8)")
    ap.add_argument("--hbm-side-scale", type=float, default=8.0,
                    help="default one-GPU line: also scan the batch against the C3 geometry with S_p x this (8: 147 GB resident; 0 = skip) "
                         "-> roofline.hbm_side")
    ap.add_argument("--exchange-chunks", type=int, default=0,
                    help="sharded runs: sub-batches the batch is cut into; hash(i+1) | scan(i) | exchange(i-1) overlap on their own "
                         "streams.  0 = automatic: 2 (1 for batches below 4000 queries and for an out-of-core run, whose passes are bound by PCIe)")
    ap.add_argument("--corrupt-exchange", action="store_true",
                    help="test hook of the self-check: one count of one assembled row is changed before the check, which must fail")
    ap.add_argument("--exchange", choices=["alltoall", "allgather"], default="alltoall")
    ap.add_argument("--extras", action="store_true",
                    help="N>1: after the headline, also measure the other forms of the job (overlapped sub-batches, all-gather, "
                         "replicated index, hits mode, configs[3]) and report them under other_forms.  Off by default: the "
                         "headline line of a multi-GPU run should not depend on five more set-ups going through")
    ap.add_argument("--no-extras", "--no-sharded-extra", dest="no_extras", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--hbm-budget-gb", type=float, default=0.0, help="per-GPU HBM budget of the index (0 = resident)")
    ap.add_argument("--index-file", default="", help="c5: path of the index file (written once if missing)")
    ap.add_argument("--one-rank-sharded", action="store_true",
                    help="N=1 smoke run of the multi-GPU code path: one-rank RCCL communicator, sharded layout, exchange")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real multi-GPU runs; gloo only "
                    "for smoke-testing the launch path with several ranks on one GPU")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the launch plan of --gpus N (JSON) and exit; needs no GPU")
    ap.add_argument("--no-preflight", action="store_true",
                    help="N>1: skip the RCCL preflight (communicator + first collectives in child processes, with the IPC-mode fallback)")
    ap.add_argument("--preflight-seconds", type=float, default=75.0, help="time limit of one preflight attempt")
    ap.add_argument("--preflight-big-mib", type=int, default=8, help="the preflight's timed all-to-all: MiB per pair of ranks (0 = none)")
    ap.add_argument("--deadline-scale", type=float, default=1.0,
                    help="the watchdog's phase deadlines times this (a phase that overruns ends the run with an error line instead of a hang)")
    ap.add_argument("--step-deadline", type=float, default=0.0,
                    help="seconds the warm-up + timed steps may take before the watchdog ends the run (0 = 150 s + 0.5 s per step)")
    ap.add_argument("--share-devices", action="store_true",
                    help="test hook: several ranks on one GPU with --dist-backend nccl -- real RCCL refuses that, which is how the "
                         "preflight's fallback chain is exercised on a one-GPU box")
    ap.add_argument("--test-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)       # test hook of the watchdog: this rank stops stepping
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ
    if args.dry_run_launch:
        plan = {"gpus": args.gpus, "launched_by": "torch.distributed.run (environment)" if launched else
                ("self" if args.gpus > 1 else "none: one process"),
                "command": None if launched or args.gpus == 1 else
                launch_plan(args.gpus, [a for a in sys.argv[1:] if a != "--dry-run-launch"], __file__),
                "ranks": args.gpus, "devices": list(range(args.gpus))}
        _RESULT_STDOUT.write(json.dumps(plan) + "\n")
        _RESULT_STDOUT.flush()
        return
    if not launched and args.gpus > 1:
        # started plainly: become the launcher of the N ranks (their rank 0 prints the JSON line to
        # the stdout this process was given).  (The device count is asked of a child process: this one must not
        # initialise the ROCm runtime before the ranks have chosen their IPC mode.)
        if args.dist_backend == "nccl" and not args.share_devices and visible_devices() < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, visible_devices()))
        sys.stderr.flush()
        os.dup2(_RESULT_STDOUT.fileno(), 1)
        cmd = launch_plan(args.gpus, sys.argv[1:], __file__)
        os.execv(cmd[0], cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def emit(line):
        _RESULT_STDOUT.write(json.dumps(line) + "\n")
        _RESULT_STDOUT.flush()
        if "error" not in line:
            wd.result_emitted = True

    wd = Watchdog(rank, world, emit, scale=args.deadline_scale, metric=METRIC)
    wd.catch_sigterm()
    wd.start()
    try:
        run_bench(args, world, rank, local_rank, wd, emit)
    except SystemExit:
        raise
    except BaseException as e:                                      # noqa: BLE001
        # no rank dies silently: the traceback to stderr, and from rank 0 a line that says so (another rank's death
        # reaches rank 0 as the launcher's SIGTERM, which the watchdog answers with the same kind of line)
        traceback.print_exc()
        sys.stderr.flush()
        wd.fail("rank %d failed in phase '%s': %s: %s" % (rank, wd.phase_name, type(e).__name__, str(e)[:500]), code=1)
    finally:
        wd.done()


def run_bench(args, world, rank, local_rank, wd, emit):
    from cobs_amd import _capi
    preflight = None
    if world > 1:
        # The control plane of an N-rank run (barriers, the max over ranks of the timing, the self-check's sums) is a
        # gloo process group: it needs no GPU, so it is up BEFORE this process initialises the ROCm runtime -- the
        # preflight may still choose the IPC mode -- and it is not a second RCCL communicator next to the one the data
        # path uses (libcobs_gpu.so's own: `rccl_ranks` in the line).
        wd.phase("process group (gloo control plane)", 120)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: the loopback device; the hostname may not resolve
        dist.init_process_group(backend="gloo")
        try:
            wd.peers = _StorePeers(dist.distributed_c10d._get_default_store())
        except Exception:                                           # noqa: BLE001
            wd.peers = None
        ndev = visible_devices()
        if args.dist_backend == "nccl" and not args.share_devices and ndev < world:
            raise RuntimeError("bench.py --gpus %d: only %d HIP device(s) visible" % (world, ndev))
        local_rank = local_rank % max(ndev, 1)
        if args.dist_backend == "nccl":
            if args.no_preflight:
                preflight = {"skipped": "--no-preflight", "transport": "rccl"}
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            else:
                preflight, transport = run_preflight(args, world, rank, local_rank, wd)
                if transport != "rccl":
                    args.dist_backend = "gloo"
        else:
            preflight = {"skipped": "--dist-backend %s: the exchange does not use RCCL" % args.dist_backend,
                         "transport": args.dist_backend}
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        wd.extra["preflight"] = preflight
        wd.phase("device set-up", 180)
        torch.cuda.set_device(local_rank)
    else:
        wd.phase("device set-up", 300)
        torch.cuda.set_device(0)
    n_gpus = world
    dev = torch.cuda.current_device()
    # several ranks, or an out-of-core run (host staging in the data path): the host side next to the GPU.  A plain
    # one-GPU run keeps all cores: its cpu_baseline leg uses them.
    numa_node = gpu_numa_node(dev)
    numa_cpus = bind_to_numa_node(numa_node) if (world > 1 or args.config == "c5" or args.hbm_budget_gb) else 0

    base = "c3" if args.config == "c5" else args.config
    cfg = {"c3": c3_config, "c2": c2_config, "c4": c4_config}[base](args.scale)
    cfg["num_hashes"] = args.num_hashes
    budget = int(args.hbm_budget_gb * 1e9)
    if args.config != "c5" and not budget and not args.no_plants:
        cfg["plants"] = planted_documents(cfg, args.kmers)
    path = None
    wd.phase("index and batch set-up (%s)" % args.config, {"c4": 900, "c5": 1800}.get(args.config, 600) * max(1.0, args.scale))
    if args.config == "c5":
        # BASELINE configs[4]: the index is a FILE larger than the HBM budget; written once
        path = args.index_file or os.path.join(os.environ.get("TMPDIR", "/tmp"), "cobs_c5_%g.cobs_compact" % args.scale)
        if rank == 0 and not os.path.exists(path):
            cobs_amd.write_synthetic(path, cfg["kind"], cfg["signature_sizes"], cfg["num_docs"],
                                     page_size=cfg["page_size"], num_hashes=cfg["num_hashes"], seed=cfg["seed"], device=dev)
        if world > 1:
            dist.barrier()
        if not budget:
            budget = int(6e9)
            args.hbm_budget_gb = 6.0
    shard_index = (world > 1 or args.one_rank_sharded) and args.shard_mode == "index"
    queries = make_queries(args.queries, args.kmers, seed=42 + (rank if world > 1 and not shard_index else 0))

    comm = None
    run = None
    if shard_index:
        # the native RCCL communicator; torch.distributed is only the launcher that passes the id
        if args.dist_backend == "nccl":
            from cobs_amd.distributed import Comm
            comm = Comm.from_torch(None, dev) if world > 1 else Comm(Comm.unique_id(), 0, 1, dev)
            comm.set_timeout(int(60000 * args.deadline_scale))     # the stream waits the library owns around collectives
            wd.attach(comm)
        ok, err = True, ""
        # Two sub-batches by default: the exchange (~0.2 ms per rank at C3 / N = 8 against 2.4 ms of scan) and K1
        # (replicated on every rank: 0.23 ms) of one half hide behind the scan of the other; every further cut adds a launch
        # tail to every rank's scan and halves what is left to hide.
        # (a small batch is not cut: at configs[3]'s 1000 queries two halves of 500 scan 14 % slower than the whole --
        # fewer look-ups per cached line, smaller grids -- which is more than the exchange costs)
        nsub = args.exchange_chunks if args.exchange_chunks > 0 else (1 if budget or args.queries < 4000 else 2)
        try:
            run = ShardedRun(cfg, queries, world, rank, dev, comm, nsub, args.threshold,
                             _capi.XCHG_ALLGATHER if args.exchange == "allgather" else _capi.XCHG_ALLTOALL,
                             hbm_budget=budget, path=path, backend=args.dist_backend)
        except Exception as e:                                      # noqa: BLE001
            ok, err = False, repr(e)
        if not all_ranks_ok(ok, args.dist_backend):
            raise RuntimeError("set-up of the sharded index failed: " + (err or "on another rank"))
        s, batch = run.s, run.sub[0]

        def step():
            wd.note(step=run.steps_seen)
            if rank == args.test_hang_rank and run.steps_seen >= 1:
                time.sleep(1e6)                 # (test hook: this rank never enters the next exchange)
            run.step()
    else:
        s = make_index(cfg, dev, hbm_budget=budget, path=path)
        batch = cobs_amd.Batch(s)
        batch.set_queries(queries)                 # H2D once; inputs now resident in HBM

        def step():
            if args.num_results > 0:
                batch.run_topk(args.threshold, args.num_results, 0, keep_counts=args.topk_with_rows)
            elif args.hits_only and args.threshold > 0:
                batch.run_hits(args.threshold, 0)
            else:
                batch.run(args.threshold, 0)

    stream_mark = {}

    def drop_warmup_events():
        if run is not None:
            run.drop_warmup_events()
        else:
            batch.sync()
            batch.kernel_ms()
        if budget:
            stream_mark["t0"] = s.stream_traffic()      # what the warm-up asked of PCIe is not the timed steps'

    wd.phase("warm-up and timed steps", args.step_deadline / args.deadline_scale if args.step_deadline > 0 else
             150 + 0.5 * (args.steps + args.warmup) * (20 if budget else 1))
    dt = timed(step, args.steps, args.warmup, world, args.dist_backend, drop_warmup_events)
    wd.phase("collecting the line", 300)
    received, xchg_ms = 0, 0.0
    if run is not None:
        scan_ms, hash_ms, xchg_ms, algo, received = run.finish()     # per step: summed over the sub-batches
    else:
        batch.sync()                               # also raises on invalid bases
        ms = batch.kernel_ms()                     # HIP events on the launch stream, averaged over the timed steps
        scan_ms, hash_ms, algo = ms["scan_ms"], ms["hash_ms"], batch.stats()["algorithmic_bytes"]
    nlaunch = batch.stats()["scan_launches"] if run is None else sum(b.stats()["scan_launches"] for b in run.sub)

    # whole job: sharded index -> one batch on all ranks; replicated index -> N independent batches
    total_queries = args.queries * (1 if shard_index or world == 1 else world)
    ms_per_step = dt / args.steps * 1e3
    qps = total_queries * args.steps / dt
    achieved = algo / (scan_ms * 1e-3) / 1e9
    # HBM traffic from the PMC counters is measured by scripts/profile_shapes.sh (separate
    # rocprofv3 --pmc passes) and replayed here from profiles/traffic.json -- only for the same
    # workload AND the same kernel source; it is not measured in this run
    traffic, traffic_source = None, None
    key = "%s_q%d_k%d_h%d%s%s" % (args.config, args.queries, args.kmers, args.num_hashes,
                                  "_hits" if args.hits_only and args.threshold > 0 else "",
                                  "_top%d%s" % (args.num_results, "rows" if args.topk_with_rows else "") if args.num_results else "")
    if args.scale != 1.0:
        key += "_x%g" % args.scale             # (another index: its own traffic entry)
    tr_path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_path) and n_gpus == 1 and args.scale == 1.0 and not budget:
        try:
            with open(tr_path) as f:
                ent = json.load(f).get(key)
            if ent and ent.get("kernels_hash") == kernels_hash():
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_source = ("replayed from profiles/traffic.json[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                  "of this workload at %s, same kernel source); not measured in this run"
                                  % (key, ent.get("git_head", "?")))
            elif ent:
                traffic_source = "profiles/traffic.json[%s] belongs to other kernel source (%s): dropped" % (key, ent.get("git_head", "?"))
        except Exception:
            traffic = None
    if shard_index:
        par = ("index sharded by sub-index block (" + ("equal bytes per rank: streamed" if budget else "equal work per rank") + ") over %d GPUs, one shared batch, %s exchange of the "
               "per-document counts over RCCL/xGMI inside libcobs_gpu.so%s"
               % (world, "all-to-all (query-owner)" if args.exchange == "alltoall" else "all-gather",
                  ", %d overlapped sub-batches" % len(run.sub) if len(run.sub) > 1 else ""))
    elif world > 1:
        par = "index replicated on %d GPUs, one %d-query batch per GPU, no data-path collective" % (world, args.queries)
    else:
        par = "1 gpu"
    out = {
        # BASELINE.json's metric, verbatim; `value` is the queries/s part, roofline.achieved the GB/s part
        "metric": METRIC,
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong" if shard_index or world == 1 else "weak",
        "vs_baseline": None,
        "dtype": "u32",        # bitwise ops on 32-bit column words (bit-sliced counters); scores leave as u8/u16
        "data": "synthetic",
        "config": {
            "workload": workload_text(args, cfg),
            "global_batch": total_queries,
            "queries_per_gpu_batch": args.queries,
            "kmers_per_query": args.kmers,
            "parallelism": par,
        },
        "kmer_lookups_per_s": round(qps * args.kmers, 1),
        "workload_key": key,
        "kernels_hash": kernels_hash(),
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": "scan_kernel (gather + AND + bit-sliced count), rank 0%s"
                      % (", %d launches per step summed" % nlaunch if nlaunch > 1 else ""),
            "achieved_is": "algorithmic bytes (SURVEY 8d) / kernel time: gathered rows + scores written; includes lines the 256 MB "
                           "Infinity Cache serves (tile-major scheduling), so it is cache-amplified bandwidth, not HBM pin traffic",
            "algorithmic_bytes_per_launch": algo,
            "scan_ms_per_launch": round(scan_ms, 4),
            "hash_ms_per_launch": round(hash_ms, 4),
        },
    }
    if not budget and not (args.hits_only and args.threshold > 0) and not args.num_results:
        # HBM-side view of the same launch: the distinct rows of the batch have to leave the pins at least once, the
        # scores are written once; what is looked up again may come from the Infinity Cache / L2
        info0 = s.info(0)
        row_bytes = int(info0.page_size) if cfg["kind"] == "compact" else int(info0.row_size)
        shards = world if shard_index else 1
        uniq, looked = unique_row_bytes(cfg, args.queries, args.kmers, row_bytes)
        uniq, looked = uniq / shards, looked / shards
        score_bytes = max(0, algo - looked)
        out["roofline"]["unique_bytes_per_launch"] = int(uniq + score_bytes)
        out["roofline"]["hbm_lower_bound_frac"] = round((uniq + score_bytes) / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["roofline"]["unique_bytes_is"] = ("sum_p row_bytes * S_p * (1 - exp(-Q*T*H/S_p)) distinct rows + the scores written: what must "
                                              "cross the HBM pins at least once per launch (lower bound of the pin traffic; achieved / "
                                              "frac above are algorithmic bytes, every look-up counted)")
        if world == 1 and not shard_index and not args.no_cpu_baseline:
            out["roofline"]["cache_cold"] = cache_cold_probe(s, cfg, queries, args.kmers, row_bytes)
    if preflight is not None:
        out["preflight"] = preflight
        out["control_plane"] = "torch.distributed/gloo (barriers, max over ranks of the timing, the self-check's sums); data path: " + \
                               ("RCCL communicator of libcobs_gpu.so" if comm is not None else "torch.distributed/" + args.dist_backend)
    if shard_index:
        out["rccl_ranks"] = comm.size if comm is not None else None       # ncclCommCount
        # what every rank measured with events on its own streams, per step (summed over the sub-batches)
        mine = {"scan_ms": round(scan_ms, 4), "hash_ms": round(hash_ms, 4), "exchange_ms": round(xchg_ms, 4)}
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        out["per_rank"] = {k: [r[k] for r in per_rank] for k in ("scan_ms", "hash_ms", "exchange_ms")}
        out["per_rank"]["numa_node_rank0"] = numa_node
        # of the time hashing and exchanging take, the part that did not show up in the step: the streams overlap them
        # with the scans (of the other sub-batch, of the next step)
        hidden = [max(0.0, min(1.0, (r["scan_ms"] + r["hash_ms"] + r["exchange_ms"] - ms_per_step) /
                                  max(r["hash_ms"] + r["exchange_ms"], 1e-9))) for r in per_rank]
        out["exchange"] = {"mode": args.exchange, "received_bytes_per_step_rank0": received,
                           "sub_batches": len(run.sub),
                           "hidden_frac": round(min(hidden), 4), "hidden_frac_per_rank": [round(h, 4) for h in hidden],
                           "hidden_frac_is": "(scan + hash + exchange ms of a rank - ms_per_step) / (hash + exchange ms), clamped to [0, 1]; min over ranks",
                           "transport": "RCCL in libcobs_gpu.so" if comm is not None else "torch.distributed/" + args.dist_backend}
        # the line proves itself: exchanged rows against the shards' slices (all rows) and against the oracle (sample)
        wd.phase("self-check of the exchanged rows", 300)
        out.update(verify_sharded(run, cfg, world, rank, args.dist_backend, corrupt=args.corrupt_exchange))
        info = s.info(0)
        out["shard_rank0"] = {"hbm_bytes": int(info.hbm_bytes), "slot_begin": int(info.slot_begin),
                              "slot_count": int(info.slot_count)}
        ok_pc = comm is not None and bool(cfg.get("plants")) and not budget
        if all_ranks_ok(ok_pc, args.dist_backend) and ok_pc:
            wd.phase("the product call (cobs_gpu_sharded_search_batch)", 600)
            out["end_to_end"] = sharded_product_calls(s, comm, cfg, args, world)
    if budget:
        info = s.info(0)
        index_bytes = sum(cfg["signature_sizes"]) * (cfg["page_size"] or (cfg["num_docs"] + 7) // 8)
        # what the timed steps asked of PCIe on this rank (the library's own accounting, cobs_gpu_stream_counters): the
        # rows of the chunks copied whole + the looked-up rows of the chunks fetched row by row
        t0 = stream_mark.get("t0", (0, 0, 0, 0))
        t1 = s.stream_traffic()
        fetched, whole = t1[0] - t0[0], t1[1] - t0[1]
        moved = ((t1[2] - t0[2]) + (t1[3] - t0[3])) / max(args.steps, 1)
        pcie = round(moved / (dt / args.steps) / 1e9, 2) if moved else None
        scan_roof = dict(out["roofline"])
        # an out-of-core step is bound by the link the index crosses, not by HBM
        out["roofline"] = {"bound": "pcie", "achieved": pcie, "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                           "frac": round(pcie / PCIE_PEAK_GBS, 4) if pcie else None, "traffic": None,
                           "kernel": "chunk copies / fetch_rows_kernel over PCIe Gen5 x16 (the scans hide behind them)",
                           "scan_kernel_in_hbm": {k: scan_roof[k] for k in ("achieved", "frac", "algorithmic_bytes_per_launch",
                                                                           "scan_ms_per_launch")},
                           "note": "scan_ms_per_launch includes waiting for the chunks"}
        out["streaming"] = {"hbm_budget_bytes": budget, "index_bytes": index_bytes, "file": path,
                            "gpu_numa_node": numa_node, "host_cpus_bound_to_that_node": numa_cpus,
                            "scan_launches_per_step": nlaunch, "chunks_fetched_by_rows": fetched, "chunks_copied_whole": whole,
                            "pcie_bytes_per_step_rank0": int(moved), "pcie_GBps_rank0": pcie}
        buf_b, kept_b, pass_b, nch = s.stream_plan()
        out["streaming"].update({"stream_buffer_bytes": buf_b, "resident_bytes": kept_b, "whole_pass_bytes": pass_b,
                                 "streamed_chunks": nch,
                                 "resident_is": "slices of the streamed file(s) the budget keeps in HBM beside the two stream buffers "
                                                "(round 5: residency per slice, not per file): they do not cross PCIe again"})
    if shard_index and args.extras and not args.no_extras:
        del run, batch, s
        torch.cuda.empty_cache()
        wd.phase("other forms (--extras)", 1800)
        out["other_forms"] = side_measurements(args, cfg, queries, world, rank, dev, comm)
    if (world > 1 and not shard_index) or (world == 1 and budget and not shard_index and args.threshold <= 0 and not args.num_results):
        # index replicated, one batch per rank (or one GPU streaming a file under a budget: no cpu_baseline leg there):
        # every rank checks a sample of ITS rows against the oracle
        class _Own:
            pass
        own = _Own()
        own.s, own.sub, own.sub_queries = s, [batch], [queries]
        own.owned_rows = lambda i: (0, len(queries), batch.counts_tensor())
        own.local_rows = lambda i: batch.counts_tensor()
        out.update(verify_sharded(own, cfg, world, rank, args.dist_backend, exchanged=False))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not budget and not shard_index:
        # the host-buffer calls write their results into this process's memory: the caller sits on the GPU's NUMA node
        # for them (the result array of the default call is 307 MB), and gets all cores back for the CPU baseline
        all_cpus = os.sched_getaffinity(0)
        bind_to_numa_node(numa_node)
        wd.phase("end-to-end calls and CPU baseline", 900)
        hit_q = planted_queries(cfg["plants"], args.queries, args.kmers) if cfg.get("plants") else None
        out["end_to_end"] = end_to_end(s, batch, queries, hit_q)
        out["end_to_end"]["caller_numa_node"] = numa_node
        os.sched_setaffinity(0, all_cpus)
        out["cpu_baseline"] = cpu_baseline(s, cfg, queries, batch=batch)
        out["bit_exact_vs_oracle"] = out["cpu_baseline"]["bit_exact_vs_gpu"]
    if (rank == 0 and world == 1 and not budget and not shard_index and args.config == "c3" and args.scale == 1.0
            and args.hbm_side_scale > 0 and not args.no_cpu_baseline and not (args.hits_only and args.threshold > 0)
            and not args.num_results):
        # the pin-side view at full occupancy: the same batch against the same geometry at 8x the rows (147 GB resident);
        # the 18.4 GB index goes first
        wd.phase("hbm_side probe (C3 x %g)" % args.hbm_side_scale, 600)
        try:
            free_b, total_b = torch.cuda.mem_get_info()
            need = sum(c3_config(args.hbm_side_scale)["signature_sizes"]) * 1664 + (8 << 30)
            del batch
            s.close()
            torch.cuda.empty_cache()
            free_b, total_b = torch.cuda.mem_get_info()
            if free_b < need:
                out["roofline"]["hbm_side"] = {"skipped": "needs %.0f GB of HBM, %.0f GB free" % (need / 1e9, free_b / 1e9)}
            else:
                out["roofline"]["hbm_side"] = hbm_side_probe(args, queries, dev, args.hbm_side_scale)
                if out["roofline"]["hbm_side"].get("bit_exact_vs_oracle") is False:
                    out["bit_exact_vs_oracle"] = False
        except Exception as e:                                      # noqa: BLE001
            out["roofline"]["hbm_side"] = {"skipped": "failed: %r" % (e,)}
    if rank == 0:
        # the line is only printed when it describes the run that was asked for
        assert out["n_gpus"] == args.gpus, (out["n_gpus"], args.gpus)
        if shard_index and args.dist_backend == "nccl":
            assert out["rccl_ranks"] == args.gpus, (out["rccl_ranks"], args.gpus)
        emit(out)
    wd.result_emitted = True            # (every rank: the run's line is out; a problem in the shutdown adds nothing to stdout)
    wd.phase("shutdown", 120)
    if world > 1:
        dist.barrier()              # (rank 0 has printed: nobody's exit makes the launcher end the others before that)
        dist.destroy_process_group()
    wd.done()
    if out.get("bit_exact_vs_oracle") is False:
        # (every rank holds the all-reduced flag: all of them leave with the same code)
        sys.stderr.write("bench.py: the counts are NOT bit-exact against the oracle: %s\n"
                         % json.dumps({k: out.get(k) for k in ("exchange_consistent_all_rows", "oracle_sample_exact")}))
        sys.exit(3)


# stdout carries the ONE JSON line and nothing else: RCCL (the library's own communicator and
# torch.distributed's) prints a version banner with printf, which would land on stdout -- buffered
# until exit, i.e. AFTER the JSON line.  File descriptor 1 is pointed at stderr for the whole run;
# the result goes to a duplicate of the original stdout.
_RESULT_STDOUT = sys.stdout

if __name__ == "__main__":
    sys.stdout.flush()
    _RESULT_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    main()
