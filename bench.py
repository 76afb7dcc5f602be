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
For N > 1 it is launched by torch.distributed.run, one rank per GPU (RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cobs_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


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
    bit_exact = None
    if exact_index and batch is not None:
        bit_exact = True
        for i in range(min(check_queries, len(queries))):
            bit_exact = bit_exact and bool(np.array_equal(batch.counts_host(i), ix.counts(queries[i])))
    # The timed GPU step ends with the per-document scores (the reference's score_list,
    # classic_search.cpp:456-467) in memory, so the CPU figure times the same step:
    # hashes + row gather + AND + expand-add, no threshold filter / ranking.  The rate of
    # the full ClassicSearch::search (which at threshold 0 also partial_sorts all
    # documents) is reported next to it.
    out = {}
    ncores = os.cpu_count() or 1
    share = seconds_target / 3
    for threads in (1, min(ncores, 8)):
        O.timers(reset=True)
        n, dt = O.search_many(ix, queries, -1.0, 0, threads=threads, seconds=share)
        out[threads] = (n / dt, n, dt, O.timers())
    n_full, dt_full = O.search_many(ix, queries, 0.0, 0, threads=1, seconds=share)
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
    res = {"value": round(qpsb, 2), "unit": "queries/s", "cores": best_threads, "kind": "port",
           "sample": "%d of the batch's queries, per-document counts (same step as the GPU: hash + "
                     "gather + AND + expand-add), %d thread(s) over document batches as the reference "
                     "does, %.1f s; %s; index resident in host RAM"
                     % (nb, best_threads, dtb, sample),
           "kmer_lookups_per_s": round(qpsb * T, 1),
           "gathered_GBps": round(qpsb * gathered / 1e9, 3),
           "phase_seconds_1thread": {k: round(v, 3) for k, v in tm1.items()},
           "host_cores": ncores,
           "threads_1": {"value": round(qps1, 2), "queries": n1},
           "threads_%d" % tmax: {"value": round(qpsm, 2), "queries": nm},
           "full_search_with_ranking_1thread": {"value": round(n_full / dt_full, 2), "queries": n_full},
           "bit_exact_vs_gpu": bit_exact}
    return res


def end_to_end(search, batch, queries):
    """PCIe-inclusive rates of the host-buffer API on the same batch (never `value`):
    query text H2D + K1 + K2 (+ selection) + D2H of the results + host ordering."""
    res = {}
    nq = len(queries)
    # the query text as one host buffer + offsets (search_packed); the passes of a call are pipelined
    text = np.frombuffer(b"".join(queries), dtype=np.uint8)
    offsets = np.zeros(nq + 1, dtype=np.uint64)
    np.cumsum([len(q) for q in queries], out=offsets[1:])
    for name, thr, k in (("threshold_0.8_all_hits", 0.8, 0), ("threshold_0_top10", 0.0, 10)):
        search.search_packed(text, offsets, thr, k)              # sizes the scratch workspaces
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            offs, hits = search.search_packed(text, offsets, thr, k)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[name] = {"queries_per_s": round(nq / best, 1), "seconds": round(best, 4), "hits": int(len(hits))}
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


def sharded_setup(s, queries, world, nsub, threshold):
    """--shard-mode index: the batch is cut into sub-batches so that the all-gather of sub-batch i
    (RCCL's own stream, xGMI) overlaps the scan of sub-batch i+1 (the stream the kernels are on).
    RCCL has no 16-bit integer type: the count slices travel as bytes.  -> (step, sub-batches)"""
    sub, gathered = [], []
    nsub = max(1, min(nsub, len(queries)))
    for i in range(nsub):
        bi = cobs_amd.Batch(s)
        bi.set_queries(queries[i * len(queries) // nsub:(i + 1) * len(queries) // nsub])
        sub.append(bi)
        local = bi.counts_tensor().view(torch.uint8).reshape(-1)
        gathered.append(torch.empty((world * local.numel(),), dtype=torch.uint8, device="cuda"))

    def step():
        works = []
        for bi, out in zip(sub, gathered):
            bi.run(threshold, 0)
            # per-document hit counts of the disjoint sub-index blocks -> every rank
            works.append(dist.all_gather_into_tensor(out, bi.counts_tensor().view(torch.uint8).reshape(-1),
                                                     async_op=True))
        for w in works:
            w.wait()
    return step, sub, gathered


def timed(step, steps, warmup, world, backend):
    """the bench contract's timing: warm-up, barrier + synchronize on both sides, max over ranks"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
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
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def sharded_extra(args, cfg, world, rank, dev):
    """N > 1, after the headline (replicated) measurement: the same 10k-query batch against the
    index SHARDED by sub-index block over the ranks, per-document counts all-gathered over
    RCCL/xGMI (north_star's multi-GPU layout; strong scaling).  Every rank must agree that its
    set-up succeeded before the first collective, so a local failure cannot strand the others."""
    ok, err, state = 1, "", None
    try:
        s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"],
                                      page_size=cfg["page_size"], term_size=cfg["term_size"],
                                      canonicalize=cfg["canonicalize"], num_hashes=cfg["num_hashes"],
                                      seed=cfg["seed"], device=dev, shard_rank=rank, shard_count=world)
        queries = make_queries(args.queries, args.kmers)            # the same batch on every rank
        step, sub, gathered = sharded_setup(s, queries, world, args.exchange_chunks, args.threshold)
        state = (s, sub, gathered)
    except Exception as e:                                          # noqa: BLE001
        ok, err = 0, repr(e)
    # all_gather_into_tensor needs equal slices: 8 sub-indexes over 2, 4 or 8 ranks are; anything else is skipped
    n_local = int(state[2][0].numel()) // world if ok else 0
    flag = torch.tensor([ok, n_local, -n_local], dtype=torch.int64,
                        device="cuda" if args.dist_backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag[0].item()) == 0:
        return {"skipped": err or "set-up failed on another rank"}
    if int(flag[1].item()) != -int(flag[2].item()):
        return {"skipped": "ranks hold count slices of different sizes"}
    steps = max(1, min(args.steps, 5))
    dt = timed(step, steps, 1, world, args.dist_backend)
    for bi in sub:
        bi.sync()
    scan_ms = sum(bi.kernel_ms()["scan_ms"] for bi in sub)
    payload = sum(int(g.numel()) for g in gathered)
    # the production form of the same layout: with a threshold only hits leave a shard, so the
    # exchange shrinks to the hit lists (none on random data at 0.8, like the reference's own
    # benchmark); what remains is every rank scanning its sub-index block for the whole batch
    whole = cobs_amd.Batch(s)
    whole.set_queries(queries)
    token = torch.zeros(1, dtype=torch.int64, device="cuda" if args.dist_backend == "nccl" else "cpu")

    def hits_step():
        whole.run(0.8, 0)
        dist.all_reduce(token, op=dist.ReduceOp.SUM)         # stands for the (empty) hit-list exchange

    dth = timed(hits_step, steps, 1, world, args.dist_backend)
    whole.sync()
    hits_scan_ms = whole.kernel_ms()["scan_ms"]
    del whole, state
    return {"queries_per_s": round(args.queries * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "scaling": "strong", "scan_ms_per_step_rank0": round(scan_ms, 4),
            "gathered_bytes_per_step": payload,
            "hits_mode_threshold_0.8": {"queries_per_s": round(args.queries * steps / dth, 1),
                                        "ms_per_step": round(dth / steps * 1e3, 3),
                                        "scan_ms_per_step_rank0": round(hits_scan_ms, 4)},
            "parallelism": "index sharded by sub-index block x%d, %d sub-batches, async RCCL all-gather of the "
                           "count slices overlapped with the next scan" % (world, len(sub))}


def sharded_c4_extra(args, world, rank, dev):
    """N > 1: BASELINE configs[3] -- the 1M-document compact index (245 sub-indexes) sharded by
    sub-index block over the ranks, one 1000-query batch, per-document counts of the (unequal)
    blocks all-gathered over RCCL/xGMI (padded to the largest block)."""
    from cobs_amd.distributed import all_gather_counts
    cfg = c4_config(args.scale)
    nq = 1000
    ok, err, b = 1, "", None
    try:
        s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"],
                                      page_size=cfg["page_size"], term_size=cfg["term_size"],
                                      canonicalize=cfg["canonicalize"], num_hashes=cfg["num_hashes"],
                                      seed=cfg["seed"], device=dev, shard_rank=rank, shard_count=world)
        b = cobs_amd.Batch(s)
        b.set_queries(make_queries(nq, args.kmers))
    except Exception as e:                                          # noqa: BLE001
        ok, err = 0, repr(e)
    flag = torch.tensor([ok], dtype=torch.int64, device="cuda" if args.dist_backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return {"skipped": err or "set-up failed on another rank"}
    sizes = {}

    def step():
        b.run(args.threshold, 0)
        parts = all_gather_counts(b.counts_tensor(), None)
        sizes["bytes"] = sum(int(t.numel()) * t.element_size() for t in parts)

    steps = max(1, min(args.steps, 3))
    dt = timed(step, steps, 1, world, args.dist_backend)
    b.sync()
    scan_ms = b.kernel_ms()["scan_ms"]
    return {"workload": "BASELINE configs[3]: compact index, 1000000 docs, 245 sub-indexes sharded over %d ranks, "
                        "batch of %d queries x %d k-mers" % (world, nq, args.kmers),
            "queries_per_s": round(nq * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "scan_ms_per_step_rank0": round(scan_ms, 4), "gathered_bytes_per_step": sizes.get("bytes", 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=10000, help="queries per batch (whole job)")
    ap.add_argument("--kmers", type=int, default=1000)
    ap.add_argument("--config", choices=["c3", "c2", "c4"], default="c3")
    ap.add_argument("--scale", type=float, default=1.0, help="scale signature sizes (smoke runs)")
    ap.add_argument("--threshold", type=float, default=0.0,
                    help="0 = benchmark-fpr semantics (all documents scored)")
    ap.add_argument("--shard-mode", choices=["queries", "index"], default="queries",
                    help="N>1: replicate the index and split the batch (no collective), or shard the "
                         "index by sub-index block and all-gather the count slices over RCCL")
    ap.add_argument("--num-results", type=int, default=0,
                    help="k > 0: the step also selects the k best documents per query on the device (K3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange-chunks", type=int, default=4,
                    help="--shard-mode index: sub-batches whose all-gather overlaps the next scan")
    ap.add_argument("--no-sharded-extra", action="store_true",
                    help="N>1, default mode: skip the additional index-sharded + RCCL all-gather measurement")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real multi-GPU runs; gloo only "
                    "for smoke-testing the launch path with several ranks on one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    else:
        torch.cuda.set_device(0)
    n_gpus = world
    dev = torch.cuda.current_device()

    cfg = {"c3": c3_config, "c2": c2_config, "c4": c4_config}[args.config](args.scale)
    shard_index = world > 1 and args.shard_mode == "index"
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"],
                                  page_size=cfg["page_size"], term_size=cfg["term_size"],
                                  canonicalize=cfg["canonicalize"], num_hashes=cfg["num_hashes"],
                                  seed=cfg["seed"], device=dev,
                                  shard_rank=rank if shard_index else 0,
                                  shard_count=world if shard_index else 1)
    if world > 1 and not shard_index:
        # weak scaling: every rank serves its own batch of --queries queries against its replica
        mine = make_queries(args.queries, args.kmers, seed=42 + rank)
    else:
        mine = make_queries(args.queries, args.kmers)
    batch = cobs_amd.Batch(s)
    batch.set_queries(mine)                    # H2D once; inputs now resident in HBM

    sub, gathered, sharded_step = [], [], None
    if shard_index:
        sharded_step, sub, gathered = sharded_setup(s, mine, world, args.exchange_chunks, args.threshold)

    def step():
        if shard_index:
            sharded_step()
        elif args.num_results > 0:
            batch.run_topk(args.threshold, args.num_results, 0)
        else:
            batch.run(args.threshold, 0)

    for _ in range(args.warmup):
        step()
    if not shard_index:
        batch.sync()
        batch.kernel_ms()                      # drop warm-up events
    for bi in sub:
        bi.sync()
        bi.kernel_ms()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if shard_index:
        for bi in sub:
            bi.sync()
        parts = [bi.kernel_ms() for bi in sub]  # one launch per sub-batch: a step's scan time is their sum
        ms = {"scan_ms": sum(p["scan_ms"] for p in parts), "hash_ms": sum(p["hash_ms"] for p in parts)}
        st = {"algorithmic_bytes": sum(bi.stats()["algorithmic_bytes"] for bi in sub)}
    else:
        batch.sync()                           # also raises on invalid bases
        ms = batch.kernel_ms()                 # HIP events on the launch stream, averaged over the timed steps
        st = batch.stats()

    # whole job: replicated index -> N independent batches; sharded index -> one batch on all ranks
    total_queries = args.queries * (world if not shard_index else 1)
    ms_per_step = dt / args.steps * 1e3
    qps = total_queries * args.steps / dt
    algo = st["algorithmic_bytes"]
    achieved = algo / (ms["scan_ms"] * 1e-3) / 1e9
    traffic = None
    tr_path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_path) and n_gpus == 1 and args.scale == 1.0 and args.queries == 10000:
        try:
            with open(tr_path) as f:
                tj = json.load(f)
            if tj.get("config") == args.config:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        # BASELINE.json's metric, verbatim; `value` is the queries/s part, roofline.achieved the GB/s part
        "metric": "k-mer queries/sec + achieved HBM GB/s, 100k-doc compact index, 1000-kmer query",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong" if shard_index else "weak",
        "vs_baseline": None,
        "dtype": "u32",        # bitwise ops on 32-bit column words (bit-sliced counters); scores leave as u16
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[%d]: synthetic compact index, %%d docs, %%d sub-indexes, page_size %%d B, "
                         "S_p %%d..%%d rows (%%.1f GB in HBM), batch of %%d queries x %%d k-mers, H=%%d, threshold %%g"
                         % (2 if args.config == "c3" else 3)
                         % (cfg["num_docs"], len(cfg["signature_sizes"]), cfg["page_size"],
                            cfg["signature_sizes"][0], cfg["signature_sizes"][-1],
                            sum(cfg["signature_sizes"]) * cfg["page_size"] / 1e9,
                            args.queries, args.kmers, cfg["num_hashes"], args.threshold)
                         + (", top-%d selected on device" % args.num_results if args.num_results else ""))
            if args.config != "c2" else
            ("BASELINE configs[1]: synthetic classic index, %d docs x %d rows, batch of %d queries x %d k-mers"
             % (cfg["num_docs"], cfg["signature_sizes"][0], args.queries, args.kmers)),
            "global_batch": total_queries,
            "queries_per_gpu_batch": args.queries,
            "kmers_per_query": args.kmers,
            "parallelism": ("1 gpu" if world == 1 else
                            ("index sharded by sub-index block x%d + RCCL all-gather of counts" % world
                             if shard_index else "index replicated on %d GPUs, one %d-query batch per GPU, no data-path collective"
                             % (world, args.queries))),
        },
        "kmer_lookups_per_s": round(qps * args.kmers, 1),
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "kernel": "scan_kernel (gather + AND + bit-sliced count), rank 0",
            "algorithmic_bytes_per_launch": algo,
            "scan_ms_per_launch": round(ms["scan_ms"], 4),
            "hash_ms_per_launch": round(ms["hash_ms"], 4),
        },
    }
    if world > 1 and not shard_index and not args.no_sharded_extra:
        del batch, s
        extra = sharded_extra(args, cfg, world, rank, dev)
        if args.config == "c3":
            extra["c4_1M_docs"] = sharded_c4_extra(args, world, rank, dev)
        out["index_sharded"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["end_to_end"] = end_to_end(s, batch, mine)
        out["cpu_baseline"] = cpu_baseline(s, cfg, mine, batch=batch)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
