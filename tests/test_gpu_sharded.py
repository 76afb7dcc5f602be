"""GPU: the sharded engine end to end -- several processes (all on GPU 0, gloo for the exchange;
on a multi-GPU node the same code runs one rank per GPU over RCCL) each stage and scan their
sub-index block, exchange count slices / hit lists, and every rank gets the oracle's result."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import cases

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, paths, queries, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cobs_amd.distributed import ShardedSearch
        from oracle import oracle as O
        ixs = [O.Index.open(p) for p in paths]
        ss = ShardedSearch(paths if len(paths) > 1 else paths[0], device=0)
        assert ss.search_local.info(0).slot_count < ixs[0].counts_size or world == 1
        got = ss.counts(queries).cpu().numpy().astype(np.int64) & 0xFFFF
        want = np.stack([np.concatenate([ix.counts(q) for ix in ixs]) for q in queries])
        assert np.array_equal(got, want)
        for t, lim in ((0.0, 0), (0.3, 0), (0.3, 4), (0.0, 6), (0.95, 0)):
            res = ss.search_hits(queries, t, lim)
            for q, r in zip(queries, res):
                assert [tuple(x) for x in r] == cases.oracle_results(ixs, q, t, lim), (t, lim)
        r = ss.search(queries[0].decode(), 0.3, 2)
        assert [(x.doc_name, x.score) for x in r] == [(n, s) for (_, _, n, s) in O.search(ixs, queries[0], 0.3, 2)]
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_processes(gpu_lib, oracle, tmp_path, world):
    q_long = oracle.random_sequence(500, 17)
    planted = {5: 1.0, 700: 0.9, 1500: 0.6, 2300: 0.97}
    pa = cases.make_compact(cases.tmp(tmp_path, "s.cobs_compact"), 2400, 64, [900, 1000, 1100, 1200, 1300], 1, 31, 1,
                            0.3, 3, planted=planted, query=q_long)
    pb = cases.make_classic(cases.tmp(tmp_path, "s.cobs_classic"), 1000, 1501, 2, 31, 1, 0.3, 4,
                            planted={9: 1.0, 990: 0.8}, query=q_long)
    queries = [q_long, q_long[:31], q_long[:250], q_long[100:340]]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, [pa, pb], queries, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(os.path.join(str(tmp_path), "ok%d" % r))


def test_bench_sharded_code_path_on_one_rank(gpu_lib):
    """bench.py's N > 1 flow (native RCCL communicator, work-balanced shard, all-to-all exchange,
    the other forms) with a one-rank communicator: the JSON line carries the fields the driver
    and DESIGN.md name; no N > 1 hardware is needed to catch a broken call sequence"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--one-rank-sharded", "--extras", "--scale", "0.02",
                        "--queries", "512", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--exchange-chunks", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    j = json.loads(line)
    assert j["rccl_ranks"] == 1 and j["scaling"] == "strong" and j["n_gpus"] == 1
    assert j["exchange"]["mode"] == "alltoall" and j["exchange"]["transport"].startswith("RCCL")
    assert "sharded by sub-index block" in j["config"]["parallelism"]
    assert j["value"] > 0 and j["roofline"]["frac"] > 0
    _check_self_proving_fields(j, 1)
    other = j["other_forms"]
    for k in ("sharded_one_sub_batch", "sharded_4_sub_batches", "sharded_allgather", "index_replicated_weak",
              "sharded_hits_threshold_0.8"):
        assert "queries_per_s" in other[k], (k, other[k])


def _check_self_proving_fields(j, ranks):
    """what makes an N > 1 line prove itself: the parity flag of the exchanged rows, who took part, what
    every rank measured, and how much of the exchange + hashing the overlapped streams hid"""
    assert j["bit_exact_vs_oracle"] is True and j["exchange_consistent_all_rows"] is True and j["oracle_sample_exact"] is True
    chk = j["checked_per_rank_at_least"]
    assert chk["rows_exchange_sums"] == 512 // ranks and chk["rows_exact_vs_oracle"] >= 16 and chk["rows_checksummed_vs_oracle"] >= 16
    for k in ("scan_ms", "hash_ms", "exchange_ms"):
        assert len(j["per_rank"][k]) == ranks and all(v > 0 for v in j["per_rank"][k]), (k, j["per_rank"])
    x = j["exchange"]
    assert x["sub_batches"] == 2 and 0.0 <= x["hidden_frac"] <= 1.0 and len(x["hidden_frac_per_rank"]) == ranks
    assert "2 overlapped sub-batches" in j["config"]["parallelism"]


def _bench(args, timeout=900, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    return r, (json.loads(lines[-1]) if lines else None), r.stdout


SMALL = ["--scale", "0.02", "--queries", "512", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--exchange-chunks", "2"]


def test_bench_gpus_1_plain_and_one_rank_sharded(gpu_lib):
    """`bench.py --gpus 1` started plainly is the single-GPU line; with --one-rank-sharded the same flag runs the
    multi-GPU code path on a one-rank communicator.  Both lines carry n_gpus == --gpus, and stdout holds the line only."""
    r, j, out = _bench(["--gpus", "1"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["n_gpus"] == 1 and "rccl_ranks" not in j and j["config"]["parallelism"] == "1 gpu"
    assert out.strip().count("\n") == 0
    r, j, out = _bench(["--gpus", "1", "--one-rank-sharded", "--no-extras"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1
    _check_self_proving_fields(j, 1)
    # one sub-batch on request: hash, scan, exchange in turn (the round-3 headline form)
    r, j, out = _bench(["--gpus", "1", "--one-rank-sharded"] + SMALL[:-2] + ["--exchange-chunks", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["exchange"]["sub_batches"] == 1 and j["bit_exact_vs_oracle"] is True


def test_bench_fails_on_a_corrupted_exchange(gpu_lib):
    """the self-check is worth something: with ONE count of ONE assembled row changed (a row outside the oracle's
    sample, so only the all-rows sums can see it) the line says so and the run exits non-zero -- on the one-rank RCCL
    communicator and across two gloo ranks"""
    for extra in (["--gpus", "1", "--one-rank-sharded"], ["--gpus", "2", "--dist-backend", "gloo"]):
        r, j, out = _bench(extra + ["--corrupt-exchange"] + SMALL)
        assert r.returncode != 0, (extra, r.stderr[-2000:])
        assert j is not None and j["bit_exact_vs_oracle"] is False
        assert j["exchange_consistent_all_rows"] is False and j["oracle_sample_exact"] is True
        assert "NOT bit-exact" in r.stderr


def test_bench_gpus_2_launches_its_own_ranks(gpu_lib):
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks by itself (the flag used to be
    parsed and ignored).  Two ranks on the one GPU of this box need gloo for the exchange (RCCL refuses two ranks on
    one device); the launch path, the sharded layout and the line are the ones an 8-GPU node runs."""
    r, j, out = _bench(["--gpus", "2", "--dist-backend", "gloo", "--no-extras"] + SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert "sharded by sub-index block" in j["config"]["parallelism"] and "over 2 GPUs" in j["config"]["parallelism"]
    assert j["shard_rank0"]["slot_count"] < 100352
    _check_self_proving_fields(j, 2)
    assert j["rccl_ranks"] is None and j["exchange"]["transport"] == "torch.distributed/gloo"
    assert [ln for ln in out.splitlines() if ln.strip()] == [ln for ln in out.splitlines() if ln.startswith('{"metric"')]


def test_bench_refuses_more_gpus_than_devices(gpu_lib):
    import torch
    n = torch.cuda.device_count() + 1
    r, j, _ = _bench(["--gpus", str(n)] + SMALL, timeout=300)
    assert r.returncode != 0 and j is None
    assert "HIP device(s) visible" in r.stderr


def test_owner_routed_hits_from_real_shards(gpu_lib, oracle, tmp_path):
    """the owner-routed hit exchange end to end minus the wire: N shards of one index opened in turn on this GPU, each
    runs the hits-only scan and buckets its pool on the device by query owner (what rank r would send), the library's
    cobs_gpu_hit_exchange_plan moves the buckets (played here with slices): every owner ends with exactly the oracle's
    hits of its queries"""
    import ctypes as C
    from cobs_amd import _capi
    lib = _capi.load()
    q_long = oracle.random_sequence(500, 71)
    p = cases.make_compact(cases.tmp(tmp_path, "o.cobs_compact"), 5 * 8 * 64 - 9, 64, [900, 1000, 1100, 1200, 1300], 1, 31, 1,
                           0.3, 3, planted={5: 1.0, 700: 0.9, 1500: 0.6, 2300: 0.97}, query=q_long)
    ix = oracle.Index.open(p)
    queries = [q_long[i:i + 60 + 17 * i] for i in range(11)]
    nq = len(queries)
    want = {qi: sorted((qi, f, d, sc) for (f, d, sc) in cases.oracle_results([ix], q, 0.31, 0)) for qi, q in enumerate(queries)}
    assert sum(len(v) for v in want.values()) > 50
    for N in (2, 3, 5, 8):
        counts, buckets = [], []
        for r in range(N):
            s = gpu_lib.Search(p, shard_rank=r, shard_count=N)
            b = gpu_lib.Batch(s)
            b.set_queries(queries)
            b.run_hits(0.31)
            b.sync()
            c, rec = b.bucketed_hits(N)
            counts.append(c)
            buckets.append(rec)
        flat = (C.c_uint64 * (N * N))(*[c for row in counts for c in row])
        plans = []
        for r in range(N):
            xf = (_capi.Xfer * N)()
            out = (C.c_uint64 * 2)()
            _capi.check(lib.cobs_gpu_hit_exchange_plan(flat, N, r, xf, out))
            plans.append((list(xf), list(out)))
        for i in range(N):
            xf, out = plans[i]
            got = np.zeros((out[0] // 16, 4), dtype=np.uint32)
            for j in range(N):
                peer = plans[j][0][i]
                assert peer.send_bytes == xf[j].recv_bytes
                got[xf[j].recv_offset // 16:(xf[j].recv_offset + xf[j].recv_bytes) // 16] = \
                    buckets[j][peer.send_offset // 16:(peer.send_offset + peer.send_bytes) // 16]
            q0, q1 = nq * i // N, nq * (i + 1) // N
            assert sorted(map(tuple, got.tolist())) == sorted(h for qi in range(q0, q1) for h in want[qi]), (N, i)


def test_bench_replicated_index_line_checks_its_rows(gpu_lib):
    """--shard-mode queries (index replicated, one batch per rank, no collective; weak scaling): every rank checks a
    sample of its own rows against the oracle and the flags are ANDed"""
    r, j, out = _bench(["--gpus", "2", "--dist-backend", "gloo", "--shard-mode", "queries"] + SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "rccl_ranks" not in j
    assert j["bit_exact_vs_oracle"] is True and j["checked_per_rank_at_least"]["rows_exact_vs_oracle"] >= 16


def test_bench_three_ranks_cut_inside_sub_indexes(gpu_lib):
    """an odd rank count: the work-balanced split cuts inside sub-indexes (on whole 128-byte lines), the owners' query
    ranges are uneven (512 queries over 3 ranks in 2 sub-batches), and the line still proves itself"""
    r, j, out = _bench(["--gpus", "3", "--dist-backend", "gloo"] + SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["n_gpus"] == 3 and j["bit_exact_vs_oracle"] is True and j["exchange_consistent_all_rows"] is True
    assert len(j["per_rank"]["scan_ms"]) == 3 and j["checked_per_rank_at_least"]["rows_exchange_sums"] >= 512 // 3 - 1
    cut = j["shard_rank0"]["slot_count"]                        # rank 0's cut: inside a sub-index (12 544 slots each) ...
    assert 0 < cut < 100352 and (cut % 12544) % 1024 == 0       # ... on whole lines: 8 chunks of 16 bytes = 1024 score slots


def _ties_worker(rank, world, port, cases_list, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["COBS_GPU_ROW_RANGE_MIN"] = "48"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cobs_amd.distributed import ShardedSearch
        from oracle import oracle as O
        for no, (paths, queries, mode, budget, combos) in enumerate(cases_list):
            ixs = [O.Index.open(p) for p in paths]
            ss = ShardedSearch(paths if len(paths) > 1 else paths[0], device=0, shard_mode=mode, hbm_budget=budget)
            for t, lim in combos:
                res = ss.search_hits(queries, t, lim)
                for q, r in zip(queries, res):
                    assert [tuple(x) for x in r] == cases.oracle_results(ixs, q, t, lim), (no, paths, mode, budget, t, lim)
            del ss
        open(os.path.join(out_dir, "ties_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_random_ties(gpu_lib, oracle, tmp_path, world):
    """the sharded search over 2 and 3 ranks (processes sharing GPU 0, torch transport of the library's exchange plan)
    on tie-heavy inputs: queries of 1..9 terms on dense filters, one or two files, every shard mode, resident and
    streamed shards, limits that cut through runs of equal scores ACROSS shard boundaries, thresholds, the
    all-documents default call -- every rank gets the oracle's result ((score desc, file, document asc) globally,
    classic_search.cpp:134-145)"""
    rng = np.random.default_rng(5150 + world + 100003 * int(os.environ.get("COBS_FUZZ_SEED", "0")))
    cases_list = []
    for idx in range(6):
        k = int(rng.choice([15, 31]))
        paths = []
        for f in range(int(rng.choice([1, 2]))):
            H = int(rng.choice([1, 2]))
            dens = float(rng.choice([0.3, 0.6]))
            if rng.random() < 0.3:
                D, S = int(rng.integers(300, 4000)), int(rng.integers(200, 1500))
                paths.append(cases.make_classic(cases.tmp(tmp_path, "w%d_%d_%d.cobs_classic" % (world, idx, f)), D, S, H, k, 1, dens, 70 * idx + f))
            else:
                ps = int(rng.choice([16, 64, 136]))
                P = int(rng.integers(2, 7))
                D = (P - 1) * 8 * ps + int(rng.integers(1, 8 * ps + 1))
                sigs = [int(x) for x in rng.integers(150, 1500, size=P)]
                paths.append(cases.make_compact(cases.tmp(tmp_path, "w%d_%d_%d.cobs_compact" % (world, idx, f)), D, ps, sigs, H, k, 1, dens, 70 * idx + f))
        q_long = oracle.random_sequence(200, 6000 + idx)
        queries = [q_long[o:o + k - 1 + int(rng.integers(1, 10))] for o in rng.integers(0, 150, size=int(rng.integers(1, 9)))]
        mode = int(rng.integers(0, 3))
        budget = 0
        if rng.random() < 0.35:
            budget = int(sum(os.path.getsize(p) for p in paths) * 0.7 / world) + 70000
        total = sum(oracle.Index.open(p).num_docs for p in paths)
        combos = [(0.0, 0), (float(rng.choice([0.2, 0.5, 1.0])), 0)]
        for lim in rng.choice([1, 2, 3, 5, 13, 100, total], size=3, replace=False):
            combos.append((float(rng.choice([0.0, 0.0, 0.5])), int(lim)))
        cases_list.append((paths, queries, mode, budget, combos))
    port = _free_port()
    mp.spawn(_ties_worker, args=(world, port, cases_list, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(os.path.join(str(tmp_path), "ties_ok%d" % r))


# ---- round 5: the N-rank line cannot come back empty ---------------------------------------------
def test_bench_two_rank_line_carries_the_preflight_record(gpu_lib):
    """two gloo ranks on the one GPU: the line says what the preflight decided and which control plane ran the barriers"""
    r, j, out = _bench(["--gpus", "2", "--dist-backend", "gloo", "--no-extras"] + SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    assert j["preflight"]["transport"] == "gloo" and "skipped" in j["preflight"]
    assert j["control_plane"].startswith("torch.distributed/gloo")
    _check_self_proving_fields(j, 2)


def test_bench_rccl_preflight_failure_falls_back_to_the_host_transport(gpu_lib):
    """the REAL RCCL refuses two ranks on one device: `--gpus 2 --share-devices` makes exactly that happen to the preflight
    -- one child process per rank and IPC mode, both modes tried, the ranks agree -- and the run still comes back with a
    true line: same shards, same exchange plan, the bytes through host memory, every attempt and error on record"""
    r, j, out = _bench(["--gpus", "2", "--share-devices", "--no-extras", "--preflight-seconds", "90"] + SMALL, timeout=1200)
    assert r.returncode == 0, r.stderr[-4000:]
    pf = j["preflight"]
    assert pf["transport"] == "gloo" and "fallback" in pf
    assert [a["ok"] for a in pf["attempts"]] == [False, False]
    assert sorted(a["HSA_ENABLE_IPC_MODE_LEGACY"] for a in pf["attempts"]) == ["0", "1"]
    assert all(set(a["errors"]) == {"0", "1"} for a in pf["attempts"]), pf
    assert j["rccl_ranks"] is None and j["exchange"]["transport"] == "torch.distributed/gloo"
    _check_self_proving_fields(j, 2)


def test_bench_hung_rank_ends_as_an_error_line_not_as_a_timeout(gpu_lib):
    """rank 1 stops stepping after the first step (test hook): rank 0 waits in the exchange.  The watchdogs end the run
    after --step-deadline: ONE line on stdout with "error", the phase and both ranks' states, a non-zero exit code"""
    import time
    t0 = time.time()
    r, j, out = _bench(["--gpus", "2", "--dist-backend", "gloo", "--no-extras", "--test-hang-rank", "1", "--step-deadline", "10"] + SMALL,
                       timeout=600)
    assert r.returncode != 0
    assert time.time() - t0 < 240
    assert j is not None and j["value"] is None and j["n_gpus"] == 2, (out[-2000:], r.stderr[-3000:])
    assert "did not finish within" in j["error"] and j["phase"] == "warm-up and timed steps"
    per = j["watchdog"]["per_rank"]
    assert per[0]["rank"] == 0 and per[1]["rank"] == 1 and per[1]["step"] >= 1, per
    assert j["preflight"]["transport"] == "gloo"
    assert r.stderr.count("[bench watchdog] rank") >= 2
