#!/usr/bin/env python3
"""tests/mock_rccl/run_fault_ranks.py CASE [N] -- TEST INFRASTRUCTURE (see run_ranks.py).  What the native exchange does when
something goes WRONG between the ranks: N ranks as threads of this process over the stand-in
communicator, which injects the fault (mock_rccl_inject) or -- case `watchdog` -- is simply never entered by one rank.

  send_error     the first ncclSend of rank 1 inside the grouped all-to-all fails: rank 1 must still close its group
                 (RCCL's group state is per thread: left open it would swallow every later call), report ERR_RCCL, abort
                 the dead communicator; nobody waits for the time limit; every later call on every rank fails at once
  size_mismatch  rank 0 sends one byte less than its peer expects: both ends report ERR_RCCL, nobody hangs, the
                 communicators are unusable afterwards and say why
  preflight      cobs_gpu_comm_preflight passes on a healthy communicator (every byte checked) and fails, bounded, on one
                 whose first exchange is short by a byte
  watchdog       bench.py's sharded flow; the last rank stops stepping; bench.Watchdog ends the run with the error line
                 (stdout) and every rank's state (stderr), exit code 4
Prints "ok <case>" (watchdog: the error line) ."""
import ctypes
import json
import os
import sys
import threading
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
assert "libmockrccl.so" in os.environ.get("LD_PRELOAD", ""), "run me through tests/test_gpu_mock_ranks.py (LD_PRELOAD=cobs_amd/libmockrccl.so)"

import bench  # noqa: E402
import cobs_amd  # noqa: E402
from cobs_amd import _capi  # noqa: E402
from cobs_amd.distributed import Comm  # noqa: E402

MOCK = ctypes.CDLL(None)            # the preloaded stand-in's test controls
MOCK.mock_rccl_inject.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
MOCK.mock_rccl_group_depth.restype = ctypes.c_int
MOCK.mock_rccl_aborts.restype = ctypes.c_int


def open_rank(r, N, cfg, queries, uid):
    import torch
    torch.cuda.set_device(0)
    s = bench.make_index(cfg, 0, r, N)
    comm = Comm(uid, r, N, device=0)
    b = cobs_amd.Batch(s)
    b.set_queries(queries)
    return s, comm, b


def fault_in_exchange(kind, N):
    cfg = bench.c3_config(0.01)
    queries = bench.make_queries(23, 120, seed=9)
    uid = Comm.unique_id()
    gate = threading.Barrier(N)
    res, errors = {}, []

    def rank_main(r):
        try:
            s, comm, b = open_rank(r, N, cfg, queries, uid)
            b.run(0.0)
            b.exchange_counts(comm, _capi.XCHG_ALLTOALL)        # a healthy exchange first
            b.sync()
            gate.wait()
            if r == 0:
                MOCK.mock_rccl_inject(1 if kind == "send_error" else 2, 1 if kind == "send_error" else 0, 0)
            gate.wait()
            out = {}
            t0 = time.time()
            b.run(0.0)
            try:
                b.exchange_counts(comm, _capi.XCHG_ALLTOALL)
                b.sync()
                out["first"] = "ok"
            except cobs_amd.CobsGpuError as e:
                out["first"] = (e.status, str(e))
            out["depth"] = MOCK.mock_rccl_group_depth()         # this thread's open groups after the failed call
            out["t_first"] = time.time() - t0
            out["state"] = comm.state()
            gate.wait()
            t0 = time.time()
            b.run(0.0)
            try:
                b.exchange_counts(comm, _capi.XCHG_ALLTOALL)
                b.sync()
                out["second"] = "ok"
            except cobs_amd.CobsGpuError as e:
                out["second"] = (e.status, str(e))
            out["t_second"] = time.time() - t0
            res[r] = out
            del b
            comm.close()
            s.close()
        except BaseException:
            errors.append((r, traceback.format_exc()))
            try:
                gate.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        for r, tb in errors:
            print("rank %d:\n%s" % (r, tb), file=sys.stderr)
        raise SystemExit(1)
    limit = float(os.environ.get("MOCK_RCCL_TIMEOUT_S", "120"))
    for r in range(N):
        assert res[r]["depth"] == 0, (r, res[r])                           # no group left open, on any rank
    if kind == "send_error":
        st, msg = res[1]["first"]
        assert st == _capi.ERR_RCCL and "ncclSend" in msg, res[1]
        assert "BROKEN" in res[1]["state"], res[1]["state"]
        assert MOCK.mock_rccl_aborts() >= 1                                # ncclSystemError: the communicator is dead -> aborted
        # the failing rank closed its group, so its peers met it there instead of waiting for the time limit
        assert all(res[r]["t_first"] < limit * 0.8 for r in range(N)), res
        assert all(res[r]["first"] != "ok" for r in range(N)), res         # its peers miss its sends: they fail too
    else:
        assert res[0]["first"] != "ok" and res[1]["first"] != "ok", res    # both ends of the short message
        assert res[0]["first"][0] == _capi.ERR_RCCL and res[1]["first"][0] == _capi.ERR_RCCL
        assert all(res[r]["t_first"] < limit * 0.8 for r in range(N)), res
    # afterwards: a rank whose communicator failed refuses at once, with the reason; a rank that was not involved
    # (size_mismatch, N = 3: rank 2) enters alone and is released by the time limit -- nobody waits for ever
    for r in range(N):
        if res[r]["first"] != "ok":
            st, msg = res[r]["second"]
            assert st == _capi.ERR_RCCL and "unusable after an earlier failure" in msg and res[r]["t_second"] < 2.0, (r, res[r])
        else:
            assert res[r]["second"] != "ok" and res[r]["t_second"] < limit + 10, (r, res[r])
    print("ok %s %d" % (kind, N))


def preflight_case(N):
    uid = Comm.unique_id()
    uid2 = Comm.unique_id()
    gate = threading.Barrier(N)
    res, errors = {}, []

    def rank_main(r):
        try:
            import torch
            torch.cuda.set_device(0)
            comm = Comm(uid, r, N, device=0)
            out = {"clean": comm.preflight(timeout_ms=20000, big_bytes=64 << 10)}
            comm.close()
            comm = Comm(uid2, r, N, device=0)
            gate.wait()
            if r == 0:
                MOCK.mock_rccl_inject(2, N - 1, 0)
            gate.wait()
            t0 = time.time()
            try:
                comm.preflight(timeout_ms=20000)
                out["faulty"] = "ok"
            except cobs_amd.CobsGpuError as e:
                out["faulty"] = (e.status, str(e))
            out["t"] = time.time() - t0
            out["depth"] = MOCK.mock_rccl_group_depth()
            res[r] = out
            comm.close()
        except BaseException:
            errors.append((r, traceback.format_exc()))
            try:
                gate.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        for r, tb in errors:
            print("rank %d:\n%s" % (r, tb), file=sys.stderr)
        raise SystemExit(1)
    limit = float(os.environ.get("MOCK_RCCL_TIMEOUT_S", "120"))
    for r in range(N):
        c = res[r]["clean"]
        assert c["alltoall_bytes"] > 0 and c["big_alltoall_bytes_received"] == (N - 1) * (64 << 10), (r, c)
        assert res[r]["depth"] == 0
        assert res[r]["t"] < limit + 25, res[r]
    assert res[N - 1]["faulty"] != "ok" and res[N - 1]["faulty"][0] == _capi.ERR_RCCL, res[N - 1]
    assert sum(1 for r in range(N) if res[r]["faulty"] != "ok") >= 2, res          # the sender and its receiver at least
    print("ok preflight %d" % N)


def watchdog_case(N):
    from oracle import oracle
    oracle.build()
    cfg = bench.c3_config(0.01)
    queries = bench.make_queries(40, 150, seed=3)
    uid = Comm.unique_id()

    class Peers:
        def __init__(self):
            self.d, self.lock = {}, threading.Lock()
        def set(self, k, v):
            with self.lock:
                self.d[k] = v
        def get(self, k):
            with self.lock:
                return self.d.get(k)
    peers = Peers()
    out_lock = threading.Lock()

    def emit(line):
        with out_lock:
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()

    def rank_main(r):
        import torch
        torch.cuda.set_device(0)
        wd = bench.Watchdog(r, N, emit, peers=peers, grace=2.0)
        wd.start()
        wd.phase("index and batch set-up", 120)
        comm = Comm(uid, r, N, device=0)
        wd.attach(comm)
        run = bench.ShardedRun(cfg, queries, N, r, 0, comm, nsub=2)
        wd.phase("warm-up and timed steps", 6)
        for i in range(1000):
            wd.note(step=i)
            if r == N - 1 and i == 2:
                time.sleep(1e6)             # this rank never enters the exchange of step 2
            run.step()
            run.sb.sync()

    ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    raise SystemExit("the watchdog did not end the run")


def main():
    case = sys.argv[1]
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if case in ("send_error", "size_mismatch"):
        fault_in_exchange(case, N)
    elif case == "preflight":
        preflight_case(N)
    elif case == "watchdog":
        watchdog_case(N)
    else:
        raise SystemExit("unknown case " + case)


if __name__ == "__main__":
    main()
