"""cobs_amd.launch -- the launcher side of a job of N ranks on one node (one process per GPU): what has to be right
BEFORE the first collective and what must happen when one is never entered.

The reference has no distributed code; the shard boundary is its own data layout (cobs/query/compact_index/
mmap_search_file.cpp:22-27).  What is here belongs to any caller that starts ranks of libcobs_gpu.so -- `bench.py
--gpus N` is one, a torch.distributed service another:
  * launch_plan: the torch.distributed.run command of N ranks on 127.0.0.1;
  * run_preflight / preflight_child: does a communicator of these ranks come up AND move bytes
    (cobs_gpu_comm_preflight), in child processes, under either HSA_ENABLE_IPC_MODE_LEGACY value, with the host
    transport as the last resort;
  * Watchdog: a phase that overruns, a rank that dies or an exception becomes ONE JSON line with "error" and every
    rank's state, instead of a silent time-out;
  * gpu_numa_node / bind_to_numa_node: the host side of a rank next to its GPU.
(Until round 6 all of this lived in bench.py; the overlapped multi-GPU flow itself is in the library: sharded.cpp.)
"""
import json
import os
import select
import signal
import subprocess
import sys
import threading
import time

import torch
import torch.distributed as dist

_RESULT_STDOUT = sys.stdout


def gpu_numa_node(dev):
    """the NUMA node the GPU's PCIe root sits on (sysfs), or None"""
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except Exception:                                               # noqa: BLE001
        return None


def bind_to_numa_node(node):
    """this process (and the host threads the library starts: staging copies, pinned buffers they first touch) on the CPUs
    of `node` -- with 8 ranks on two sockets every rank's host side then sits next to its GPU.  -> CPUs bound to, or 0"""
    if node is None:
        return 0
    try:
        cpus = set()
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return len(cpus)
    except Exception:                                               # noqa: BLE001
        return 0


def visible_devices():
    """HIP devices visible to this job, counted by a child process (so that the caller does not initialise the runtime)"""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:                                               # noqa: BLE001
        return 0


class Watchdog(threading.Thread):
    """A run of N ranks must not end as a silent driver time-out: a collective that one rank never
    enters does not fail, it waits.  One watchdog thread per rank: the main thread names the PHASE it is in and how long
    that may take; when a phase overruns -- or the launcher sends SIGTERM because another rank died -- every rank prints
    what it was doing (phase, step, what its communicator entered last, whether that stream is idle) to stderr and
    leaves it for rank 0, and rank 0 emits ONE JSON line with "error" and the ranks' states, then every rank exits
    non-zero.  Deadlines are the same on every rank and phases are aligned by barriers, so all ranks fire together."""

    def __init__(self, rank, world, emit, peers=None, exit_fn=os._exit, scale=1.0, grace=3.0, metric=None):
        super().__init__(daemon=True, name="bench-watchdog")
        self.metric = metric
        self.rank, self.world, self.emit, self.peers, self.exit_fn = rank, world, emit, peers, exit_fn
        self.scale, self.grace = scale, grace
        self.lock = threading.Lock()
        self.phase_name, self.deadline, self.t_phase = "start", None, time.time()
        self.notes, self.comm, self.extra = {}, None, {}
        self.stop_ev = threading.Event()
        self.fired = False
        self.result_emitted = False         # rank 0 has printed the run's line: whatever goes wrong afterwards must not add another
        self.wake_r = None

    def phase(self, name, seconds=None):
        with self.lock:
            self.phase_name, self.t_phase = name, time.time()
            self.deadline = None if seconds is None else self.t_phase + seconds * self.scale

    def note(self, **kv):
        with self.lock:
            self.notes.update(kv)

    def attach(self, comm):
        self.comm = comm

    def done(self):
        self.stop_ev.set()

    def catch_sigterm(self):
        """main thread only.  The launcher answers a dead rank with SIGTERM to the others; a Python handler would wait
        for the main thread to come back from the call it is stuck in -- the wake-up descriptor is written by the C-level
        handler at once and read by this thread."""
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        self.wake_r = r

    def state(self):
        with self.lock:
            st = {"rank": self.rank, "phase": self.phase_name, "seconds_in_phase": round(time.time() - self.t_phase, 1),
                  "deadline_s": None if self.deadline is None else round(self.deadline - self.t_phase, 1)}
            st.update(self.notes)
        comm = self.comm
        if comm is not None:
            box = []
            t = threading.Thread(target=lambda: box.append(comm.state()), daemon=True)      # (a query of a wedged runtime may block too)
            t.start()
            t.join(2.0)
            st["comm"] = box[0] if box else "no answer from the runtime within 2 s"
        return st

    def fail(self, why, code=4):
        """-> does not return: states to stderr / to rank 0, the error line from rank 0, exit"""
        if self.fired:
            return
        self.fired = True
        st = self.state()
        sys.stderr.write("[bench watchdog] rank %d: %s -- %s\n" % (self.rank, why, json.dumps(st)))
        sys.stderr.flush()
        if self.result_emitted:
            # (a rank that hangs or dies in the shutdown: the measurement is complete and printed; stdout keeps its ONE line)
            self.exit_fn(0)
            return
        if self.peers is not None:
            try:
                self.peers.set("wd/%d" % self.rank, json.dumps(st))
            except Exception:                                       # noqa: BLE001
                pass
        if self.rank == 0:
            states = {0: st}
            t_end = time.time() + self.grace
            for r in range(1, self.world):
                got = None
                while self.peers is not None and got is None:
                    try:
                        got = self.peers.get("wd/%d" % r)
                    except Exception:                               # noqa: BLE001
                        got = None
                    if got is not None or time.time() > t_end:
                        break
                    time.sleep(0.1)
                states[r] = json.loads(got) if got else "no state received within %.0f s" % self.grace
            line = {"metric": self.metric, "value": None, "unit": "queries/s", "n_gpus": self.world, "higher_is_better": True,
                    "error": why, "phase": st["phase"], "watchdog": {"per_rank": [states[r] for r in range(self.world)]}}
            line.update(self.extra)
            try:
                self.emit(line)
            except Exception:                                       # noqa: BLE001
                pass
        else:
            time.sleep(self.grace + 2.0)        # (the launcher kills every rank as soon as one exits: let rank 0 print first)
        self.exit_fn(code)

    def run(self):
        while not self.stop_ev.is_set():
            if self.wake_r is not None:
                ready, _, _ = select.select([self.wake_r], [], [], 0.25)
                if ready:
                    sigs = os.read(self.wake_r, 64)
                    if signal.SIGTERM in sigs:
                        self.fail("terminated by the launcher (SIGTERM) in phase '%s': another rank failed or the run was timed out"
                                  % self.phase_name, code=143)
            else:
                self.stop_ev.wait(0.25)
            with self.lock:
                late = self.deadline is not None and time.time() > self.deadline
                name, limit = self.phase_name, (self.deadline or 0) - self.t_phase
            if late:
                self.fail("phase '%s' did not finish within %.0f s" % (name, limit))


class _StorePeers:
    """the ranks' states for rank 0, through the store torch.distributed's rendezvous runs on (served by the launcher /
    rank 0 in a background thread: it answers while the main threads are stuck)"""
    def __init__(self, store):
        self.store = store
    def set(self, key, value):
        self.store.set("cobs_bench/" + key, value)
    def get(self, key):
        if not self.store.check(["cobs_bench/" + key]):
            return None
        return self.store.get("cobs_bench/" + key).decode()


def other_ipc_mode(v):
    return "1" if v == "0" else "0"


def preflight_child(args):
    """one rank of one preflight attempt, in its OWN process (HSA_ENABLE_IPC_MODE_LEGACY is read once, when the ROCm
    runtime initialises: another value needs another process; and a hang in ncclCommInitRank can only be ended by
    killing the process that sits in it).  Prints one JSON line."""
    res = {"ok": False, "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    try:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), None, False,
                              timeout=__import__("datetime").timedelta(seconds=args.preflight_seconds))
        key = "cobs_bench/pf/%s/uid" % args.preflight_child
        dev = args.preflight_device
        torch.cuda.set_device(dev)
        from cobs_amd.distributed import Comm
        if rank == 0:
            store.set(key, Comm.unique_id())
        uid = store.get(key)
        t0 = time.time()
        comm = Comm(uid, rank, world, device=dev)
        res["comm_init_s"] = round(time.time() - t0, 2)
        res["rccl_ranks"] = comm.size
        res.update(comm.preflight(timeout_ms=int(args.preflight_seconds * 400), big_bytes=args.preflight_big_mib << 20))
        comm.close()
        res["ok"] = True
    except BaseException as e:                                      # noqa: BLE001
        res["error"] = "%s: %s" % (type(e).__name__, str(e)[:400])
    _RESULT_STDOUT.write(json.dumps(res) + "\n")
    _RESULT_STDOUT.flush()
    os._exit(0 if res["ok"] else 5)


def run_preflight(args, world, rank, device, wd):
    """Before the index is built: does a communicator of these ranks come up AND move bytes --
    uneven grouped send / receive all-to-all, all-gather, all-reduce, every byte checked, each step under a time limit
    (cobs_gpu_comm_preflight) -- under the HSA_ENABLE_IPC_MODE_LEGACY value of the environment?  If not, once more
    under the other value; if neither works the run falls back to the host transport (torch.distributed / gloo: the
    library's own exchange plan, the bytes through host memory) and says so.  Every attempt is one child process per
    rank; the ranks agree on each attempt's outcome over the (gloo) process group.
    -> (dict for the JSON line, transport "rccl" | "gloo")"""
    info = {"attempts": []}
    first = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for attempt, mode in enumerate((first, other_ipc_mode(first))):
        wd.phase("preflight attempt %d (HSA_ENABLE_IPC_MODE_LEGACY=%s)" % (attempt, mode), args.preflight_seconds + 45)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=mode)
        cmd = [sys.executable, "-m", "cobs_amd.launch", "--preflight-child", "a%d" % attempt, "--preflight-device", str(device),
               "--preflight-seconds", str(args.preflight_seconds), "--preflight-big-mib", str(args.preflight_big_mib)]
        env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
        mine = {"ok": False}
        t0 = time.time()
        try:
            child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            try:
                out, err = child.communicate(timeout=args.preflight_seconds)
                lines = [ln for ln in out.splitlines() if ln.startswith("{")]
                mine = json.loads(lines[-1]) if lines else {"ok": False, "error": "no result line; exit code %d; stderr: %s"
                                                            % (child.returncode, err[-300:])}
            except subprocess.TimeoutExpired:
                child.kill()
                child.communicate()
                mine = {"ok": False, "error": "no answer within %g s (killed): the communicator did not come up or a collective hung"
                                              % args.preflight_seconds}
        except Exception as e:                                      # noqa: BLE001
            mine = {"ok": False, "error": "could not run the preflight child: %r" % (e,)}
        mine["seconds"] = round(time.time() - t0, 2)
        every = [None] * world
        dist.all_gather_object(every, mine)
        ok = all(bool(r and r.get("ok")) for r in every)
        rec = {"HSA_ENABLE_IPC_MODE_LEGACY": mode, "ok": ok, "seconds_max": max(r.get("seconds", 0) for r in every)}
        if ok:
            rec["comm_init_s_max"] = max(r.get("comm_init_s", 0) for r in every)
            for k in ("alltoall_us", "allgather_us", "allreduce_us", "big_alltoall_us"):
                if all(k in r for r in every):
                    rec[k + "_max"] = max(r[k] for r in every)
            if all("big_alltoall_recv_GBps" in r for r in every):
                rec["big_alltoall_recv_GBps_min"] = min(r["big_alltoall_recv_GBps"] for r in every)
                rec["big_alltoall_MiB_per_pair"] = args.preflight_big_mib
        else:
            rec["errors"] = {str(i): r.get("error", "?") for i, r in enumerate(every) if not (r and r.get("ok"))}
        info["attempts"].append(rec)
        if ok:
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = mode     # (this process has not initialised the ROCm runtime yet)
            info["ipc_mode_legacy_used"] = mode
            info["transport"] = "rccl"
            return info, "rccl"
    info["transport"] = "gloo"
    info["fallback"] = ("RCCL did not pass the preflight under either IPC mode: the exchange of this run goes through host memory "
                        "(torch.distributed / gloo executing the library's own exchange plan) -- a slow but true number")
    return info, "gloo"


def launch_plan(gpus, argv, script):
    """`python <script> --gpus N` started without a launcher: the command that runs the N ranks
    (one process per GPU, rendezvous on 127.0.0.1, a free port)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)


if __name__ == "__main__":
    # one rank of one preflight attempt (run_preflight starts it).  stdout carries its ONE JSON line: RCCL prints a
    # version banner with printf, so file descriptor 1 points at stderr and the line goes to a duplicate of the original.
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--preflight-child", required=True)
    ap.add_argument("--preflight-device", type=int, default=0)
    ap.add_argument("--preflight-seconds", type=float, default=75.0)
    ap.add_argument("--preflight-big-mib", type=int, default=8)
    sys.stdout.flush()
    _RESULT_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    preflight_child(ap.parse_args())
