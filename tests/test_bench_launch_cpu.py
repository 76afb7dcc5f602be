"""CPU: `bench.py --gpus N` decides by itself how the N ranks come to exist (no GPU needed for the plan)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=300, env=e)


def test_plain_start_with_gpus_2_plans_its_own_launch():
    r = _run(["--gpus", "2", "--steps", "3", "--dry-run-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout)
    assert plan["gpus"] == 2 and plan["ranks"] == 2 and plan["launched_by"] == "self"
    cmd = plan["command"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "2", "--steps", "3"]          # the ranks get the same arguments, minus the dry run


def test_under_a_launcher_no_second_launch():
    r = _run(["--gpus", "2", "--dry-run-launch"], {"WORLD_SIZE": "2", "RANK": "0"})
    plan = json.loads(r.stdout)
    assert plan["command"] is None and plan["launched_by"].startswith("torch.distributed.run")
    r = _run(["--gpus", "1", "--dry-run-launch"])
    assert json.loads(r.stdout)["command"] is None


def test_gpus_flag_and_world_size_must_agree():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr


# ---- round 5: the N-rank run cannot come back empty -------------------------------------------------------------------
def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Exit(Exception):
    pass


def test_watchdog_turns_an_overrun_phase_into_an_error_line():
    """three ranks as three watchdogs sharing a dict: rank 2 'hangs' in its phase; every rank reports its state, rank 0
    emits ONE line with "error" and the three states, every rank leaves with a non-zero code"""
    import threading
    import time
    bench = _load_bench()

    class Peers:
        def __init__(self):
            self.d, self.lock = {}, threading.Lock()
        def set(self, k, v):
            with self.lock:
                self.d[k] = v
        def get(self, k):
            with self.lock:
                return self.d.get(k)
    peers, lines, codes = Peers(), [], {}

    class FakeComm:
        def __init__(self, r):
            self.r = r
        def state(self):
            return "rank %d/3 device 0: rccl calls entered 7 returned 6 (INSIDE a call); last: ncclGroupEnd(); stream busy; communicator ok" % self.r
    dogs = []
    for r in range(3):
        def leave(code, r=r):
            codes[r] = code
            raise _Exit()
        wd = bench.Watchdog(r, 3, lines.append, peers=peers, exit_fn=leave, grace=1.0)
        wd.attach(FakeComm(r))
        wd.phase("warm-up and timed steps", 0.5)
        wd.note(step=4)
        dogs.append(wd)

    def body(wd):
        try:
            wd.run()
        except _Exit:
            pass
    ts = [threading.Thread(target=body, args=(wd,)) for wd in dogs]
    t0 = time.time()
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert time.time() - t0 < 20 and not any(t.is_alive() for t in ts)
    assert codes == {0: 4, 1: 4, 2: 4}
    assert len(lines) == 1
    line = lines[0]
    assert line["value"] is None and line["n_gpus"] == 3 and "did not finish within" in line["error"]
    per = line["watchdog"]["per_rank"]
    assert [p["rank"] for p in per] == [0, 1, 2] and all(p["phase"] == "warm-up and timed steps" and p["step"] == 4 for p in per)
    assert all("INSIDE a call" in p["comm"] for p in per)
    json.dumps(line)                    # serialisable as it stands


def test_watchdog_is_silent_when_phases_finish():
    import time
    bench = _load_bench()
    lines = []
    wd = bench.Watchdog(0, 1, lines.append, exit_fn=lambda c: (_ for _ in ()).throw(_Exit()))
    wd.start()
    wd.phase("a", 5)
    time.sleep(0.6)
    wd.phase("b", None)
    time.sleep(0.6)
    wd.done()
    wd.join(5)
    assert not wd.is_alive() and lines == []


PREFLIGHT_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import importlib.util
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(%r, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch.distributed as dist
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
class Args: pass
a = Args(); a.preflight_seconds = 60.0; a.preflight_big_mib = 0
wd = bench.Watchdog(rank, world, lambda l: None)
info, transport = bench.run_preflight(a, world, rank, 0, wd)
if rank == 0:
    print("RESULT " + json.dumps({"info": info, "transport": transport, "env": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
dist.barrier()
dist.destroy_process_group()
"""


def test_preflight_without_a_gpu_falls_back_and_says_why(tmp_path):
    """world_size 2, gloo, no GPU: both preflight attempts (one child process per rank and IPC mode) fail in
    cobs_gpu_comm_create -- 'no HIP device' -- the ranks agree on that, the other HSA_ENABLE_IPC_MODE_LEGACY value is tried,
    and the verdict is the host transport, with every attempt and every rank's error in the record"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this is the no-GPU case")
    script = tmp_path / "pf_worker.py"
    script.write_text(PREFLIGHT_WORKER % (ROOT, ROOT))
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["transport"] == "gloo" and res["info"]["transport"] == "gloo" and "fallback" in res["info"]
    att = res["info"]["attempts"]
    assert [a["HSA_ENABLE_IPC_MODE_LEGACY"] for a in att] == ["0", "1"] and not any(a["ok"] for a in att)
    assert all(set(a["errors"]) == {"0", "1"} for a in att)
    assert all("No HIP GPUs" in v or "no HIP device" in v or "NO_DEVICE" in v for a in att for v in a["errors"].values()), att
    assert res["env"] == "0"            # nothing worked: the environment is left as it was
