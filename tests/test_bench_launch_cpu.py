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
