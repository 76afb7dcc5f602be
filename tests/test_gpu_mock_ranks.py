"""GPU: the NATIVE multi-GPU path (cobs_amd/csrc/multi.cpp + comm.cpp) with N > 1 ranks on the one GPU of this box.

RCCL refuses two ranks on one device, so until round 4 the native exchange had only ever run over a one-rank
communicator (tests/test_gpu_rccl.py) while the N > 1 arithmetic was covered by a torch.distributed restatement.
tests/mock_rccl/ is a stand-in for librccl for one process whose ranks are threads sharing a GPU (it moves the bytes
through host staging and CHECKS what hardware answers with a hang: every rank in the same collective, every send met
by a receive of the same size); preloaded, it answers the RCCL calls of the SHIPPED libcobs_gpu.so, which runs the
device-list handle of the C ABI --
N worker threads, one communicator, collective searches -- exactly as an N-GPU node does, against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mock_library(gpu_lib):
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "mock_rccl", "build.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lib = os.path.join(ROOT, "cobs_amd", "libmockrccl.so")
    assert os.path.exists(lib)
    return lib


@pytest.mark.parametrize("ranks", [2, 3, 4, 8])
def test_device_list_handle_over_n_ranks_sharing_the_gpu(mock_library, ranks):
    env = dict(os.environ, LD_PRELOAD=mock_library)
    seed = os.environ.get("COBS_FUZZ_SEED", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "run_ranks.py"), str(ranks), seed],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok "), r.stdout[-3000:] + r.stderr[-6000:]
    assert "[mock rccl]" not in r.stderr, r.stderr[-6000:]


@pytest.mark.parametrize("ranks", [2, 4])
def test_device_list_handle_with_distinct_device_ordinals(mock_library, ranks):
    """round 6: the same, with the stand-in virtualising the device ordinals
    (MOCK_RCCL_VIRTUAL_DEVICES: hipGetDeviceCount answers N, hipSetDevice(d) selects the one GPU and remembers d for the
    thread) -- cobs_gpu_multi_open with devices [0, 1, .., N-1] as on an N-GPU node: the range and duplicate checks,
    one communicator and one index handle per ordinal, every hipSetDevice(ix->device) of the pass / exchange / ranking
    code with an ordinal other than 0.  An ordinal beyond the count is refused before any rank meets another."""
    env = dict(os.environ, LD_PRELOAD=mock_library, MOCK_RCCL_VIRTUAL_DEVICES=str(ranks))
    seed = os.environ.get("COBS_FUZZ_SEED", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "run_ranks.py"), str(ranks), seed],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok "), r.stdout[-3000:] + r.stderr[-6000:]
    assert "[mock rccl]" not in r.stderr, r.stderr[-6000:]
    code = ("import sys; sys.path.insert(0, %r); import cobs_amd\n"
            "try:\n    cobs_amd.MultiSearch(%r, [0, %d])\n    print('opened')\n"
            "except cobs_amd.CobsGpuError as e:\n    print('refused', e.status, e)\n") % (
                ROOT, os.path.join(ROOT, "tests", "golden", "c1.cobs_compact"), ranks)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "refused 7" in r.stdout and "out of range" in r.stdout, r.stdout + r.stderr[-2000:]


@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_batch_exchange_entry_points_over_n_ranks(mock_library, ranks):
    """cobs_gpu_batch_exchange_counts (all-gather / all-to-all to query owners / all-reduce), _exchange_hits,
    _exchange_hits_owned, _exchange_topk -- the calls bench.py's sharded flow and a torch.distributed launcher make,
    one rank per process there -- with N ranks as threads of one process over the stand-in communicator; every rank's
    view afterwards equals the oracle's"""
    env = dict(os.environ, LD_PRELOAD=mock_library, MOCK_RCCL_TIMEOUT_S="30")
    seed = os.environ.get("COBS_FUZZ_SEED", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "run_batch_ranks.py"), str(ranks), seed],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok "), r.stdout[-3000:] + r.stderr[-8000:]
    assert "[mock rccl]" not in r.stderr, r.stderr[-6000:]


def test_cli_device_list_over_three_ranks(mock_library, oracle, golden_dir, tmp_path):
    """the SHIPPED binaries -- cobs_amd/cobs_gpu_query and cobs_amd/libcobs_gpu.so as built, nothing re-linked -- with
    the stand-in's symbols preloaded in place of librccl's: `-d 0,0,0` is cobs_gpu::ShardedClassicSearch over three
    ranks on the one GPU (C++ class -> cobs_gpu_multi_* -> comm.cpp), its output that of the oracle"""
    tool = os.path.join(ROOT, "cobs_amd", "cobs_gpu_query")
    mock = mock_library
    a = os.path.join(golden_dir, "c1.cobs_compact")
    b = os.path.join(golden_dir, "c1.cobs_classic")
    q50 = "AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"
    qf = tmp_path / "q.fa"
    qf.write_text(">first query\n%s\n%s\n\n;second\n%s\n" % (q50[:25], q50[25:], q50[3:40]))
    ixs = [oracle.Index.open(a), oracle.Index.open(b)]
    env = dict(os.environ, LD_PRELOAD=mock, MOCK_RCCL_TIMEOUT_S="30")     # (the stand-in's marker symbol lets ranks share the device)
    for extra in (["-t", "0.05"], ["-t", "0"], ["-t", "0", "-l", "3"]):
        r = subprocess.run([tool, "-d", "0,0,0", "-i", a, "-i", b, "-f", str(qf)] + extra, capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert "[mock rccl]" not in r.stderr, r.stderr
        t = float(extra[1])
        lim = int(extra[3]) if len(extra) > 2 else 0
        want = ""
        for comment, q in (("*first query", q50), ("*second", q50[3:40])):
            res = oracle.search(ixs, q.encode(), t, lim)
            want += "%s\t%d\n" % (comment, len(res)) + "".join("%s\t%d\n" % (n, s) for (_, _, n, s) in res)
        assert r.stdout == want, extra
    r = subprocess.run([tool, "-d", "0,0,0", "-i", b, q50.replace("G", "N", 1)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "Invalid DNA base pair" in r.stderr


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_sharded_flow_over_n_ranks(mock_library, ranks):
    """bench.py's ShardedRun -- the overlapped N > 1 flow the driver's scaling run times: sub-batches, hashing / scan /
    exchange on three streams tied by events, the native all-to-all to query owners -- with N ranks as threads; every
    rank's assembled rows equal the oracle's, for 1, 2 and 3 sub-batches"""
    env = dict(os.environ, LD_PRELOAD=mock_library, MOCK_RCCL_TIMEOUT_S="30")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "run_bench_ranks.py"), str(ranks)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok "), r.stdout[-3000:] + r.stderr[-8000:]
    assert "[mock rccl]" not in r.stderr, r.stderr[-6000:]


# ---- round 5: faults between the ranks (the one-shot 8-GPU run must not come back empty) ----------
def _fault(mock_library, case, ranks, timeout_s="8"):
    env = dict(os.environ, LD_PRELOAD=mock_library, MOCK_RCCL_TIMEOUT_S=timeout_s)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "run_fault_ranks.py"), case, str(ranks)],
                          capture_output=True, text=True, timeout=900, env=env)


@pytest.mark.parametrize("ranks", [2, 3])
def test_fault_a_failing_send_inside_a_group(mock_library, ranks):
    """an ncclSend that returns an error inside ncclGroupStart .. ncclGroupEnd: the group is still closed on that rank
    (comm.cpp: GroupScope), the dead communicator is aborted, the peers fail with it instead of waiting for the time
    limit, every later call fails at once and says why"""
    r = _fault(mock_library, "send_error", ranks)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok send_error"), r.stdout[-3000:] + r.stderr[-8000:]
    assert "never entered the collective" not in r.stderr, r.stderr[-6000:]      # nobody had to be released by the time limit
    assert "[mock rccl fault] rank 1: injected failure of ncclSend" in r.stderr


@pytest.mark.parametrize("ranks", [2, 3])
def test_fault_a_size_mismatch_between_send_and_receive(mock_library, ranks):
    """one byte less sent than the peer expects: both ends come back with ERR_RCCL (on hardware: a hang or silent
    corruption), no group stays open, a third rank that enters the next collective alone is released by the time limit"""
    r = _fault(mock_library, "size_mismatch", ranks)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok size_mismatch"), r.stdout[-3000:] + r.stderr[-8000:]
    assert "[mock rccl fault] rank 0: injected short ncclSend" in r.stderr


@pytest.mark.parametrize("ranks", [2, 5])
def test_preflight_passes_when_healthy_and_fails_bounded_when_not(mock_library, ranks):
    """cobs_gpu_comm_preflight -- what bench.py --gpus N runs before it builds the index: uneven all-to-all, all-gather,
    all-reduce, every byte checked -- over N ranks; with a short first message it fails on the ranks involved, in time"""
    r = _fault(mock_library, "preflight", ranks)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok preflight"), r.stdout[-3000:] + r.stderr[-8000:]


def test_fault_a_rank_that_never_enters_the_exchange_ends_as_an_error_line(mock_library):
    """bench.py's overlapped sharded flow with three ranks; the last one stops stepping.  The others wait in the
    exchange (the stand-in would wait 60 s here, hardware for ever): bench.Watchdog ends the run after the phase's 6 s --
    ONE JSON line with "error" and every rank's state (phase, step, what its communicator entered last) on stdout, the
    same states on stderr, exit code 4"""
    import json
    r = _fault(mock_library, "watchdog", 3, timeout_s="60")
    assert r.returncode == 4, (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 3 and "did not finish within" in line["error"]
    per = line["watchdog"]["per_rank"]
    assert [p["rank"] for p in per] == [0, 1, 2]
    assert per[2]["step"] == 2 and "INSIDE a call" not in per[2]["comm"]            # the rank that stopped: outside RCCL
    assert any("INSIDE a call" in p["comm"] or "stream busy" in p["comm"] for p in per[:2]), per       # its peers: waiting for it
    assert r.stderr.count("[bench watchdog] rank") == 3
