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
    env = dict(os.environ, LD_PRELOAD=mock, COBS_GPU_TEST_RANKS_SHARE_A_DEVICE="1", MOCK_RCCL_TIMEOUT_S="30")
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
