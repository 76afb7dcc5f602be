import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _native_code_is_built():
    """A fresh checkout has no binaries (they are git-ignored): build libcobs_gpu.so with hipcc
    (cross-compiles for gfx950 without a GPU) before the first test.  A tree that already holds
    the library -- e.g. the snapshot on the GPU box -- is left alone."""
    import subprocess
    lib = os.path.join(ROOT, "cobs_amd", "libcobs_gpu.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cobs_amd", "csrc"), "-j8"],
                              stdout=subprocess.DEVNULL)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """the CPU restatement (checker); building it is part of the test session"""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def construct():
    from oracle import construct as K
    return K


@pytest.fixture(scope="session")
def gpu_lib():
    """libcobs_gpu.so with a device; GPU tests fail (not skip) if it is missing"""
    import torch  # noqa: F401  (loads the HIP runtime the library binds to)
    import cobs_amd
    from cobs_amd import _capi
    lib = _capi.load()
    assert lib.cobs_gpu_device_count() > 0, "no HIP device visible"
    return cobs_amd


@pytest.fixture(scope="module")
def comm_one_rank(gpu_lib):
    """a one-rank RCCL communicator (the real library: what the multi-GPU entry points run over on a one-GPU box)"""
    from cobs_amd.distributed import Comm
    c = Comm(Comm.unique_id(), 0, 1, device=0)
    yield c
    c.close()
