import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
