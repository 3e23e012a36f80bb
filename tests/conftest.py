import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle_ctypes
    oracle_ctypes.lib()
    return oracle_ctypes


@pytest.fixture(scope="session")
def rt():
    """The shipped ctypes binding; the library must already be built."""
    return importlib.import_module("racer-tracer_amd")


@pytest.fixture(scope="session")
def abi():
    return importlib.import_module("racer-tracer_amd.abi")


@pytest.fixture(scope="session")
def gpu(rt):
    """Skips nothing: a -m gpu run without a device is a failure, not a skip."""
    n = rt.device_count()
    assert n > 0, "pytest -m gpu needs a visible HIP device (rt_device_count() == 0)"
    return n
