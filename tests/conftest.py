import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gdyn():
    return importlib.import_module("2022a-genome-dynamics_amd")


@pytest.fixture(scope="session")
def oracle(gdyn):
    """The CPU fp64 oracle bound through the same ctypes ABI (test infrastructure)."""
    path = os.environ.get("GDYN_ORACLE_LIB", os.path.join(ROOT, "oracle", "liboracle.so"))   # e.g. a sanitizer build
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = gdyn.Lib(path)
    assert lib.backend == "oracle"
    return lib


@pytest.fixture(scope="session")
def hip(gdyn):
    """The product library; GPU tests fail loudly if it is not built.  (GDYN_TEST_LIB: a developer build of csrc/ instead -- the
    test harness reads it, the package does not.)"""
    return gdyn.load(os.environ.get("GDYN_TEST_LIB") or None)
