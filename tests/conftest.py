import ctypes as C
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    # (pytest-timeout registers this mark itself when it is installed; registered here so that the mark is known without it)
    config.addinivalue_line("markers", "timeout(seconds): per-test time limit, enforced by pytest-timeout when installed")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("rust-raytracer_amd")


@pytest.fixture(scope="session")
def oracle(pkg):
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("oracle_capi.cpp", "caustic_oracle.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return pkg.bindings.Backend(C.CDLL(so), "orc_", "oracle", sum_dtype="float64")


@pytest.fixture(scope="session")
def gpu_lib(pkg):
    """libcaustic_hip.so loaded (host-side entry points work without a GPU)."""
    return pkg.load_gpu()


@pytest.fixture(scope="session")
def gpu(pkg, gpu_lib):
    """The GPU backend; fails loudly when there is no device."""
    n = pkg.gpu_device_count()
    assert n >= 1, "no HIP device visible"
    return gpu_lib
