import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hostmath():
    """Host instantiation (double/float) of the kernels' per-profile math -- test harness only."""
    import ctypes
    src = os.path.join(ROOT, "tests", "hostmath", "hostmath.cpp")
    so = os.path.join(ROOT, "tests", "hostmath", "libhostmath.so")
    csrc = os.path.join(ROOT, "gigalens_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h") and not f.endswith(".hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture
def rng():
    return np.random.default_rng(0)
