import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def sgo():
    """the CPU oracle (oracle/sgo.py) -- checker only"""
    from oracle import sgo as _sgo
    _sgo.lib()
    return _sgo


@pytest.fixture(scope="session")
def sg():
    """the product package (ctypes mirror of the C-ABI); builds nothing"""
    from __graft_entry__ import load_package
    return load_package()
