import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gc():
    """The host-side package (ctypes mirror of libgnsscorr.so)."""
    import gnsscorr_loader
    return gnsscorr_loader.load()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def synth(gc):
    import importlib
    return importlib.import_module("erlangnetwork_gnsslib_sdr_amd.synth")


@pytest.fixture()
def engine(gc):
    """A fresh GPU context; fails loudly (no fallback) when no device is visible."""
    e = gc.Engine(0)
    yield e
    e.close()


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
