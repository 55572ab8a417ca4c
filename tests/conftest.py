"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` runs in the build container (no GPU): oracle vs golden vectors,
host logic, C-ABI symbol check, gloo world_size-2 tests.
``-m gpu`` runs on an MI355X: HIP-vs-oracle parity through the C-ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    """Load one golden-vector file written by tests/golden/make_golden.py."""
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def load_golden():
    return golden
