import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def board_hashes(rows):
    """FNV-1a over the 20 rows of every board (the per-step fingerprint stored in the F3 fixtures)."""
    rows = np.asarray(rows).astype(np.uint64)
    h = np.full(rows.shape[0], 0xcbf29ce484222325, dtype=np.uint64)
    prime = np.uint64(0x100000001b3)
    with np.errstate(over="ignore"):
        for r in range(20):
            h = (h ^ rows[:, r]) * prime
    return h
