import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """The CPU oracle (PyTorch-CPU) must not oversubscribe: a GPU box gives one job a 16-CPU share of a much larger host, and torch
    would start one thread per HOST core (the twin expected-gradient tests took 10 minutes that way instead of one)."""
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, n)))


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a device; skip them (never silently pass) when there is none."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            path = os.path.join(GOLDEN, name)
            if name.endswith(".json"):
                with open(path) as f:
                    cache[name] = json.load(f)
            else:
                cache[name] = dict(np.load(path))
        return cache[name]
    return load


def rel_err(a, b):
    """max|a-b| / max(|b|max, tiny): norm-wise relative error used throughout the parity tests."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
