import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gold_small():
    return dict(np.load(os.path.join(GOLDEN, "arae_small.npz")))


@pytest.fixture(scope="session")
def gold_eos():
    return dict(np.load(os.path.join(GOLDEN, "arae_eos.npz")))


@pytest.fixture(scope="session")
def gold_full():
    p = os.path.join(GOLDEN, "arae_full_T4000.npz")
    if not os.path.exists(p):
        pytest.skip("full-size golden not generated")
    return dict(np.load(p))


@pytest.fixture(scope="session")
def gold_batch():
    p = os.path.join(GOLDEN, "arae_batch.npz")
    if not os.path.exists(p):
        pytest.skip("batched-path golden not generated")
    return dict(np.load(p))


@pytest.fixture(scope="session")
def gold_long24():
    p = os.path.join(GOLDEN, "arae_long24.npz")
    if not os.path.exists(p):
        pytest.skip("full-depth long-context golden not generated (oracle/make_golden.py long24)")
    return dict(np.load(p))


@pytest.fixture(scope="session")
def gold_long():
    p = os.path.join(GOLDEN, "arae_long.npz")
    if not os.path.exists(p):
        pytest.skip("long-context golden not generated")
    return dict(np.load(p))
