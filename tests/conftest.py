import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from medaka_amd import lib
        return lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The engine must be built before any test: no silent fallback."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def gold():
    d = {}
    for name in ("gru_inputs", "gru_outputs", "majority_outputs", "consensus_decode",
                 "weights_init", "weights_trained"):
        d[name] = dict(np.load(os.path.join(GOLD, name + ".npz")))
    return d


def weight_set(gold, name):
    if name == "init":
        return gold["weights_init"]
    if name == "x3":
        return {k: v * np.float32(3.0) for k, v in gold["weights_init"].items()}
    if name == "trained":
        return gold["weights_trained"]
    raise KeyError(name)
