import os
import sys

# CPU oracles run PyTorch's OpenMP GRU.  On a box whose cores are busy with something else (other jobs in the container, a
# second pytest) OpenMP workers that SPIN at every barrier of a 10 000-step loop turn a 30-second suite into tens of
# minutes (seen by the round-3 review: > 25 min at 600 % CPU; reproduced here next to three training jobs).  Idle
# workers sleep instead, and the CPU-only suite uses at most four threads -- its cases are small.  (Set before torch
# is imported; explicit settings in the environment win.)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


# The GPU tests written before the split scan existed pin down the SEQUENTIAL machinery (time-slab host path,
# overlapped projection, multi-pass batches, tile sizes ...) with bit-for-bit comparisons across shapes and batch
# compositions; a split call agrees with a sequential one to ~1e-7, not bit for bit.  They keep running on the
# sequential scan; tests/test_scan_split_gpu.py covers the product default (MDK_SCAN_SPLIT unset = auto).
os.environ.setdefault("MDK_SCAN_SPLIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from medaka_amd import lib
        return lib.device_count() > 0
    except Exception:
        return False


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    CPUs and grants 16: 256 torch threads on a 16-core quota made one CPU-oracle pass take > 20 minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pytest_collection_modifyitems(config, items):
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:            # no test may hang a GPU box: the slowest one takes ~40 s
            if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(420))
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
    import torch
    n = usable_cores()                         # CPU oracles: never more threads than the cgroup grants ...
    if not _gpu_available():
        n = min(n, 4)                          # ... and no more than the small cases of the CPU-only suite can use
    torch.set_num_threads(n)


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


@pytest.fixture
def debug_hooks(request):
    """Tests that need the test hooks of the DEBUG library (include/medaka_amd.h, MDK_DEBUG_HOOKS block: the release library
    does not carry them).  Inside a process that has the debug library loaded: True.  Otherwise the test is re-executed in a
    child process against medaka_amd/libmedaka_amd_debug.so, must pass there, and the fixture returns False (the caller
    returns at once)."""
    import subprocess
    import sys
    from medaka_amd import build, lib
    if lib.is_debug_library():
        return True
    assert os.path.exists(build.LIB_DEBUG), f"{build.LIB_DEBUG} not built (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, MDK_LIB=build.LIB_DEBUG, MDK_SKIP_BUILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-p", "no:cacheprovider", request.node.nodeid],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return False
