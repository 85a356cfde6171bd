"""Why is a host-path call 1 ms slower with page-locked buffers from mdk_host_alloc (hipHostMalloc, exact size) than with
torch's page-locked tensors (8.99 against 8.0 ms at 200 x 10 000)?  The same call with result / input buffers allocated in
different ways."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import engine, synth  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
B, T = 200, 10000
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
e = engine.GruEngine(w)
e.set_option("scan_split_adapt", 0)
x = synth.counts_windows(40, T, depth=50, seed=1234)
x = np.concatenate([x] * 5)[:B]
NX, NP = x.nbytes, B * T * 5 * 4


def hmalloc(n, flags=0):
    p = ctypes.c_void_p()
    rc = hip.hipHostMalloc(ctypes.byref(p), n, flags)
    assert rc == 0, rc
    return p.value


def time_calls(label, xp, pp, n=14):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        e.forward_ptr(xp, B, T, pp, host=True)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{label:64s} x@{xp % (2 << 20):#9x} p@{pp % (2 << 20):#9x}  median of last 8: {sorted(ts[-8:])[4]:.2f} ms", flush=True)


tx = torch.from_numpy(x).pin_memory()
tp = torch.empty((B, T, 5), dtype=torch.float32, pin_memory=True)
time_calls("torch in, torch out (warm-up: audit etc.)", tx.data_ptr(), tp.data_ptr(), 20)
time_calls("torch in, torch out", tx.data_ptr(), tp.data_ptr())


def filled(ptr):
    ctypes.memmove(ptr, x.ctypes.data, NX)
    return ptr


up = lambda n, a: (n + a - 1) // a * a
variants = [
    ("hipHostMalloc exact size, default flags (mdk_host_alloc)", lambda n: hmalloc(n)),
    ("hipHostMalloc rounded up to 2 MB", lambda n: hmalloc(up(n, 2 << 20))),
    ("hipHostMalloc rounded up to a power of two", lambda n: hmalloc(1 << (n - 1).bit_length())),
    ("hipHostMalloc exact, hipHostMallocNonCoherent (0x80000000)", lambda n: hmalloc(n, 0x80000000)),
    ("hipHostMalloc exact, hipHostMallocCoherent (0x40000000)", lambda n: hmalloc(n, 0x40000000)),
    ("hipHostMalloc exact, hipHostMallocPortable|Mapped (0x3)", lambda n: hmalloc(n, 0x3)),
    ("hipHostMalloc exact, hipHostMallocNumaUser (0x20000000)", lambda n: hmalloc(n, 0x20000000)),
    ("hipHostMalloc + 2 MB, pointer aligned up to 2 MB", lambda n: up(hmalloc(n + (2 << 20)), 2 << 20)),
]
for label, alloc in variants:
    try:
        time_calls("out: " + label, tx.data_ptr(), alloc(NP))
    except AssertionError as exc:
        print(f"out: {label}: hipHostMalloc failed ({exc})")
for label, alloc in variants[:4] + variants[-1:]:
    try:
        time_calls("in : " + label, filled(alloc(NX)), tp.data_ptr())
    except AssertionError as exc:
        print(f"in : {label}: hipHostMalloc failed ({exc})")
time_calls("torch in, torch out (again)", tx.data_ptr(), tp.data_ptr())
