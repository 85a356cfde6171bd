"""Workload for a rocprofv3 kernel + memory-copy trace of the split scan's host path (profiles/timeline.py reads the
result): page-locked buffers, 200 x 10000, a few calls per configuration.  Wall time of every call on stdout."""
import sys
import time

import numpy as np

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import engine, synth  # noqa: E402

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 200, 10000
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
x = np.concatenate([synth.counts_windows(8, T, depth=50, seed=s) for s in range(-(-B // 8))])[:B]
e = engine.GruEngine(w)
e.set_option("scan_split_audit", 0)
px, pp = engine.PinnedArray(x.shape), engine.PinnedArray((B, T, 5))
px.array[...] = x
for label, opts in (("result in column chunks under the scan's second half", {"stream_host": 1}), ("one copy each way", {"stream_host": 0})):
    for k, v in opts.items():
        e.set_option(k, v)
    for i in range(5):
        t0 = time.perf_counter()
        e.forward_host(px.array, out=pp.array)
        print(f"{label}: call {i}: {1e3 * (time.perf_counter() - t0):.2f} ms  {e.split()['status']}", flush=True)
e.close()
