"""Workload for a rocprofv3 kernel + memory-copy trace of the HALF-precision host path (what `medaka inference` runs on a GPU
by default, reference prediction.py:164-168), 200 x 10000:
  (a) `n_cold` host-to-host calls on the same page-locked batch, wall time of each (who keeps calls 3..100 slow?);
  (b) the fed loop (bench.py::fed_loop: loader threads -> Batcher/collate -> predict_on_batch -> writer), wall time of
      every predict_on_batch and collate.
   python profiles/host_trace_half.py [--fp32] [--cold N] [--loop N] [--stage 0|1]
profiles/timeline.py prints the kernels and copies of chosen calls from the trace."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

import bench  # noqa: E402
from medaka_amd import models, synth, torch_ext  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--cold", type=int, default=120)
ap.add_argument("--loop", type=int, default=16)
ap.add_argument("--batch", type=int, default=200)
ap.add_argument("--device-first", type=int, default=0, help="device-resident forwards before the cold calls (what bench.py does)")
args = ap.parse_args()

B, T = args.batch, 10000
dev = torch.device("cuda", 0)
state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
model = models.GRUModel()
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model = model.to(dev).eval()
if not args.fp32:
    model.half()
x = np.concatenate([synth.counts_windows(8, T, depth=50, seed=s) for s in range(-(-B // 8))])[:B]
eng = model.engine()
out = {"half": not args.fp32, "batch": B}
if args.device_first:
    xd = torch.from_numpy(x).to(dev)
    for _ in range(args.device_first):
        with torch.inference_mode():
            model.forward(xd)
    torch.cuda.synchronize(dev)
    out["split_after_device_calls"] = eng.split()
xb = Batch(counts_matrix=torch.from_numpy(x).pin_memory())
cold = []
for i in range(args.cold):
    t0 = time.perf_counter()
    model.predict_on_batch(xb)
    cold.append(round(1e3 * (time.perf_counter() - t0), 3))
out["cold_calls_ms"] = cold
out["split_after_cold_calls"] = eng.split()
print("cold calls (ms):", " ".join(f"{c:.2f}" for c in cold), flush=True)
if args.loop:
    windows = bench.loop_windows(T, 50, 4321)
    fast = lambda data: torch_ext.Batch.collate(data)
    bench.fed_loop(model, windows, B, 3, fast, warm=1)
    r = bench.fed_loop(model, windows, B, args.loop, fast, detail=True)
    out["fed_loop"] = r
    print("fed loop:", json.dumps(r), flush=True)
print(json.dumps(out), flush=True)
