"""Long fed loop (bench.py::fed_loop with the engine's collate: every batch handed over early, every forward started ahead of its
call): rate over thousands of batches, device memory and host RSS before / after -- nothing may grow.
   python profiles/r6_experiments/soak_fed_loop.py [--batches N] [--half]"""
import argparse
import os
import resource
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

import bench  # noqa: E402
from medaka_amd import models, torch_ext  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=1500)
ap.add_argument("--half", action="store_true")
args = ap.parse_args()
torch.set_num_threads(bench.usable_cores())
state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
model = models.GRUModel()
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model = model.to("cuda").eval()
if args.half:
    model.half()
windows = bench.loop_windows(10000, 50, 4321)
fast = lambda data: torch_ext.Batch.collate(data)
bench.fed_loop(model, windows, 200, 20, fast, warm=2)


def mem():
    free, total = torch.cuda.mem_get_info()
    return round((total - free) / 2**30, 3), round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 3)


m0 = mem()
rates = []
for part in range(3):
    r = bench.fed_loop(model, windows, 200, args.batches // 3, fast, warm=2)
    rates.append(round(r["value"] / 1e6, 1))
    print(f"part {part}: {rates[-1]} M columns/s, predict {r['predict_ms_median']:.2f} ms, started ahead {r['forwards_started_ahead']} of {r['timed_batches']}; "
          f"device GiB / host max RSS GiB: {mem()}", flush=True)
m1 = mem()
sp = model.engine().split()
print(f"soak: {args.batches} batches, half={args.half}: rates {rates}; device memory {m0[0]} -> {m1[0]} GiB, host max RSS {m0[1]} -> {m1[1]} GiB; "
      f"split {sp['status']} margin {sp['margin']} fallbacks {sp['fallbacks']} audits {sp['audits']} (worst {sp['audit_worst_dp']:.2e}) probes {sp['probes']}")
assert m1[0] - m0[0] < 0.3, "device memory grew"
