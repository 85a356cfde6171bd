"""Fed loop (bench.py::fed_loop: loader threads -> engine collate -> predict_on_batch -> writer) with and without the early
start of the next batch's forward (include/medaka_amd.h `mdk_gru_forward_pipelined`), split and sequential scans, both
precisions.   python profiles/r6_experiments/early_start_probe.py [--batches N] [--batch B]
Environment knobs worth sweeping: GPU_MAX_HW_QUEUES (HIP's hardware queues per process, default 4: streams that share one are
serialised), MDK_EARLY_START, MDK_SCAN_SPLIT."""
import argparse
import json
import os
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
ap.add_argument("--batches", type=int, default=24)
ap.add_argument("--batch", type=int, default=200)
ap.add_argument("--modes", default="split,sequential")
ap.add_argument("--precisions", default="fp32,half")
ap.add_argument("--early", default="0,1")
args = ap.parse_args()
B, T = args.batch, 10000
dev = torch.device("cuda", 0)
state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
windows = bench.loop_windows(T, 50, 4321)
fast = lambda data: torch_ext.Batch.collate(data)
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "batch": B}
for prec in args.precisions.split(","):
    model = models.GRUModel()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    if prec == "half":
        model.half()
    eng = model.engine()
    for mode in args.modes.split(","):
        eng.set_option("scan_split", 0 if mode == "sequential" else 1)
        for early in [int(v) for v in args.early.split(",")]:
            eng.set_option("early_start", early)
            bench.fed_loop(model, windows, B, 4, fast, warm=1)
            r = bench.fed_loop(model, windows, B, args.batches, fast)
            key = f"{prec}/{mode}/early_start={early}"
            out[key] = {"M_columns_per_s": round(r["value"] / 1e6, 1), "ms_per_batch": round(r["ms_per_batch"], 3),
                        "predict_ms_median": round(r["predict_ms_median"], 3), "collate_ms_median": round(r.get("collate_ms_median", 0), 3),
                        "wait_for_batch_ms_median": round(r.get("main_thread_wait_for_batch_ms_median", 0), 3)}
            print(key, out[key], flush=True)
    eng.close()
print(json.dumps(out))
