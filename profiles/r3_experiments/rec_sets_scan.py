"""Per-layer recurrence time of the one- and two-set kernels over the batch size (device-resident, hipEvent spans)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from medaka_amd import engine, synth  # noqa: E402

st = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
e = engine.GruEngine(st)
e.enable_timing(True)
e.set_option("max_rows_per_pass", int(os.environ.get("MAX_ROWS", "0")))
T = 10000
base = synth.counts_windows(8, T, seed=3)
for B in [int(a) for a in sys.argv[1:]] or [504, 1000, 1496, 2000]:
    x = torch.from_numpy(np.concatenate([base] * (B // 8))).cuda()
    y = torch.empty((B, T, 5), dtype=torch.float32, device="cuda")
    for sets in [int(v) for v in os.environ.get("SETS", "1,2").split(",")]:
        e.set_option("rec_sets", sets)
        e.set_option("overlap_gemm", 0)
        for _ in range(2):
            e.forward_ptr(x.data_ptr(), B, T, y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        t = e.timing()
        tiles = B // 8
        wgs = tiles * 2 if sets == 1 else (tiles + 1) // 2 * 2
        print(f"B={B:5d} sets={sets} work-groups {wgs:4d}: layer 0 {t['rec_ms'][0]:7.2f} ms  layer 1 {t['rec_ms'][1]:7.2f} ms  "
              f"gemm {sum(t['gi_ms']):6.2f}  total {t['total_ms']:7.2f} ms   us/step L0 {t['rec_ms'][0] * 1e3 / T:.3f} L1 {t['rec_ms'][1] * 1e3 / T:.3f}", flush=True)
    del x, y
