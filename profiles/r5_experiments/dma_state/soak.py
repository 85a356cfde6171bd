"""600 host-path calls of one engine: the standing audit (every 256th certified call) must neither leave slow calls behind it
nor grow the footprint."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import models, synth  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402

B, T = 200, 10000
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
m = models.GRUModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
m = m.to(dev).eval()
x = synth.counts_windows(40, T, depth=50, seed=1234)
x = np.concatenate([x] * 5)[:B]
xb = Batch(counts_matrix=torch.from_numpy(x).pin_memory())
eng = m.engine()
hold, ts, audited, free = {}, [], [], []
for i in range(600):
    t0 = time.perf_counter()
    hold["p"] = m.predict_on_batch(xb)
    ts.append(round(1e3 * (time.perf_counter() - t0), 3))
    if eng.split()["audited"]:
        audited.append(i)
    if i in (3, 599):
        free.append(torch.cuda.mem_get_info()[0])
s = eng.split()
around = {a: ts[a:a + 6] for a in audited}
steady = sorted(t for i, t in enumerate(ts) if i > 10 and i not in audited)
rep = {"audited_calls": audited, "ms_of_and_behind_each_audited_call": around, "median_ms": steady[len(steady) // 2],
       "p99_ms": steady[int(0.99 * len(steady))], "calls_over_9.5_ms_not_audited": [i for i, t in enumerate(ts) if t > 9.5 and i not in audited],
       "device_memory_in_use_grew_by_MB": round((free[0] - free[1]) / 2**20, 1), "audits": s["audits"], "audit_failures": s["audit_failures"],
       "audit_worst_dp": s["audit_worst_dp"]}
print(json.dumps(rep))
out = os.path.join(ROOT, "gpurun_out", "r5_dma_state")
os.makedirs(out, exist_ok=True)
json.dump(rep, open(os.path.join(out, "soak.json"), "w"), indent=1)
