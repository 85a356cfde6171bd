"""Which activity advances whatever ends the slow stretch after exactly 40 host-path calls of a process?
PROBE_MODE = A: 60 device-resident forwards first; B: 45 host calls with stream_host = 0 (one linear copy out) first;
C: 45 host calls of the sequential scan first (scan_split = 0); D: a second model object after the first has settled;
E: 45 calls through the engine's pointer entry with ONE fixed page-locked result buffer first."""
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

mode = os.environ.get("PROBE_MODE", "A")
B, T = 200, 10000
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))


def make():
    m = models.GRUModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.to(dev).eval()


m = make()
x = synth.counts_windows(40, T, depth=50, seed=1234)
x = np.concatenate([x] * 5)[:B]
x_dev = torch.from_numpy(x).to(dev)
xb = Batch(counts_matrix=torch.from_numpy(x).pin_memory())
hold = {}


def host_calls(model, n, label):
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        hold["p"] = model.predict_on_batch(xb)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"mode {mode} {label}: " + " ".join(f"{t:.1f}" for t in ts), flush=True)


eng = m.engine()
if mode == "A":
    for i in range(60):
        with torch.inference_mode():
            hold["y"] = m.forward(x_dev)
    torch.cuda.synchronize()
    host_calls(m, 50, "after 60 device-resident forwards")
elif mode == "B":
    eng.set_option("stream_host", 0)
    host_calls(m, 45, "stream_host=0")
    eng.set_option("stream_host", 1)
    host_calls(m, 50, "then stream_host=1")
elif mode == "C":
    eng.set_option("scan_split", 0)
    host_calls(m, 45, "scan_split=0")
    eng.set_option("scan_split", 1)
    host_calls(m, 50, "then scan_split=1")
elif mode == "D":
    host_calls(m, 50, "first model")
    m2 = make()
    host_calls(m2, 50, "second model object")
elif mode == "E":
    out = torch.empty((B, T, 5), dtype=torch.float32).pin_memory()
    xp = xb.counts_matrix
    ts = []
    for i in range(45):
        t0 = time.perf_counter()
        eng.forward_ptr(xp.data_ptr(), B, T, out.data_ptr(), host=True)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"mode E forward_ptr into one fixed buffer: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
    host_calls(m, 50, "then predict_on_batch")
elif mode in ("F", "G"):
    # the mechanism by itself: settle, then allocate and FREE device memory (F: 12 GB, G: 3 GB) and go on calling
    eng.set_option("scan_split_audit", 0)
    host_calls(m, 12, "settled (audit off)")
    n = (12 if mode == "F" else 3) << 30
    t = torch.empty(n, dtype=torch.uint8, device=dev)
    t.fill_(1)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()                 # hipFree
    host_calls(m, 60, f"after hipFree of {n >> 30} GB")
elif mode == "H":
    host_calls(m, 60, "default (audit on, lean)")
