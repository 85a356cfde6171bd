"""bench.py's set-up in front of the host-to-host calls of h2h_probe.py, one piece at a time (argv[1]: how many pieces)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import dist, engine, models, synth  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
B, T = 200, 10000
if level >= 1:
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
if level >= 2:
    ranks = dist.Ranks(backend=None)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
m = models.GRUModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
m = m.to(dev).eval()
x = synth.counts_windows(40, T, depth=50, seed=1234)
x = np.concatenate([x] * 5)[:B]
x_dev = torch.from_numpy(x).to(dev)
eng = m.engine()
if level >= 3:
    eng.set_option("rec_windows_per_tile", 0)
    eng.set_option("overlap_gemm", 1)
    eng.enable_timing(True)
hold = {}
if level >= 4:
    for i in range(8):
        with torch.inference_mode():
            hold["y"] = m.forward(x_dev)
    torch.cuda.synchronize()
if level >= 5:
    eng.set_option("scan_split", 0)
    for i in range(4):
        with torch.inference_mode():
            hold["y"] = m.forward(x_dev)
    torch.cuda.synchronize()
    eng.set_option("scan_split", 1)
    eng.set_option("scan_split_margin", 256)
    for i in range(4):
        with torch.inference_mode():
            hold["y"] = m.forward(x_dev)
    torch.cuda.synchronize()
    eng.set_option("scan_split_margin", 128)
    with torch.inference_mode():
        hold["y"] = m.forward(x_dev)
    torch.cuda.synchronize()
if level >= 3:
    eng.enable_timing(False)
x_cpu = torch.from_numpy(x)
for name, xv in (("page-locked", x_cpu.pin_memory()), ("pageable", x_cpu)):
    xb = Batch(counts_matrix=xv)
    ts = []
    for i in range(7):
        t0 = time.perf_counter(); hold["p"] = m.predict_on_batch(xb); ts.append(time.perf_counter() - t0)
    print(f"level {level}, {name} input: " + " ".join(f"{1e3 * t:.2f}" for t in ts) + f" | {eng.split()['status']} streamed={eng.timing()['host_streamed']}", flush=True)

# which side is it?  the same internal device buffer into (a) page-locked arrays of the engine's own (hipHostMalloc now),
# (b) a torch page-locked tensor allocated now, (c) the tensor predict_on_batch just returned
def t_ptr(label, xin, out_ptr, n=5):
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); eng.forward_ptr(xin, B, T, out_ptr, host=True); ts.append(time.perf_counter() - t0)
    print(f"level {level}   {label}: " + " ".join(f"{1e3 * t:.2f}" for t in ts), flush=True)


px, pp = engine.PinnedArray(x.shape), engine.PinnedArray((B, T, 5))
px.array[...] = x
t_ptr("(a) engine PinnedArray out", px.array.ctypes.data, pp.array.ctypes.data)
tp = torch.empty((B, T, 5), dtype=torch.float32, pin_memory=True)
t_ptr("(b) new torch page-locked out", px.array.ctypes.data, tp.data_ptr())
t_ptr("(c) predict_on_batch's last tensor", px.array.ctypes.data, hold["p"].data_ptr())
print(f"level {level}   pointers: PinnedArray {pp.array.ctypes.data:#x} torch-new {tp.data_ptr():#x} torch-returned {hold['p'].data_ptr():#x}", flush=True)
