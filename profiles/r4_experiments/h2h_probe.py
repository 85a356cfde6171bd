"""Host-to-host call of the split scan, three ways in one process: the engine's forward on page-locked arrays
(profiles/host_trace.py's call), `model.predict_on_batch` on a page-locked tensor (bench.py's call), and the latter with the
previous result still held (bench.py keeps it)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import engine, models, synth  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402

B, T = 200, 10000
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
x = np.concatenate([synth.counts_windows(8, T, depth=50, seed=s) for s in range(-(-B // 8))])[:B]
m = models.GRUModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
m = m.to("cuda").eval()
eng = m.engine()
eng.set_option("scan_split_audit", 0)
px, pp = engine.PinnedArray(x.shape), engine.PinnedArray((B, T, 5))
px.array[...] = x


def show(label, ts):
    print(f"{label}: " + " ".join(f"{1e3 * t:.2f}" for t in ts) + f"  | {eng.split()['status']} streamed={eng.timing()['host_streamed']}", flush=True)


ts = []
for i in range(6):
    t0 = time.perf_counter(); eng.forward_host(px.array, out=pp.array); ts.append(time.perf_counter() - t0)
show("engine.forward_host, page-locked in/out", ts)
xb = Batch(counts_matrix=torch.from_numpy(x).pin_memory())
ts = []
for i in range(6):
    t0 = time.perf_counter(); out = m.predict_on_batch(xb); ts.append(time.perf_counter() - t0); del out
show("predict_on_batch, result dropped before the next call", ts)
ts, hold = [], {}
for i in range(6):
    t0 = time.perf_counter(); hold["p"] = m.predict_on_batch(xb); ts.append(time.perf_counter() - t0)
show("predict_on_batch, previous result held", ts)
for opt in (0, 1):
    eng.set_option("stream_host", opt)
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); hold["p"] = m.predict_on_batch(xb); ts.append(time.perf_counter() - t0)
    show(f"  ... stream_host={opt}", ts)
eng.enable_timing(True)
ts = []
for i in range(4):
    t0 = time.perf_counter(); hold["p"] = m.predict_on_batch(xb); ts.append(time.perf_counter() - t0)
show("  ... with kernel timing on", ts)
print(eng.timing())

# ---- what in bench.py's history makes the same call 2.8 ms slower?  one suspect at a time
eng.enable_timing(False)


def again(label):
    ts = []
    for i in range(5):
        t0 = time.perf_counter(); hold["p"] = m.predict_on_batch(xb); ts.append(time.perf_counter() - t0)
    show(label, ts)


again("baseline again")
xd = torch.from_numpy(x).cuda()
yd = torch.empty(B, T, 5, device="cuda")
for i in range(3):
    eng.forward_ptr(xd.data_ptr(), B, T, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
again("after device-resident forwards on torch's stream")
eng.set_option("scan_split", 0)
eng.forward_ptr(xd.data_ptr(), B, T, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
eng.set_option("scan_split", 1)
again("after a sequential forward (scan_split 0 -> 1)")
eng.set_option("scan_split_margin", 256)
eng.forward_ptr(xd.data_ptr(), B, T, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
eng.set_option("scan_split_margin", 128)
again("after a forward at margin 256 (-> 128)")
eng.set_option("scan_split_audit", 1)
again("audits on (first call audited)")
hold.clear()
big = [torch.empty(64 << 20, dtype=torch.uint8).fill_(1) for _ in range(16)]      # 1 GB of pageable memory touched
del big
xb2 = Batch(counts_matrix=torch.from_numpy(x.copy()).pin_memory())
ts = []
for i in range(5):
    t0 = time.perf_counter(); hold["p"] = m.predict_on_batch(xb2); ts.append(time.perf_counter() - t0)
show("fresh page-locked input and outputs", ts)
