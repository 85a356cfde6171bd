"""Determinism soak of the streamed host path (mdk_gru_forward): N back-to-back predict_on_batch calls on the
BASELINE batch must return identical bits every time, with a second thread allocating / freeing pinned and
pageable buffers meanwhile (event pool reuse, copy streams, torch's host allocator under the engine's feet).
    python profiles/soak_host_path.py [n]"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import models, synth  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
st = dict(np.load("tests/golden/weights_trained.npz"))
m = models.GRUModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
m = m.to("cuda").eval()
x = torch.from_numpy(np.concatenate([synth.counts_windows(8, 10000, seed=s) for s in range(25)]))
batch = Batch(counts_matrix=x)
ref = m.predict_on_batch(batch).clone()
stop = threading.Event()


def churn():
    while not stop.is_set():
        a = torch.empty((int(np.random.randint(1, 64)) << 20,), dtype=torch.uint8, pin_memory=True)
        b = torch.empty((int(np.random.randint(1, 64)) << 20,), dtype=torch.uint8)
        b[::4096] = 1
        del a, b


t = threading.Thread(target=churn, daemon=True)
t.start()
t0 = time.time()
bad = 0
for i in range(n):
    out = m.predict_on_batch(batch)
    if not torch.equal(out, ref):
        bad += 1
    if i % 3 == 0:          # a differently shaped call in between (remainder-pass shape)
        m.predict_on_batch(Batch(counts_matrix=x[:1, : 1000 + 16 * (i % 50)]))
stop.set()
t.join()
dt = time.time() - t0
print(f"{n} forwards of 200 x 10000 in {dt:.1f}s ({n * 2e6 / dt / 1e6:.1f} M columns/s incl. the small calls), {bad} differing")
assert bad == 0
