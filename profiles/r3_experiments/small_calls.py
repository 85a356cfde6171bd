"""Latency of small predict_on_batch calls (the reference's remainder pass runs B = 1 batches of arbitrary length one by
one, prediction.py:196-209)."""
import os, statistics, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from medaka_amd import models, synth
from medaka_amd.torch_ext import Batch
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
m = models.GRUModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}); m = m.to("cuda").eval()
for half in (False, True):
    if half: m.half()
    for B, T in ((1, 13), (1, 777), (1, 2048), (1, 5000), (1, 9999), (4, 5000), (10, 10000)):
        b = Batch(counts_matrix=torch.from_numpy(synth.counts_windows(B, T, seed=T)))
        for _ in range(3): m.predict_on_batch(b)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); m.predict_on_batch(b); ts.append(time.perf_counter() - t0)
        med = statistics.median(ts)
        print(f"{'half' if half else 'fp32'} B={B:2d} T={T:5d}: {med * 1e3:7.3f} ms per call = {med * 1e6 / (2 * T):.3f} us per dependent step (2 layers x T)", flush=True)
