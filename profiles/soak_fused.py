"""Determinism soak of round 4's paths: N split forwards of the BASELINE batch -- fused layer 1 + head, fp32 parity and
half precision -- through (a) predict_on_batch on a reused Batch (ordinary host path), (b) batches collated in a
Batcher thread and handed over early (mdk_gru_stage_input), with a churn thread allocating / freeing pinned and pageable
buffers meanwhile.  Every result must be identical to the first, bit for bit.
    python profiles/soak_fused.py [n]"""
import os
import queue
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import models, synth  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
xs = np.concatenate([synth.counts_windows(8, 10000, seed=s) for s in range(25)])


class S:
    def __init__(self, f):
        self.features = f


stop = threading.Event()


def churn():
    while not stop.is_set():
        a = torch.empty((int(np.random.randint(1, 64)) << 20,), dtype=torch.uint8, pin_memory=True)
        b = torch.empty((int(np.random.randint(1, 64)) << 20,), dtype=torch.uint8)
        b[::4096] = 1
        del a, b


threading.Thread(target=churn, daemon=True).start()
for half in (False, True):
    m = models.GRUModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    m = m.to("cuda").eval()
    if half:
        m.half()
    eng = m.engine()
    eng.enable_timing(True)
    batch = Batch(counts_matrix=torch.from_numpy(xs))
    ref = m.predict_on_batch(batch).clone()           # (the model's first call: audited against the sequential scan)
    assert torch.equal(m.predict_on_batch(batch), ref)
    assert eng.timing()["fused_layers"] == (2 | 256 | 512) and eng.split()["status"] == "certified", (eng.timing(), eng.split())
    t0 = time.time()
    bad = 0
    for i in range(n):
        if not torch.equal(m.predict_on_batch(batch), ref):
            bad += 1
        if i % 5 == 0:
            m.predict_on_batch(Batch(counts_matrix=torch.from_numpy(xs[:1, : 1000 + 16 * (i % 50)])))
    dt = time.time() - t0
    # (b) the reference's loop shape: a Batcher thread collates (and stages) ahead, the main thread predicts
    q = queue.Queue(maxsize=8)

    def batcher():
        for _ in range(n):
            q.put(Batch.collate([S(r) for r in xs]))
        q.put(None)
    threading.Thread(target=batcher, daemon=True).start()
    staged = bad_b = 0
    t1 = time.time()
    while True:
        b = q.get()
        if b is None:
            break
        out = m.predict_on_batch(b)
        staged += bool(eng.timing()["host_streamed"] & 4)
        bad_b += not torch.equal(out, ref)
    dt_b = time.time() - t1
    print(f"{'half' if half else 'fp32'}: {n} forwards of 200 x 10000 on a reused Batch in {dt:.1f}s, {bad} differing; {n} collated in a Batcher "
          f"thread in {dt_b:.1f}s ({n * 2e6 / dt_b / 1e6:.0f} M columns/s), {staged} of them handed over early, {bad_b} differing; "
          f"audits so far {eng.split()['audits']}, failures {eng.split()['audit_failures']}", flush=True)
    assert bad == 0 and bad_b == 0 and staged >= n - 12
stop.set()
time.sleep(0.5)
