"""Soak of the pipelined staged entry (include/medaka_amd.h `mdk_gru_forward_pipelined`): a long random sequence of what a loader
and a caller can do -- batches collated 1..5 ahead, redeemed in order, out of order, after an in-place edit, with another
entry of the model or another shape in between, engine options changed, the learner moving margins -- every result compared
with the lone call's for the same input (bitwise with the learner held still, 2e-6 / 4e-4 with it running).
   python profiles/r6_experiments/soak_pipeline.py [--iters N] [--half] [--sequential] [--adapt N] [--seed S]"""
import argparse
import os
import random
import sys
import time

# (the GPU box shows 256 CPUs and grants 16: OpenMP workers that spin after a parallel region exhaust the cgroup's quota and the
# whole process is throttled for tens of milliseconds at a time -- seen here as "every call takes 20 ms" in one run of four)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

import bench  # noqa: E402
from medaka_amd import models, synth  # noqa: E402

torch.set_num_threads(bench.usable_cores())
from medaka_amd.torch_ext import Batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--half", action="store_true")
ap.add_argument("--sequential", action="store_true")
ap.add_argument("--adapt", type=int, default=0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--windows", type=int, default=48)
ap.add_argument("--columns", type=int, default=4096)
ap.add_argument("--opt", action="append", default=[], help="engine option key=value, e.g. overlap_gemm=0")
ap.add_argument("--timing", action="store_true", help="engine timing on (no batch is started ahead then): device time of every call")
args = ap.parse_args()
rnd = random.Random(args.seed)


class S:
    def __init__(self, f):
        self.features = f


state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
m = models.GRUModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
m = m.to("cuda").eval()
if args.half:
    m.half()
eng = m.engine()
eng.set_option("scan_split_adapt", args.adapt)
if args.sequential:
    eng.set_option("scan_split", 0)
W, T = args.windows, args.columns
for kv in args.opt:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
if args.timing:
    eng.enable_timing(True)
pool = [synth.counts_windows(W, T, depth=40, seed=900 + i) for i in range(6)]
short = [p[: W // 3] for p in pool[:2]]
want = [eng.forward_host(x).copy() for x in pool]
want_short = [eng.forward_host(x).copy() for x in short]
tol = 0.0 if args.adapt == 0 else (4e-4 if args.half else 2e-6)


def check(out, ref, what):
    d = float(np.abs(out.numpy() - ref).max())
    assert d <= tol, (what, d)


def cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}


cs0 = cpu_stat()
t0 = time.time()
queue = []          # (batch, expected, label)
slow = []
started = n_calls = 0
for it in range(args.iters):
    for _ in range(rnd.randint(1, 5)):
        if rnd.random() < 0.1:
            k = rnd.randrange(2)
            queue.append((Batch.collate([S(r) for r in short[k]]), want_short[k], f"short{k}"))
        else:
            k = rnd.randrange(6)
            queue.append((Batch.collate([S(r) for r in pool[k]]), want[k], f"pool{k}"))
    while queue:
        r = rnd.random()
        if r < 0.05 and len(queue) > 1:
            b, ref, lab = queue.pop(rnd.randrange(1, len(queue)))            # out of order
        else:
            b, ref, lab = queue.pop(0)
        if r > 0.97:                                                          # another entry of the model in between
            k = rnd.randrange(6)
            d = float(np.abs(eng.forward_host(pool[k]) - want[k]).max())
            assert d <= tol, ("forward_host in between", d)
        if 0.93 < r <= 0.97:                                                  # edited in place after the hand-over
            k = rnd.randrange(6)
            if b.counts_matrix.shape[0] == W:
                b.counts_matrix.copy_(torch.from_numpy(pool[k]))
                ref, lab = want[k], f"edited->pool{k}"
        if 0.90 < r <= 0.93:                                                  # an option changes under a batch started ahead
            eng.set_option("stage_overlap", rnd.choice([0, 1, 2]))
        tc = time.perf_counter()
        out = m.predict_on_batch(b)
        slow.append((time.perf_counter() - tc, lab, bool(eng.timing()["host_streamed"] & 8), bool(eng.timing()["host_streamed"] & 4), (eng.split()["audited"], round(eng.timing()["total_ms"], 2))))
        n_calls += 1
        started += bool(eng.timing()["host_streamed"] & 8)
        check(out, ref, (it, lab))
        if rnd.random() < 0.3:
            break                                                             # leave the rest staged; collate more first
info = eng.split()
print(f"soak ok: {n_calls} calls, {started} found their forward started ahead, {time.time() - t0:.1f} s; half={args.half} "
      f"sequential={args.sequential} adapt={args.adapt}; last split: {info['status']} margin {info['margin']} fallbacks {info['fallbacks']} "
      f"audits {info['audits']} probes {info['probes']}")
cs1 = cpu_stat()
print("cgroup cpu.stat over the loop:", {k: cs1[k] - cs0[k] for k in cs1 if k in cs0 and cs1[k] != cs0[k]}, "threads:", len(os.listdir("/proc/self/task")))
ts = sorted(t for t, *_ in slow)
print(f"predict_on_batch: median {1e3 * ts[len(ts) // 2]:.2f} ms, p90 {1e3 * ts[int(0.9 * len(ts))]:.2f} ms, max {1e3 * ts[-1]:.2f} ms; slowest: "
      + "; ".join(f"{1e3 * t:.1f} ms {lab} ahead={a} staged={st} audited={au}" for t, lab, a, st, au in sorted(slow, reverse=True)[:6]))
eng.close()
