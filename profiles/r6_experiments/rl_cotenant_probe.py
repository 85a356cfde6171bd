"""Would the next batch's read-level FRONT END fit under the current batch's LSTM(384) cluster recurrences?
The cheapest honest probe: two `rl_lstm384` engines on one GPU, each in a thread of its own, forwards of the same batch running
against each other with a time offset -- engine B's front end (58 ms, every CU, matrix pipe 81 % busy) meets engine A's cluster
recurrences (69 ms on 156 CUs, members spinning on each other) and vice versa.  Reported: ms per forward alone, ms per forward
together, `wide_retries`, errors.   python profiles/r6_experiments/rl_cotenant_probe.py [--reps N]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

from medaka_amd import models, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--batch", type=int, default=100)
ap.add_argument("--offset-ms", type=float, default=60.0, help="engine B starts this long after engine A")
args = ap.parse_args()
dev = torch.device("cuda", 0)
kw = dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
state = synth.synth_rl_state(seed=21, **kw)


def make():
    m = models.LatentSpaceLSTM(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
    return m.to(dev).eval()


B, P, D = args.batch, 10000, 50
xs = synth.synth_reads(8, P, D, use_dwells=True, seed=1, empty_tail=False)
x = torch.from_numpy(xs).repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous().to(dev)
ma, mb = make(), make()
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def run(m, st, n, out, delay=0.0):
    time.sleep(delay)
    ts, retries, errs = [], 0, []
    with torch.cuda.stream(st), torch.inference_mode():
        for _ in range(n):
            t0 = time.perf_counter()
            try:
                y = m(x)
                st.synchronize()
                retries += m.engine().timing()["wide_retries"]
            except Exception as exc:             # a cluster that could not place its members in time
                errs.append(str(exc)[:120])
            ts.append(1e3 * (time.perf_counter() - t0))
    out.update(ms=[round(t, 1) for t in ts], retries=retries, errors=errs)


for m, st in ((ma, streams[0]), (mb, streams[1])):
    o = {}
    run(m, st, 2, o)
alone = {}
run(ma, streams[0], args.reps, alone)
print("alone:", alone, flush=True)
ra, rb = {}, {}
ta = threading.Thread(target=run, args=(ma, streams[0], args.reps, ra))
tb = threading.Thread(target=run, args=(mb, streams[1], args.reps, rb, args.offset_ms * 1e-3))
t0 = time.perf_counter()
ta.start(); tb.start(); ta.join(); tb.join()
wall = time.perf_counter() - t0
print("together, engine A:", ra, flush=True)
print("together, engine B:", rb, flush=True)
n_ok = 2 * args.reps - len(ra["errors"]) - len(rb["errors"])
print(f"together: {n_ok} forwards in {1e3 * wall:.0f} ms = {1e3 * wall / max(n_ok, 1):.1f} ms per forward against "
      f"{sum(alone['ms']) / len(alone['ms']):.1f} ms alone; retries {ra['retries'] + rb['retries']}, errors {len(ra['errors']) + len(rb['errors'])}")
