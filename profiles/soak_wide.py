"""Determinism soak of the LSTM(384) cluster exchange: repeated forwards must return identical bits
(a lost or stale granule would show up as a changed result or a time-out error).
    python profiles/soak_wide.py [repeats]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import engine  # noqa: E402
from oracle import rl_oracle  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kw = dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
st = rl_oracle.synth_rl_state(seed=21, **kw)
e = engine.RlEngine(st, **kw)
t0 = time.time()
for B, P, D in ((40, 3000, 3), (300, 1200, 2)):
    x = rl_oracle.synth_reads(B, P, D, use_dwells=True, seed=B)
    for half in (False, True):
        e.set_precision(half)
        for wt in (0, 1):
            e.set_option("wide_write_through", wt)
            ref = e.forward_host(x)
            bad = sum(not np.array_equal(e.forward_host(x), ref) for _ in range(reps))
            print(f"B={B} P={P} half={half} write_through={wt}: {reps} repeats, {bad} differing", flush=True)
            assert bad == 0
print(f"soak ok in {time.time() - t0:.0f}s")
