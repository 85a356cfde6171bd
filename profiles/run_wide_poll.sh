# sweep of the first-poll delay of k_lstm_wide<NGRP=1> (B=100, P=2000)
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
g.build()
from medaka_amd import engine
from oracle import rl_oracle
kw = dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
st = rl_oracle.synth_rl_state(seed=21, **kw)
x = rl_oracle.synth_reads(8, 2000, 4, use_dwells=True, seed=1, empty_tail=False)
x = np.ascontiguousarray(np.tile(x, (13, 1, 1, 1))[:100])
e = engine.RlEngine(st, **kw)
import os
e.set_precision(bool(int(os.environ.get("HALF", "0"))))
for ngrp in (1, 2):
    e.set_option("wide_groups_per_cluster", ngrp)
    for d in (0, 2, 4, 6, 7, 8, 10, 12, 16):
        e.set_option("wide_poll_delay", d)
        e.forward_host(x)
        t0 = time.perf_counter()
        for _ in range(3):
            e.forward_host(x)
        print("ngrp", ngrp, "delay", d, "ms", round((time.perf_counter() - t0) / 3 * 1e3, 2))
        if ngrp == 2:
            break
PY
