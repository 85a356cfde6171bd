"""Wave-specialised read-level front end (k_rl_front_ws) against k_rl_front: bitwise, several shapes, both precisions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from medaka_amd import engine, synth  # noqa: E402

bad = 0
for name, kw, dw in (("rl128", dict(), False), ("rl128u_dw", dict(bidirectional=False, use_dwells=True, lstm_size=128), True),
                     ("rl384", dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False), True),
                     ("rl384nd", dict(lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False), False)):
    st = synth.synth_rl_state(seed=5, **{**dict(lstm_size=128, cnn_size=128, use_dwells=False, bidirectional=True), **kw})
    e = engine.RlEngine(st, **kw)
    for B, P, D, seed in ((1, 1, 1, 0), (3, 250, 9, 1), (17, 1100, 6, 2), (5, 97, 30, 3)):
        x = synth.synth_reads(B, P, D, use_dwells=dw, seed=seed)
        if B == 3:
            x[1] = 0                                   # a window without reads
        for half in (False, True):
            e.set_precision(half)
            e.set_option("front_ws", 0)
            a = e.forward_host(x)
            e.set_option("front_ws", 1)
            b = e.forward_host(x)
            same = np.array_equal(a, b, equal_nan=True)
            bad += not same
            print(f"{name:10s} {B}x{P}x{D} {'half' if half else 'fp32'}: {'bit-identical' if same else 'DIFFERS'}")
    e.close()
sys.exit(1 if bad else 0)
