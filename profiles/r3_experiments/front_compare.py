"""Bitwise comparison of the read-level forward between two builds of the engine (MDK_LIB selects the library):
   python front_compare.py dump <out.npz>   /   python front_compare.py check <a.npz> <b.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump(path):
    from medaka_amd import engine, synth
    out = {}
    for name, kw, dw in (("rl128", dict(), False), ("rl128u_dw", dict(bidirectional=False, use_dwells=True, lstm_size=128), True),
                         ("rl384", dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False), True),
                         ("rl384nd", dict(lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False), False)):
        st = synth.synth_rl_state(seed=5, **{**dict(lstm_size=128, cnn_size=128, use_dwells=False, bidirectional=True), **kw})
        e = engine.RlEngine(st, **kw)
        for B, P, D, seed in ((3, 250, 9, 1), (17, 1100, 6, 2)):
            x = synth.synth_reads(B, P, D, use_dwells=dw, seed=seed)
            for half in (False, True):
                e.set_precision(half)
                out[f"{name}/{B}x{P}x{D}/{'half' if half else 'fp32'}"] = e.forward_host(x)
        e.close()
    np.savez(path, **out)
    print("dumped", len(out), "cases with", os.environ.get("MDK_LIB", "the in-tree library"))


def check(a, b):
    A, Bz = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k], Bz[k], equal_nan=True)
        d = float(np.nanmax(np.abs(A[k] - Bz[k])))
        print(f"{k:32s} {'bit-identical' if same else f'DIFFERS max|d| {d:.3e}'}")
        bad += not same
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else check(sys.argv[2], sys.argv[3])
