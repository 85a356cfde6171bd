import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd import synth
from medaka_amd.engine import GruEngine
w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden", "weights_init.npz"))
eng = GruEngine({k: w[k] for k in w.files})
eng.set_precision(True)
B, T = 200, 10000
x = synth.counts_windows(B, T, depth=60, seed=B).astype(np.float32)
eng.set_option("scan_split", 0)
ref = eng.forward_host(x)
for tw in (0, 4, 8, 16):
    eng.set_option("rec_windows_per_tile", tw)
    for S in (2, 3, 5):
        eng.set_option("scan_split", S)
        got = eng.forward_host(x)
        d = np.abs(got - ref)
        bad = np.argwhere(d > 1e-3)
        print("tile windows", tw, "S", S, eng.split(), "max|dp|", d.max(), "n>1e-3", len(bad), "first", bad[:3].tolist(), "windows", sorted(set(bad[:, 0].tolist()))[:12], flush=True)
