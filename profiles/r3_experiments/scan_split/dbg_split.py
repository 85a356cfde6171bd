import sys, os
import numpy as np
sys.path.insert(0, '/root/repo')
from medaka_amd import synth
from medaka_amd.engine import GruEngine
w = np.load('/root/repo/tests/golden/weights_init.npz')
eng = GruEngine({k: w[k] for k in w.files})
for B, T, S in ((8, 6000, 2), (8, 6000, 3), (8, 6000, 5)):
    x = synth.counts_windows(B, T, depth=60, seed=B).astype(np.float32)
    eng.set_option("scan_split", 0)
    ref = eng.forward_host(x)
    eng.set_option("scan_split", S)
    got = eng.forward_host(x)
    print(B, T, S, eng.split(), np.abs(got - ref).max(), flush=True)
