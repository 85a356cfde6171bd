import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MDK_SCAN_SPLIT_ADAPT"] = "0"
import numpy as np
from medaka_amd import engine, synth
gold = dict(np.load("tests/golden/weights_init.npz")) if os.path.exists("tests/golden/weights_init.npz") else None
import glob
print([os.path.basename(f) for f in glob.glob("tests/golden/weights*")])
st = dict(np.load("tests/golden/weights_init.npz"))
x = synth.counts_windows(24, 6000, depth=50, seed=33)
big = x * np.float32(3000.0)
e2 = engine.GruEngine(st)
outs = []
for i in range(4):
    o = e2.forward_host(big)
    print(i, e2.split())
    outs.append(o)
for i in range(1, 4):
    d = np.abs(outs[i] - outs[0])
    print("diff vs call 0:", i, float(d.max()), int((d > 0).sum()))
e2.set_option("scan_split", 0)
seq = e2.forward_host(big)
for i in range(4):
    print("vs sequential", i, float(np.abs(outs[i] - seq).max()))
