import sys, ctypes, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g; g.build()
import torch
from medaka_amd import engine, synth, lib
st = dict(np.load('tests/golden/weights_trained.npz'))
e = engine.GruEngine(st)
B, T = int(sys.argv[1]), int(sys.argv[2])
x = synth.uniform_windows(B, T, seed=1)
e.set_option("ablate", 64)
e.set_option("fuse_l0", int(sys.argv[3]) if len(sys.argv) > 3 else 1)
e.forward_host(x)
buf = (ctypes.c_ulonglong * 384)()
lib.check(lib.load().mdk_gru_debug_read(e._h, buf, 384), "dbg")
a = np.array(buf[:], dtype=np.float64).reshape(2, 4, 8, 6) / T   # [dir][block][wave][phase] cycles per step (last layer launched)
names = ["refill/loop", "LDS read", "MFMA+sig", "drain+tanh", "split+write", "barrier"]
print("cycles per step, dir 0, block 0, per wave:")
for w in range(8): print(w, " ".join(f"{names[i]}={a[0,0,w,i]:7.1f}" for i in range(6)), "sum=%.1f" % a[0,0,w].sum())
print("mean over waves/blocks:", " ".join(f"{names[i]}={a[..., i].mean():7.1f}" for i in range(6)), "sum=%.1f" % a.sum(-1).mean())
