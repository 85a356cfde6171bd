import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.build()
from medaka_amd import engine, synth
zoo = np.load("tests/golden/weights_zoo.npz")
st = {k[len("depthmix/"):]: zoo[k] for k in zoo.files if k.startswith("depthmix/")}
B, T = 200, 10000
x = np.concatenate([synth.counts_windows(8, T, depth=50, seed=300 + s) for s in range(25)])
e = engine.GruEngine(st); e.set_precision(True); e.set_option("scan_split_adapt", 0); e.set_option("scan_split_audit", 2)
out = e.forward_host(x); info = e.split(); print("split:", info["status"], info["margin"], "audited", info["audited"], "audit_max_dp", info["audit_max_dp"], "max_delta", info["max_delta"])
e.set_option("scan_split", 0); seq = e.forward_host(x); print("non-lean sequential vs split:", float(np.abs(out - seq).max()))
e.set_option("fuse_proj", 2); e.set_option("rec_windows_per_tile", 8); seq2 = e.forward_host(x); print("fused 8-window sequential vs non-lean:", float(np.abs(seq2 - seq).max()), "vs split", float(np.abs(out - seq2).max()), e.timing()["fused_layers"])
