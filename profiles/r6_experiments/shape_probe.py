import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g; g.build()
import torch
from medaka_amd import engine, synth
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
x = synth.counts_windows(48, 4096, depth=40, seed=1)
for half in (False, True):
    for split in (1, 0):
        e = engine.GruEngine(st); e.set_precision(half); e.set_option("scan_split", split); e.set_option("scan_split_adapt", 0)
        e.enable_timing(True)
        for _ in range(3): e.forward_host(x)
        t0 = time.perf_counter()
        for _ in range(20): e.forward_host(x)
        dt = (time.perf_counter() - t0) / 20
        print("half", half, "split", split, round(dt * 1e3, 3), "ms", e.split()["chunks"], e.split()["columns"], e.timing())
        e.close()
