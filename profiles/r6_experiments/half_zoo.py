"""Half precision with the fp32-parity probe across the weight zoo (tests/golden/weights_zoo.npz + the round-1 trained set), 200 x 10 000:
the margin each learner settles at (fp32 parity / half), probes, and the half result against the fp32 PyTorch-CPU oracle on all 2 M
columns, next to the half SEQUENTIAL scan's own distance from that oracle (what the split adds must be nothing).
   python profiles/r6_experiments/half_zoo.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

import bench  # noqa: E402
from medaka_amd import engine, synth  # noqa: E402
from oracle import oracle  # noqa: E402

torch.set_num_threads(bench.usable_cores())
B, T, adapt = 200, 10000, 2
x = np.concatenate([synth.counts_windows(8, T, depth=50, seed=300 + s) for s in range(25)])
zoo = np.load(os.path.join(ROOT, "tests", "golden", "weights_zoo.npz"))
sets = {"trained": dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))}
for name in sorted(set(k.split("/")[0] for k in zoo.files)):
    sets[name] = {k[len(name) + 1:]: zoo[k] for k in zoo.files if k.startswith(name + "/")}


def settle(e, max_calls=40):
    margins, same, last, out = [], 0, None, None
    for _ in range(max_calls):
        out = e.forward_host(x)
        info = e.split()
        key = (info["margin"], info["fallbacks"], info["status"])
        same = same + 1 if key == last else 0
        last = key
        margins.append(info["margin"] if info["chunks"] > 1 else 0)
        if info["chunks"] <= 1 or same >= adapt + 2:
            break
    return out, margins


report = {}
for name, st in sets.items():
    e32 = engine.GruEngine(st); e32.set_option("scan_split_adapt", adapt)
    _, m32 = settle(e32); e32.close()
    eh = engine.GruEngine(st); eh.set_precision(True); eh.set_option("scan_split_adapt", adapt)
    out, mh = settle(eh)
    info = eh.split()
    eh.set_option("scan_split", 0); seq = eh.forward_host(x); eh.set_option("scan_split", 1)
    cpu = oracle.make_torch_oracle(st)
    ref = np.concatenate([cpu.predict(x[lo:lo + 50]).numpy() for lo in range(0, B, 50)])
    r = {"fp32_parity_margin": m32[-1], "half_margin": mh[-1], "half_status": info["status"], "probes": info["probes"],
         "half_split_vs_fp32_oracle": float(np.abs(out - ref).max()), "half_sequential_vs_fp32_oracle": float(np.abs(seq - ref).max()),
         "half_split_vs_half_sequential": float(np.abs(out - seq).max()),
         "argmax_identical_split": int((out.argmax(-1) == ref.argmax(-1)).sum()), "argmax_identical_sequential": int((seq.argmax(-1) == ref.argmax(-1)).sum()),
         "columns": B * T}
    report[name] = r
    print(name, json.dumps(r), flush=True)
    eh.close()
if len(sys.argv) > 1:
    json.dump(report, open(sys.argv[1], "w"), indent=1)
