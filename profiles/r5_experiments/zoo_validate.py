"""The adaptive margin on every trained weight set (tests/golden/weights_zoo.npz + the round-1 set), 200 x 10000:
`python -m medaka_amd.validate` per set, fp32 parity; summary -> gpurun_out/r5_zoo/summary.json"""
import json, os, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from medaka_amd import validate
out = os.path.join(ROOT, "gpurun_out", "r5_zoo"); os.makedirs(out, exist_ok=True)
zoo = np.load(os.path.join(ROOT, "tests", "golden", "weights_zoo.npz"))
sets = {"trained": dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))}
for k in zoo.files:
    name, key = k.split("/", 1)
    sets.setdefault(name, {})[key] = zoo[k]
summary = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, st in sets.items():
        p = os.path.join(tmp, name + ".npz"); np.savez(p, **st)
        rep = validate.main([p, "--precision", "fp32", "--sample-windows", "2", "--json", os.path.join(out, f"validate_{name}.json")])
        r = rep["fp32"]
        summary[name] = {"smallest_certified_margin": r["smallest_certified_margin"], "settled_at": r["learned"]["settled_at"],
                         "status": r["learned"]["status"], "margins_over_16_calls": r["learned"]["margins_over_16_calls"],
                         "device_resident_M": r["device_resident"]["columns_per_s"] / 1e6, "sequential_M": r["sequential_scan"]["columns_per_s"] / 1e6,
                         "host_to_host_M": r["host_to_host"]["columns_per_s"] / 1e6,
                         "margin_table": {g: (v["status"], v["max_junction_delta"]) for g, v in r["margin_table_iid"].items()}}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in summary.items():
    print(k, v["smallest_certified_margin"], v["settled_at"], v["status"], round(v["device_resident_M"], 1), round(v["sequential_M"], 1), round(v["host_to_host_M"], 1))
