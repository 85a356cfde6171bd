"""Round 5: what changes on the box during the 0.15-0.3 s in which strided DMA copies are slow after a stretch of
device-resident work (profiles/r4_experiments/README.md "the box's DMA needs ~0.3 s to wake up")?

Samples the driver's DPM tables (sysfs pp_dpm_* of the GPU: the `*` marks the level in use) and the PCIe link state right
before every host-path call: settled state, straight after 2 s of device-resident forwards, and while it settles again.
Also times predict_on_counts (the PCIe diet entry) with its page-locked result buffers.
"""
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import models, synth  # noqa: E402
from medaka_amd.torch_ext import Batch  # noqa: E402


def gpu_sysfs():
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            return d
    return None


SYS = gpu_sysfs()
FILES = ["pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_pcie", "current_link_speed", "current_link_width",
         "power_dpm_force_performance_level", "gpu_busy_percent", "mem_busy_percent"]


def sample():
    out = {}
    if not SYS:
        return out
    for f in FILES:
        try:
            txt = open(os.path.join(SYS, f)).read().strip()
        except OSError:
            continue
        if f.startswith("pp_dpm"):
            cur = [l.strip() for l in txt.splitlines() if l.strip().endswith("*")]
            out[f] = cur[0] if cur else txt.replace("\n", " | ")
        else:
            out[f] = txt
    return out


SHORT = bool(os.environ.get("PROBE_SHORT"))        # only the start-of-process stretch
MARK = bool(os.environ.get("PROBE_MARK"))          # call markers on stderr (to cut an AMD_LOG_LEVEL log by)
TAG = os.environ.get("PROBE_TAG", "probe")
B, T = 200, 10000
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
m = models.GRUModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
m = m.to(dev).eval()
x = synth.counts_windows(40, T, depth=50, seed=1234)
x = np.concatenate([x] * 5)[:B]
x_dev = torch.from_numpy(x).to(dev)
xb = Batch(counts_matrix=torch.from_numpy(x).pin_memory())
hold = {}
report = {"sysfs": SYS, "tables": {}}
if SYS:
    for f in FILES:
        try:
            report["tables"][f] = open(os.path.join(SYS, f)).read().strip().splitlines()
        except OSError:
            pass


def host_calls(n, label):
    rows = []
    for i in range(n):
        s = sample()
        if MARK:
            sys.stderr.write(f"### begin {label} {i}\n"); sys.stderr.flush()
        t0 = time.perf_counter()
        hold["p"] = m.predict_on_batch(xb)
        rows.append({"ms": round(1e3 * (time.perf_counter() - t0), 3), **s})
        if MARK:
            sys.stderr.write(f"### end {label} {i} {rows[-1]['ms']}\n"); sys.stderr.flush()
    report[label] = rows
    print(label, " ".join(f"{r['ms']:.2f}" for r in rows), flush=True)


def device_burst(seconds, label):
    t_end = time.perf_counter() + seconds
    n = 0
    mid = None
    while time.perf_counter() < t_end:
        with torch.inference_mode():
            hold["y"] = m.forward(x_dev)
        n += 1
        if n % 32 == 0:
            torch.cuda.synchronize()
            mid = sample()
    torch.cuda.synchronize()
    report[label] = {"forwards": n, "during": mid, "after": sample()}


for kv in filter(None, os.environ.get("PROBE_OPTS", "").split(",")):          # engine options, e.g. scan_split_adapt=0
    k, v = kv.split("=")
    m.engine().set_option(k, int(v))
SLEEP = float(os.environ.get("PROBE_SLEEP", "0"))     # idle seconds in front of the two stretches that are slow: time or count?
if SLEEP:
    torch.cuda.synchronize()
    time.sleep(SLEEP)
host_calls(60 if SHORT else 100, "warm_up")                 # into the settled state
if SHORT:
    out = os.path.join(ROOT, "gpurun_out", "r5_dma_state")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, TAG + ".json"), "w"), indent=1)
    raise SystemExit(0)
host_calls(10, "settled")
device_burst(2.0, "burst_1")
host_calls(40, "after_burst")
# does an idle stretch do the same?  (no GPU work at all for 2 s)
time.sleep(2.0)
report["after_idle_sample"] = sample()
host_calls(20, "after_idle")
# does a trickle of PCIe traffic during the burst keep the DMA awake?  (one 4 KB copy every forward)
small_h = torch.empty(1024, dtype=torch.float32).pin_memory()
small_d = torch.empty(1024, dtype=torch.float32, device=dev)
t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    with torch.inference_mode():
        hold["y"] = m.forward(x_dev)
    small_d.copy_(small_h, non_blocking=True)
torch.cuda.synchronize()
if SLEEP:
    time.sleep(SLEEP)
host_calls(20, "after_burst_with_trickle")

# the PCIe diet entry with recycled page-locked results
cnt = np.minimum(np.rint(x * 60.0), 65535).astype(np.uint16)
dep = np.full(x.shape[:2], 60, dtype=np.uint32)
ts = []
for i in range(12):
    t0 = time.perf_counter()
    hold["d"] = m.predict_on_counts(cnt, dep, decoded=True)
    ts.append(round(1e3 * (time.perf_counter() - t0), 3))
report["diet_ms"] = ts
print("diet", ts, flush=True)
ts = []
for i in range(8):
    t0 = time.perf_counter()
    hold["d2"] = m.predict_on_counts(cnt, dep)
    ts.append(round(1e3 * (time.perf_counter() - t0), 3))
report["counts_in_probs_out_ms"] = ts
print("counts in, probs out", ts, flush=True)
out = os.path.join(ROOT, "gpurun_out", "r5_dma_state")
os.makedirs(out, exist_ok=True)
json.dump(report, open(os.path.join(out, TAG + ".json"), "w"), indent=1)
