#!/bin/bash
# Average shader clock of the recurrence kernel at a given batch: GRBM_GUI_ACTIVE (summed over the
# 8 XCDs) / 8 / kernel duration.  Two separate rocprofv3 runs (counters, then kernel trace).
#     bash profiles/clock_probe.sh <outdir> <batch>
set -u
OUT=$1; B=$2; R=$PWD
mkdir -p "$R/$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$R/$OUT/pmc_b$B" -o pmc -- \
   python "$R/bench.py" --steps 1 --warmup 0 --cpu-budget 0 --batch $B > "$R/$OUT/pmc_b$B.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/kt_b$B" -o kt -- \
   python "$R/bench.py" --steps 1 --warmup 0 --cpu-budget 0 --batch $B > "$R/$OUT/kt_b$B.log" 2>&1
cd "$R"
python - "$OUT" "$B" <<'PY'
import csv, glob, sys, collections
out, b = sys.argv[1], sys.argv[2]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_b{b}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/kt_b{b}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in cnt:
    if k in dur and "mdk" in k:
        g = sum(cnt[k]["GRBM_GUI_ACTIVE"]) / len(cnt[k]["GRBM_GUI_ACTIVE"])
        d = sum(dur[k]) / len(dur[k])
        print(f"B={b} {k:60s} dur {d/1e6:8.3f} ms  GUI_ACTIVE/8 {g/8:12.0f}  clock {g/8/d:6.3f} GHz")
PY
