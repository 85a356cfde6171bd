#!/bin/bash
# Clock and matrix-pipe occupancy of the layer-1 kernel, alternating (MDK_ROLL=0) against rolled (MDK_ROLL=1):
# one rocprofv3 --pmc pass and one --kernel-trace pass per form, 12 forwards each (separate runs: counters and traces
# are never collected together).   bash profiles/r5_experiments/pmc_roll.sh [extra env ...]
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD
O=$R/gpurun_out/r5_roll_pmc; mkdir -p $O; export TMPDIR=/tmp
for mode in 0 1; do
  cd /tmp
  MDK_ROLL=$mode "$@" timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --output-format csv -d $O/pmc$mode -o pmc -- python $R/bench.py --device-only --steps 10 --warmup 2 > $O/pmc$mode.log 2>&1
  MDK_ROLL=$mode "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$mode -o kt -- python $R/bench.py --device-only --steps 10 --warmup 2 > $O/kt$mode.log 2>&1
  cd $R
done
python - $O <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode in (0, 1):
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/kt{mode}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0][:48]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(cnt):
        if k in dur and ("k_rec" in k):
            c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
            d = sum(dur[k]) / len(dur[k])
            g = c.get("GRBM_GUI_ACTIVE", 0) / 8
            # SQ counters are summed over the chip's SIMDs/CUs: per-SIMD share = value / 1024
            print(f"MDK_ROLL={mode} {k:48s} n={len(dur[k]):3d} dur {d/1e6:7.3f} ms clock {g/d if d else 0:5.3f} GHz  "
                  f"mfma_busy/GUI {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/g if g else 0:5.3f}  insts_mfma {c.get('SQ_INSTS_MFMA',0):.3e}  "
                  f"sq_busy {c.get('SQ_BUSY_CYCLES',0):.3e} lds_conf/idx {c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):5.3f} "
                  f"vmem_cyc {c.get('SQ_INST_CYCLES_VMEM',0):.3e}")
PY
find $O -name "*.csv" -size +1M -delete
