#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r4c21; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python bench.py ) > $O/bench_default_box3.json 2> $O/bench_default_box3.log
cd /tmp
for tag in "half --half" "B100 --batch 100" "B1000 --batch 1000"; do
  set -- $tag; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o k -- python $R/bench.py --device-only --steps 5 --warmup 2 "$@" > $O/kt_$name.log 2>&1
  db=$(find $O/kt_$name -name "*_results.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize.py "$db" $O/kernel_stats_$name.csv > /dev/null
  find $O/kt_$name -name "*.db" -delete
done
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc_half/pass1 -o pmc -- python $R/bench.py --device-only --steps 1 --warmup 0 --half > $O/pmc_half.log 2>&1
cd $R
python profiles/pmc_step.py $O/pmc_half $O/pmc_step_half.csv > /dev/null
find $O -name "*counter_collection.csv" -size +2M -delete
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4c21/bench_default_box3.json") if l.startswith("{")][-1])
print("default:", round(d["value"]/1e6,1), round(d["ms_per_step"],3), "h2h", round(d["host_to_host"]["value"]/1e6,1), "fed", round(d["fed_loop"]["value"]/1e6,1), "frac", round(d["roofline"]["frac"],3), round(d["roofline"]["step"]["frac_issued_of_fp16_peak"],3))
PY
head -4 $O/kernel_stats_half.csv | cut -c1-110; cat $O/pmc_step_half.csv | head -12
