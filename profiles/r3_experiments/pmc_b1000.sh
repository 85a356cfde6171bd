#!/bin/bash
# counters + kernel trace of one forward at 1000 x 10000 (throughput regime): clocks, matrix-pipe busy, traffic
set -u
R=$PWD; OUT=$R/gpurun_out/r3_b1000; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for PASS in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc/pass$i" -o pmc -- python "$R/bench.py" --device-only --batch 1000 --steps 1 --warmup 0 > /dev/null 2>&1
done
python "$R/profiles/pmc_step.py" "$OUT/pmc" "$OUT/pmc_step_B1000.csv" > /dev/null
rm -rf "$OUT/pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o gru -- python "$R/bench.py" --device-only --batch 1000 --steps 3 --warmup 1 > "$OUT/kt.log" 2>&1
cd "$R"
db=$(find "$OUT/kt" -name "*_results.db" | head -1)
[ -n "$db" ] && python profiles/summarize.py "$db" "$OUT/kernel_stats_B1000.csv" > /dev/null
rm -rf "$OUT/kt"
grep -E "GRBM_GUI_ACTIVE|SQ_VALU_MFMA_BUSY|FETCH_SIZE|WRITE_SIZE" "$OUT/pmc_step_B1000.csv" | grep -v fallback
head -6 "$OUT/kernel_stats_B1000.csv" | cut -c1-160
