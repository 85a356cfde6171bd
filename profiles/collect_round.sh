#!/bin/bash
# Round-end evidence, run ON THE GPU BOX from the repo root:  bash profiles/collect_round.sh <tag>
# Writes gpurun_out/<tag>/...; the summaries are then copied into profiles/ by hand.
set -u
TAG=${1:-r1_final}; R=$PWD; OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
python bench.py > "$OUT/bench_B200.json" 2> "$OUT/bench_B200.log"
python bench.py --half --cpu-budget 0 > "$OUT/bench_B200_half.json" 2>/dev/null
python bench.py --batch 1000 --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/bench_B1000.json" 2>/dev/null
python bench.py --overlap 0 --cpu-budget 0 > "$OUT/bench_B200_no_overlap.json" 2>/dev/null
python profiles/bench_rl.py 100 10000 50 > "$OUT/rl128_B100.json" 2>/dev/null
python profiles/bench_rl.py 100 10000 50 --half > "$OUT/rl128_B100_half.json" 2>/dev/null
python profiles/bench_rl.py 100 10000 50 --wide > "$OUT/rl384_B100.json" 2>/dev/null
python profiles/bench_rl.py 256 10000 50 --wide > "$OUT/rl384_B256.json" 2>/dev/null
python profiles/bench_rl.py 100 10000 50 --wide --half > "$OUT/rl384_B100_half.json" 2>/dev/null
cd /tmp
# kernel traces (their own runs; counters below are separate passes)
rocprofv3 --kernel-trace --stats -d "$OUT/kt_gru" -o gru -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-budget 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_gru_no_overlap" -o gru -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-budget 0 --overlap 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_rl128" -o rl -- python "$R/profiles/bench_rl.py" 100 10000 50 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_rl384" -o rl -- python "$R/profiles/bench_rl.py" 100 10000 50 --wide > /dev/null 2>&1
cd "$R"
for d in kt_gru kt_gru_no_overlap kt_rl128 kt_rl384; do
  db=$(find "$OUT/$d" -name "*_results.db" | head -1)
  [ -n "$db" ] && python profiles/summarize.py "$db" "$OUT/${d}_kernel_stats.csv" > /dev/null
  find "$OUT/$d" -name "*.db" -delete
done
# counters: separate passes, no tracing (pmc_collect.sh), without the overlap so that kernels do not share the chip
bash profiles/pmc_collect.sh "gpurun_out/$TAG/pmc" --overlap 0 > "$OUT/pmc.log" 2>&1
python profiles/pmc_summarize.py "gpurun_out/$TAG/pmc" "$OUT/pmc.csv" > /dev/null
find "$OUT/pmc" -name "*.csv" -size +2M -delete
ls -la "$OUT"
