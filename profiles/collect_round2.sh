#!/bin/bash
# Round-2 evidence, run ON THE GPU BOX from the repo root:  bash profiles/collect_round2.sh <tag> [stage ...]
# Stages: bench variants prof pmc soak (default: all).  Writes gpurun_out/<tag>/...; summaries are copied into profiles/.
set -u
TAG=${1:-r2}; shift || true
STAGES=${*:-bench variants prof pmc soak}
R=$PWD; OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
if has bench; then
  timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench_B200.json" 2> "$OUT/bench_B200.log"
fi
if has variants; then
  timeout 120 python bench.py --half --cpu-budget 0 > "$OUT/bench_B200_half.json" 2>/dev/null
  timeout 120 python bench.py --batch 1000 --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/bench_B1000.json" 2>/dev/null
  timeout 120 python bench.py --overlap 0 --cpu-budget 0 > "$OUT/bench_B200_no_overlap.json" 2>/dev/null
  timeout 200 python bench.py --model rl384 --steps 3 --warmup 1 > "$OUT/bench_rl384_B100.json" 2> "$OUT/bench_rl384.log"
  timeout 200 python bench.py --model rl128 --steps 3 --warmup 1 > "$OUT/bench_rl128_B100.json" 2> "$OUT/bench_rl128.log"
  timeout 200 python bench.py --model rl384 --half --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/bench_rl384_B100_half.json" 2>/dev/null
fi
if has prof; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_gru" -o gru -- python "$R/bench.py" --device-only --steps 5 --warmup 2 > "$OUT/kt_gru.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_rl384" -o rl -- python "$R/bench.py" --model rl384 --steps 2 --warmup 1 --cpu-budget 0 > "$OUT/kt_rl384.log" 2>&1
  cd "$R"
  for d in kt_gru kt_rl384; do
    db=$(find "$OUT/$d" -name "*_results.db" | head -1)
    [ -n "$db" ] && python profiles/summarize.py "$db" "$OUT/${d}_kernel_stats.csv" > /dev/null
    find "$OUT/$d" -name "*.db" -delete
  done
fi
if has pmc; then
  cd /tmp
  i=0
  for PASS in \
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
    "FETCH_SIZE" \
    "WRITE_SIZE" \
    "TCC_HIT_sum TCC_MISS_sum" ; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc/pass$i" -o pmc -- \
        python "$R/bench.py" --device-only --steps 1 --warmup 0 > "$OUT/pmc_pass$i.log" 2>&1
    echo "pmc pass $i ($PASS) rc=$?"
  done
  cd "$R"
  python profiles/pmc_step.py "$OUT/pmc" "$OUT/pmc_step.csv" "$OUT/traffic.json" > /dev/null
  find "$OUT/pmc" -name "*.csv" -size +2M -delete
fi
if has soak; then
  timeout 300 python profiles/soak_wide.py 6 --compete > "$OUT/soak_wide.log" 2>&1
  tail -3 "$OUT/soak_wide.log"
fi
ls -la "$OUT"
