#!/bin/bash
# Round-6 evidence, run ON THE GPU BOX from the repo root:  bash profiles/collect_round6.sh <tag> [stage ...]
# Stages: tests bench variants rl prof pmc pmc_rl (default: tests bench).  Writes gpurun_out/<tag>/...
set -u
TAG=${1:-r6}; shift || true
STAGES=${*:-tests bench}
R=$PWD; OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=10 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; grep "^\[swap\]" "$OUT/pytest_gpu.log" | head -3
fi
if has bench; then
  ( time timeout 900 python bench.py --full-out "$OUT/bench_default_full.json" ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.log"; echo "bench default rc=$?"
  wc -c "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.log" | cut -c1-300
  timeout 300 python bench.py --batch 100 --steps 20 --warmup 5 --cpu-budget 0 --extra-rl 0 --full-out "$OUT/bench_B100_full.json" > "$OUT/bench_B100.json" 2> "$OUT/bench_B100.log"
fi
if has variants; then
  timeout 200 python bench.py --half --cpu-budget 0 --loop-batches 0 --extra-rl 0 > "$OUT/bench_B200_half.json" 2>/dev/null
  timeout 200 python bench.py --batch 1000 --steps 3 --warmup 1 --cpu-budget 0 --loop-batches 0 --extra-rl 0 > "$OUT/bench_B1000.json" 2>/dev/null
  timeout 200 python bench.py --scan-split 0 --cpu-budget 0 --loop-batches 14 --extra-rl 0 > "$OUT/bench_B200_sequential.json" 2>/dev/null
  MDK_FUSE_PROJ=0 timeout 200 python bench.py --cpu-budget 0 --loop-batches 0 --extra-rl 0 > "$OUT/bench_B200_unfused.json" 2>/dev/null
fi
if has small; then
  python profiles/r3_experiments/small_calls.py > "$OUT/small_calls.txt" 2>/dev/null; tail -14 "$OUT/small_calls.txt"
fi
if has loops; then
  for p in fp32 half; do python profiles/r6_experiments/early_start_probe.py --precisions $p 2>/dev/null | grep early_start= ; done > "$OUT/fed_loops.txt"; cat "$OUT/fed_loops.txt"
fi
if has rl; then
  timeout 300 python bench.py --model rl384 --steps 3 --warmup 1 --cpu-budget 40 > "$OUT/bench_rl384_B100.json" 2> "$OUT/bench_rl384.log"
  timeout 300 python bench.py --model rl128 --steps 3 --warmup 1 --cpu-budget 40 > "$OUT/bench_rl128_B100.json" 2> "$OUT/bench_rl128.log"
  tail -2 "$OUT/bench_rl384.log"
fi
if has prof; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_gru" -o gru -- python "$R/bench.py" --device-only --steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-5} > "$OUT/kt_gru.log" 2>&1
  cd "$R"
  db=$(find "$OUT/kt_gru" -name "*_results.db" | head -1)
  [ -n "$db" ] && python profiles/summarize.py "$db" "$OUT/kt_gru_kernel_stats.csv" > /dev/null
  find "$OUT/kt_gru" -name "*.db" -delete
fi
if has prof_more; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_half" -o half -- python "$R/bench.py" --half --device-only --steps 20 --warmup 5 > "$OUT/kt_half.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_rl384" -o rl -- python "$R/bench.py" --model rl384 --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/kt_rl384.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_rl128" -o rl -- python "$R/bench.py" --model rl128 --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/kt_rl128.log" 2>&1
  cd "$R"
  for n in half rl384 rl128; do
    db=$(find "$OUT/kt_$n" -name "*_results.db" | head -1)
    [ -n "$db" ] && python profiles/summarize.py "$db" "$OUT/kt_${n}_kernel_stats.csv" > /dev/null
    find "$OUT/kt_$n" -name "*.db" -delete
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
if has pmc_half; then
  cd /tmp
  i=0
  for PASS in \
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
    "FETCH_SIZE" \
    "WRITE_SIZE" ; do
    i=$((i+1))
    # (MDK_SCAN_SPLIT_PROBE=0: the one forward of the process is the half-precision one, not its fp32-parity probe in front of it)
    MDK_SCAN_SPLIT_PROBE=0 timeout 200 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_half/pass$i" -o pmc -- \
        python "$R/bench.py" --half --device-only --steps 1 --warmup 0 > "$OUT/pmc_half_pass$i.log" 2>&1
    echo "pmc half pass $i ($PASS) rc=$?"
  done
  cd "$R"
  python profiles/pmc_step.py "$OUT/pmc_half" "$OUT/pmc_step_half.csv" "$OUT/traffic_half.json" > /dev/null
  find "$OUT/pmc_half" -name "*.csv" -size +2M -delete
fi
ls -la "$OUT"
