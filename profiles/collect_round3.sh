#!/bin/bash
# Round-3 evidence, run ON THE GPU BOX from the repo root:  bash profiles/collect_round3.sh <tag> [stage ...]
# Stages: tests bench procs variants rl soak prof pmc pmc_rl (default: tests bench procs).  Writes gpurun_out/<tag>/...
set -u
TAG=${1:-r3}; shift || true
STAGES=${*:-tests bench procs}
R=$PWD; OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  timeout 700 python -m pytest tests -m gpu -x -q -s --durations=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; grep "^\[swap\]" "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 500 python bench.py --steps 20 --warmup 5 > "$OUT/bench_B200.json" 2> "$OUT/bench_B200.log"
  timeout 200 python bench.py --batch 100 --steps 20 --warmup 5 --cpu-budget 0 > "$OUT/bench_B100.json" 2> "$OUT/bench_B100.log"
  tail -4 "$OUT/bench_B200.log"
fi
if has procs; then
  # K inference processes sharing ONE MI355X (medaka_amd.launch --procs-per-gpu K): K ranks of bench.py on device 0,
  # started behind a gloo barrier; value = device-resident aggregate, host_to_host = predict_on_batch aggregate
  for B in 200 100; do
    for K in 1 2 3 4; do
      if [ "$K" = 1 ]; then
        timeout 200 python bench.py --shared-gpu --gpus 1 --batch $B --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 \
            > "$OUT/procs_B${B}_K${K}.json" 2> "$OUT/procs_B${B}_K${K}.log"
      else
        timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $K --master-addr 127.0.0.1 --master-port $((29600 + K)) \
            bench.py --shared-gpu --gpus $K --batch $B --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 \
            > "$OUT/procs_B${B}_K${K}.json" 2> "$OUT/procs_B${B}_K${K}.log"
      fi
      echo "procs B=$B K=$K rc=$?"
    done
  done
  python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for p in sorted(glob.glob(os.path.join(out, "procs_B*_K*.json"))):
    try:
        r = json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e:
        rows.append(f"{os.path.basename(p)}: no result ({e})"); continue
    rows.append(f"B={r['config']['batch_windows']:4d} K={r['n_gpus']}  device-resident {r['value'] / 1e6:7.1f} M columns/s "
                f"({r['ms_per_step']:.2f} ms/step/process)   host-to-host {r['host_to_host']['value'] / 1e6:7.1f} M columns/s "
                f"({r['host_to_host']['ms_per_batch_median']:.2f} ms/batch/process)")
open(os.path.join(out, "procs_per_gpu.txt"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
fi
if has variants; then
  timeout 120 python bench.py --half --cpu-budget 0 --loop-batches 0 > "$OUT/bench_B200_half.json" 2>/dev/null
  timeout 150 python bench.py --batch 1000 --steps 3 --warmup 1 --cpu-budget 0 --loop-batches 0 > "$OUT/bench_B1000.json" 2>/dev/null
  timeout 200 python bench.py --batch 2000 --steps 2 --warmup 1 --cpu-budget 0 --loop-batches 0 --host-reps 3 > "$OUT/bench_B2000.json" 2>/dev/null
fi
if has rl; then
  timeout 300 python bench.py --model rl384 --steps 3 --warmup 1 --cpu-budget 40 > "$OUT/bench_rl384_B100.json" 2> "$OUT/bench_rl384.log"
  timeout 300 python bench.py --model rl128 --steps 3 --warmup 1 --cpu-budget 40 > "$OUT/bench_rl128_B100.json" 2> "$OUT/bench_rl128.log"
  timeout 200 python bench.py --model rl384 --half --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/bench_rl384_B100_half.json" 2>/dev/null
  tail -2 "$OUT/bench_rl384.log"
fi
if has soak; then
  timeout 300 python profiles/soak_wide.py 6 --compete > "$OUT/soak_wide.log" 2>&1
  tail -3 "$OUT/soak_wide.log"
fi
if has pmc_rl; then
  cd /tmp
  for M in rl384 rl128; do
    i=0
    for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      timeout 200 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$M/pass$i" -o pmc -- \
          python "$R/bench.py" --model $M --steps 1 --warmup 0 --cpu-budget 0 > "$OUT/pmc_${M}_pass$i.log" 2>&1
    done
    python "$R/profiles/traffic_rl.py" "$OUT/pmc_$M" $M 100 10000 50 "$OUT/traffic_rl.json"
    rm -rf "$OUT/pmc_$M"
  done
  cd "$R"
fi
if has prof; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_gru" -o gru -- python "$R/bench.py" --device-only --steps 5 --warmup 2 > "$OUT/kt_gru.log" 2>&1
  cd "$R"
  cd /tmp
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
ls -la "$OUT"
