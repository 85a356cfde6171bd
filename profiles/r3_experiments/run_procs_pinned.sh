#!/bin/bash
# K processes per GPU with page-locked input batches (what the engine's Batch.collate produces)
set -u
R=$PWD; OUT=$R/gpurun_out/r3_procs_pinned; mkdir -p "$OUT"
for B in 200 100; do for K in 1 2 3 4 5; do
  if [ "$K" = 1 ]; then
    timeout 200 python bench.py --shared-gpu --pinned-input --gpus 1 --batch $B --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 > "$OUT/procs_B${B}_K${K}.json" 2>/dev/null
  else
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $K --master-addr 127.0.0.1 --master-port $((29700 + K)) \
        bench.py --shared-gpu --pinned-input --gpus $K --batch $B --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 > "$OUT/procs_B${B}_K${K}.json" 2>/dev/null
  fi
done; done
python - "$OUT" <<'PY'
import glob, json, os, sys
for p in sorted(glob.glob(os.path.join(sys.argv[1], "procs_B*_K*.json"))):
    try:
        r = json.loads([l for l in open(p) if l.startswith("{")][-1])
        print(f"B={r['config']['batch_windows']:4d} K={r['n_gpus']}  device-resident {r['value'] / 1e6:7.1f} M columns/s   host-to-host (pinned input) {r['host_to_host']['value'] / 1e6:7.1f} M columns/s ({r['host_to_host']['ms_per_batch_median']:.2f} ms/batch/process)")
    except Exception as e:
        print(os.path.basename(p), "no result", e)
PY
