#!/bin/bash
# A/B within one box: recurrence wave priority 0 (lib_prio0.so) vs 3 (shipped), K = 3 and 4 processes, alternating, 3 rounds
set -u
for ROUND in 1 2 3; do for TAG in prio0 prio3; do
  if [ $TAG = prio0 ]; then export MDK_LIB=$PWD/profiles/r3_experiments/lib_prio0.so MDK_SKIP_BUILD=1; else unset MDK_LIB MDK_SKIP_BUILD; fi
  for K in 3 4; do
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $K --master-addr 127.0.0.1 --master-port $((30000 + K + 10 * ROUND)) \
        bench.py --shared-gpu --pinned-input --gpus $K --batch 200 --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 2>/dev/null | tail -1 | \
        python -c "import json,sys; r=json.loads(sys.stdin.read()); print('round $ROUND $TAG K=$K device', round(r['value']/1e6,1), 'h2h', round(r['host_to_host']['value']/1e6,1))"
  done
done; done
