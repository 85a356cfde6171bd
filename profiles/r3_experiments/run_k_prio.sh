#!/bin/bash
# K processes per GPU with the recurrence waves at s_setprio 3 (lib_prio3.so) against the shipped build
set -u
for LIBTAG in shipped prio3; do
  if [ $LIBTAG = prio3 ]; then export MDK_LIB=$PWD/profiles/r3_experiments/lib_prio3.so MDK_SKIP_BUILD=1; else unset MDK_LIB MDK_SKIP_BUILD; fi
  for K in 2 3 4; do
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $K --master-addr 127.0.0.1 --master-port $((29900 + K)) \
        bench.py --shared-gpu --pinned-input --gpus $K --batch 200 --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 2>/dev/null | tail -1 | \
        python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$LIBTAG K=$K device', round(r['value']/1e6,1), 'h2h', round(r['host_to_host']['value']/1e6,1))"
  done
done
