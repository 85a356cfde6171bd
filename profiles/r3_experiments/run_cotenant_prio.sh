#!/bin/bash
# the same co-tenant experiment with the recurrence waves at s_setprio 3
set -u
export MDK_LIB=$PWD/profiles/r3_experiments/lib_prio3.so MDK_SKIP_BUILD=1
for B in 200 1000; do
  echo "--- prio 3, bench alone B=$B"; python bench.py --device-only --batch $B --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms_per_step', round(r['ms_per_step'],2), 'rec', round(r['rec_ms_per_step'],2), 'gi', round(r['gi_ms_per_step'],2))"
  echo "--- prio 3, bench B=$B next to the burner"
  ./profiles/probes/mfma_burn 25 256 > /tmp/burn.log &
  BP=$!
  sleep 12
  python bench.py --device-only --batch $B --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms_per_step', round(r['ms_per_step'],2), 'rec', round(r['rec_ms_per_step'],2), 'gi', round(r['gi_ms_per_step'],2))"
  wait $BP; cat /tmp/burn.log
done
