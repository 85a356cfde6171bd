#!/bin/bash
# how much matrix-pipe time does the latency-bound recurrence leave to a co-resident MFMA kernel, and at what cost?
set -u
echo "--- alone"; ./profiles/probes/mfma_burn 3 256
for B in 200 1000; do
  echo "--- bench alone B=$B"; python bench.py --device-only --batch $B --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms_per_step', round(r['ms_per_step'],2), 'rec', round(r['rec_ms_per_step'],2), 'gi', round(r['gi_ms_per_step'],2))"
  echo "--- bench B=$B next to the burner"
  ./profiles/probes/mfma_burn 25 256 > /tmp/burn.log &
  BP=$!
  sleep 12          # torch import + engine set-up happen while the burner already runs
  python bench.py --device-only --batch $B --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms_per_step', round(r['ms_per_step'],2), 'rec', round(r['rec_ms_per_step'],2), 'gi', round(r['gi_ms_per_step'],2))"
  wait $BP; cat /tmp/burn.log
done
