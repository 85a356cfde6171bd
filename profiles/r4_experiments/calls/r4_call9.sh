#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c9; mkdir -p $O
for cfg in "208 6024" "208 5512" "208 5256" "128 8024" "256 10000"; do
  set -- $cfg
  timeout 200 python bench.py --model rl384 --batch $1 --chunk-len $2 --steps 3 --warmup 1 --cpu-budget 0 > $O/rl_$1_$2.json 2> $O/rl_$1_$2.err
  python - $O/rl_$1_$2.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k=d["roofline"]["kernel_ms_per_step"]
print(d["config"]["batch_windows"], d["config"]["chunk_len"], "ms/step %.1f front %.1f rest %.1f" % (d["ms_per_step"], k["front"], d["ms_per_step"]-k["front"]))
PY
done
