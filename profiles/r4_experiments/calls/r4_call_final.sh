#!/bin/bash
# final tree of round 4: whole GPU suite, every bench line, kernel stats and counters
cd "$GRAFT_REPO_ROOT" || exit 1
bash profiles/collect_round4.sh r4final tests bench variants prof pmc 2>&1 | tail -30
O=gpurun_out/r4final
for K in 2 3; do
  timeout 400 python bench.py --procs-per-gpu $K --cpu-budget 0 --extra-rl 0 --loop-batches 14 > $O/procs_K$K.json 2> $O/procs_K$K.err; echo "procs K=$K rc=$?"
done
for f in bench_default bench_B100 bench_B200_half bench_B1000 bench_B200_sequential bench_B200_unfused procs_K2 procs_K3; do
  python - "$O/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("host_to_host") or {}
    f = d.get("fed_loop") or d.get("fed_loop_shared") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M", round(d["ms_per_step"], 3), "ms | h2h", round(h.get("value", 0) / 1e6, 1), round(h.get("ms_per_batch_median", 0), 2),
          "| fed", round((f.get("value") or 0) / 1e6, 1), "| frac", round((d.get("roofline") or {}).get("frac", 0), 3))
except Exception as e:
    print(sys.argv[1], "parse", e)
PY
done
