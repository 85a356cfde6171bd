#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c10; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python bench.py --half --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 --extra-rl 0 > $O/bench_half.json 2> $O/bench_half.err; echo "half rc=$?" >> $O/rc.txt
MDK_FUSE_PROJ=0 timeout 300 python bench.py --half --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 --extra-rl 0 > $O/bench_half_unfused.json 2> $O/bench_half_unfused.err; echo "half unfused rc=$?" >> $O/rc.txt
tail -n 14 $O/pytest_gpu.log; cat $O/rc.txt
python - <<'PY'
import json
for n in ("half","half_unfused"):
    d=json.loads([l for l in open(f"gpurun_out/r4c10/bench_{n}.json") if l.startswith("{")][-1])
    print(n, round(d["value"]/1e6,1), round(d["ms_per_step"],3), d["scan_split"]["status"], d["scan_split"]["max_delta"], d["roofline"]["kernel_ms_per_step"], round(d["host_to_host"]["value"]/1e6,1))
PY
