#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c11; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "fused_projection or half" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests/test_scan_split_gpu.py -m gpu -x -q -k "shapes_vs_sequential or half" > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?" >> $O/rc.txt
timeout 300 python bench.py --half --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 14 --extra-rl 0 > $O/bench_half.json 2> $O/bench_half.err; echo "half rc=$?" >> $O/rc.txt
timeout 300 python bench.py --device-only --steps 10 --warmup 3 > $O/bench_fp32_dev.json 2> $O/bench_fp32_dev.err
timeout 300 python bench.py --half --batch 100 --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 --extra-rl 0 > $O/bench_half_B100.json 2> $O/bench_half_B100.err
tail -n 3 $O/pytest_a.log $O/pytest_b.log; cat $O/rc.txt
python - <<'PY'
import json
for n in ("half","half_B100"):
    d=json.loads([l for l in open(f"gpurun_out/r4c11/bench_{n}.json") if l.startswith("{")][-1])
    print(n, round(d["value"]/1e6,1), round(d["ms_per_step"],3), d["scan_split"]["status"], d["roofline"]["kernel_ms_per_step"], round(d["host_to_host"]["value"]/1e6,1), d.get("fed_loop",{}).get("value"))
d=json.loads([l for l in open("gpurun_out/r4c11/bench_fp32_dev.json") if l.startswith("{")][-1]); print("fp32", round(d["value"]/1e6,1), d["ms_per_step"], d["rec_ms_per_step"])
PY
