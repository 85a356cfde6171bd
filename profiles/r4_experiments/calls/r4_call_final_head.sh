#!/bin/bash
# final-head mode (rec_fused.hpp HEAD = 2): parity tests touching it, then the default bench and the old form beside it
mkdir -p gpurun_out/fh
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_scan_split_gpu.py -m gpu -x -q -k "fused or streamed or three_layer or handed_over or full_batch_split" > gpurun_out/fh/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/fh/pytest.log
timeout 600 python bench.py > gpurun_out/fh/bench_default.json 2> gpurun_out/fh/bench_default.err; echo "bench rc=$?"
MDK_FINAL_HEAD=0 timeout 600 python bench.py > gpurun_out/fh/bench_combine.json 2> gpurun_out/fh/bench_combine.err; echo "bench(final_head=0) rc=$?"
python - <<'PY'
import json
for n in ("bench_default", "bench_combine"):
    try:
        d = json.loads(open(f"gpurun_out/fh/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"] / 1e6, 1), "M col/s", round(d["ms_per_step"], 3), "ms | host_to_host", json.dumps(d.get("host_to_host"))[:400], "| fed", json.dumps(d.get("fed_loop"))[:300])
    except Exception as e:
        print(n, "parse", e)
PY
