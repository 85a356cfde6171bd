#!/bin/bash
# whole GPU suite + the default bench on the final-head build
mkdir -p gpurun_out/fh3
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/fh3/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/fh3/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/fh3/bench_default.json 2> gpurun_out/fh3/bench_default.err; echo "bench rc=$?"
grep -h "host-to-host\|fed loop" gpurun_out/fh3/bench_default.err
cut -c1-400 gpurun_out/fh3/bench_default.json
