#!/bin/bash
# round 4, third GPU call: whole GPU suite on the new build, host path with and without page-locked buffers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c3; mkdir -p $O
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 14 --pinned-input > $O/bench_pinned.json 2> $O/bench_pinned.err; echo "bench pinned rc=$?" > $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 > $O/bench_pageable.json 2> $O/bench_pageable.err; echo "bench pageable rc=$?" >> $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 --pinned-input --batch 100 > $O/bench_pinned_B100.json 2> $O/bench_pinned_B100.err; echo "bench B100 rc=$?" >> $O/rc.txt
grep -h "host-to-host\|fed loop" $O/bench_*.err
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -n 25 $O/pytest_gpu.log; cat $O/rc.txt
