#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r4c6; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scan_split_gpu.py tests/test_scan_split_evidence_gpu.py -m gpu -x -q > $O/pytest_split.log 2>&1; echo "split rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "wide_read_level_retry or fails_fast" > $O/pytest_wide.log 2>&1; echo "wide rc=$?" >> $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 20 --pinned-input > $O/bench_pinned.json 2> $O/bench_pinned.err; echo "bench pinned rc=$?" >> $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 > $O/bench_pageable.json 2> $O/bench_pageable.err; echo "bench pageable rc=$?" >> $O/rc.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o host -- python $R/profiles/host_trace.py 200 > $O/host_trace.log 2>&1
cd $R
db=$(find $O/kt -name "*_results.db" | head -1)
python profiles/timeline.py "$db" $O/timeline.txt > /dev/null 2> $O/timeline.err
find $O/kt -name "*.db" -delete
tail -n 6 $O/pytest_split.log $O/pytest_wide.log; cat $O/rc.txt; grep -h "host-to-host\|fed loop" $O/bench_*.err; grep -v "^\[\|^W2026\|^E2026" $O/host_trace.log | tail -10
