#!/bin/bash
# round 4, second GPU call: fused-projection bitwise test, split host streaming, spot audit, structured-input evidence
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c2; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "fused_projection" > $O/pytest_fused.log 2>&1; echo "fused rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_scan_split_gpu.py -m gpu -x -q > $O/pytest_split.log 2>&1; echo "split rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_scan_split_evidence_gpu.py -m gpu -x -q -s -k "structured_pileups or spot_audit" > $O/pytest_evidence.log 2>&1; echo "evidence rc=$?" >> $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 14 --pinned-input > $O/bench_pinned.json 2> $O/bench_pinned.err; echo "bench pinned rc=$?" >> $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 > $O/bench_pageable.json 2> $O/bench_pageable.err; echo "bench pageable rc=$?" >> $O/rc.txt
tail -n 4 $O/pytest_fused.log $O/pytest_split.log $O/pytest_evidence.log; cat $O/rc.txt; grep -h "host-to-host\|fed loop" $O/bench_pinned.err $O/bench_pageable.err
