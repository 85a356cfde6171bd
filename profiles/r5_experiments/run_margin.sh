#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r5_margin}; mkdir -p $O
timeout 1500 python -m pytest tests/test_scan_split_gpu.py tests/test_scan_split_evidence_gpu.py -x -q -m gpu -s --durations=5 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -8 $O/tests.log
grep -E "maj1: margins|weights x3|half precision, split" $O/tests.log
