#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c20; mkdir -p $O; rm -f gpurun_out/split_evidence.json gpurun_out/split_margins.json
timeout 2400 python -m pytest tests/test_scan_split_evidence_gpu.py -m gpu -x -q -s > $O/pytest_evidence.log 2>&1; echo "evidence rc=$?" > $O/rc.txt
grep -v "^{\|^\[medaka_amd\]" $O/pytest_evidence.log | tail -16; cat $O/rc.txt
