#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c8; mkdir -p $O
bash profiles/collect_round4.sh r4c8 bench prof pmc > $O/collect.log 2>&1
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "two_rank" > $O/pytest_two_rank.log 2>&1; echo "two_rank rc=$?" >> $O/rc.txt
tail -30 $O/collect.log; tail -5 $O/pytest_two_rank.log; cat $O/rc.txt
