#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4final; mkdir -p $O
bash profiles/collect_round4.sh r4final bench variants rl prof pmc > $O/collect.log 2>&1
# K processes per GPU, whole fed loop in every process
for K in 2 3; do
  timeout 400 python bench.py --procs-per-gpu $K --cpu-budget 0 --extra-rl 0 --loop-batches 14 > $O/procs_K$K.json 2> $O/procs_K$K.err; echo "procs K=$K rc=$?" >> $O/rc.txt
done
tail -20 $O/collect.log; cat $O/rc.txt
