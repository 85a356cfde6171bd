#!/bin/bash
# round 4, first GPU call: the fused projection kernel (bitwise vs the GEMM + recurrence pair), the restored f2/f3/a8
# tests, and the step time with and without fusion
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c1; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "fused_projection or counts_in or majority_vote or goldens_from_unmodified" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests/test_scan_split_gpu.py -m gpu -x -q -k "counts_in or full_batch" > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?" >> $O/rc.txt
for fp in 1 0; do
  MDK_FUSE_PROJ=$fp timeout 300 python bench.py --device-only --steps 10 --warmup 3 > $O/bench_dev_fp$fp.json 2> $O/bench_dev_fp$fp.err; echo "bench fp=$fp rc=$?" >> $O/rc.txt
done
MDK_FUSE_PROJ=1 timeout 300 python bench.py --device-only --steps 5 --warmup 2 --batch 1000 > $O/bench_dev_B1000_fp1.json 2> $O/bench_dev_B1000_fp1.err
MDK_FUSE_PROJ=0 timeout 300 python bench.py --device-only --steps 5 --warmup 2 --batch 1000 > $O/bench_dev_B1000_fp0.json 2> $O/bench_dev_B1000_fp0.err
tail -3 $O/pytest_a.log $O/pytest_b.log; cat $O/rc.txt; cat $O/bench_dev_fp1.json $O/bench_dev_fp0.json | cut -c1-600
