#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r4c7; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "fused_projection" > $O/pytest_fused.log 2>&1; echo "fused rc=$?" > $O/rc.txt
timeout 900 python -m pytest tests/test_scan_split_gpu.py -m gpu -x -q > $O/pytest_split.log 2>&1; echo "split rc=$?" >> $O/rc.txt
for fh in 1 0; do
  MDK_FUSE_HEAD=$fh timeout 300 python bench.py --device-only --steps 10 --warmup 3 > $O/bench_dev_fh$fh.json 2> $O/bench_dev_fh$fh.err; echo "bench fh=$fh rc=$?" >> $O/rc.txt
done
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 20 --pinned-input > $O/bench_pinned.json 2> $O/bench_pinned.err; echo "bench pinned rc=$?" >> $O/rc.txt
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 --pinned-input --stream-host 2 > $O/bench_pinned_sh2.json 2> $O/bench_pinned_sh2.err; echo "bench pinned sh2 rc=$?" >> $O/rc.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o host -- python $R/profiles/host_trace.py 200 > $O/host_trace.log 2>&1
cd $R
db=$(find $O/kt -name "*_results.db" | head -1)
python profiles/timeline.py "$db" $O/timeline.txt > /dev/null 2> $O/timeline.err
find $O/kt -name "*.db" -delete
tail -n 6 $O/pytest_fused.log $O/pytest_split.log; cat $O/rc.txt; cut -c1-330 $O/bench_dev_fh1.json; echo; cut -c1-330 $O/bench_dev_fh0.json; echo; grep -h "host-to-host\|fed loop" $O/bench_*.err; grep -v "^\[\|^W2026\|^E2026" $O/host_trace.log | tail -10
