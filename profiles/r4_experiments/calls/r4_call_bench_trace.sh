#!/bin/bash
# kernel + memory-copy trace of bench.py's host-to-host section (why 11.3 ms there and 8.5 ms in h2h_probe.py?)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/fh5; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o bench -- python $R/bench.py --cpu-budget 0 --loop-batches 0 > $O/bench.json 2> $O/bench.err
cd $R
grep -h "host-to-host," $O/bench.err
db=$(find $O/kt -name "*_results.db" | head -1)
python profiles/timeline.py "$db" $O/timeline.txt "-12,-8,-2" > /dev/null 2> $O/timeline.err
find $O/kt -name "*.db" -delete
wc -l $O/timeline.txt
