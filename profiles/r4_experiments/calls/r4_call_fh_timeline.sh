#!/bin/bash
# timeline of the host path with the final-head scan (kernel + memory-copy trace), and the number of cuts of its second half
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/fh2; mkdir -p $O; export TMPDIR=/tmp
for h in 2 3 4; do
  MDK_OUT_HALVINGS=$h timeout 300 python profiles/host_trace.py 200 2>/dev/null | grep "call [34]" | sed "s/^/halvings $h: /"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o host -- python $R/profiles/host_trace.py 200 > $O/host_trace.log 2>&1
cd $R
db=$(find $O/kt -name "*_results.db" | head -1)
python profiles/timeline.py "$db" $O/timeline.txt > /dev/null 2> $O/timeline.err
find $O/kt -name "*.db" -delete
grep -v "^\[\|^W2026\|^E2026" $O/host_trace.log | tail -10
