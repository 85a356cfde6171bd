#!/bin/bash
# ON THE GPU BOX: bash profiles/r6_experiments/run_loop_trace.sh <tag> [--fp32]   -- traced fed loop, gap analysis
set -u
TAG=${1:-r6_loop}; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/tr" -o loop -- python "$R/profiles/host_trace_half.py" --cold 12 --loop 24 "$@" > "$OUT/traced.log" 2>&1
cd "$R"
db=$(find "$OUT/tr" -name "*_results.db" | head -1)
if [ -n "$db" ]; then
  python profiles/r6_experiments/gaps.py "$db" 10 > "$OUT/gaps.txt"
  python profiles/timeline.py "$db" "$OUT/timeline_loop.txt" "-4,-3" > /dev/null
fi
find "$OUT/tr" -name "*.db" -delete
cat "$OUT/gaps.txt"; grep "fed loop" "$OUT/traced.log" | cut -c1-400
