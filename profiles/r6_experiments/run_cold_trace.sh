#!/bin/bash
# ON THE GPU BOX: bash profiles/r6_experiments/run_cold_trace.sh <tag> [--fp32]   -- traced cold host-to-host calls, timeline of call 8
set -u
TAG=${1:-r6_cold}; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/tr" -o cold -- python "$R/profiles/host_trace_half.py" --cold 12 --loop 0 "$@" > "$OUT/traced.log" 2>&1
cd "$R"
db=$(find "$OUT/tr" -name "*_results.db" | head -1)
[ -n "$db" ] && python profiles/timeline.py "$db" "$OUT/timeline_cold.txt" "8" > /dev/null
find "$OUT/tr" -name "*.db" -delete
grep "cold calls" "$OUT/traced.log" | cut -c1-120
