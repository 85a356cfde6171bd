#!/bin/bash
# Round 6: where the half-precision host path loses its third.  ON THE GPU BOX:  bash profiles/run_half_trace.sh <tag>
set -u
TAG=${1:-r6_half}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python profiles/host_trace_half.py --cold 120 --loop 16 > "$OUT/plain_fresh.log" 2>&1
python profiles/host_trace_half.py --cold 120 --loop 0 --device-first 40 > "$OUT/plain_after_device.log" 2>&1
python profiles/host_trace_half.py --fp32 --cold 30 --loop 16 > "$OUT/plain_fp32.log" 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/tr" -o half -- python "$R/profiles/host_trace_half.py" --cold 10 --loop 10 > "$OUT/traced.log" 2>&1
cd "$R"
db=$(find "$OUT/tr" -name "*_results.db" | head -1)
if [ -n "$db" ]; then
  python profiles/timeline.py "$db" "$OUT/timeline_cold.txt" "3,8" > /dev/null
  python profiles/timeline.py "$db" "$OUT/timeline_loop.txt" "-4,-3,-2" > /dev/null
fi
find "$OUT/tr" -name "*.db" -delete
tail -3 "$OUT/plain_fresh.log" | cut -c1-1500
