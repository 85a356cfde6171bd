#!/bin/bash
# one vs two 8-window tiles per recurrence work-group (rec_mfma2.hpp): bitwise test, then device-resident timings
set -u
R=$PWD; OUT=$R/gpurun_out/r3_sets; mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "two_set or tile_sizes or schedule_variants" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for B in 1000 2000 4000; do for SETS in 1 2; do
  timeout 300 python bench.py --device-only --batch $B --steps 3 --warmup 1 --rec-sets $SETS 2>/dev/null | tail -1 | tee "$OUT/b${B}_sets${SETS}.json"
done; done
