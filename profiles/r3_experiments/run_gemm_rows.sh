#!/bin/bash
# 64- vs 128-row work-groups of the layer-1 projection GEMM (gi_proj.hpp MT): bitwise test, then device-resident timings
set -u
R=$PWD; OUT=$R/gpurun_out/r3_gemm; mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "gemm_row or overlapped or multi_pass" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
for B in 200 1000; do for ROWS in 64 128; do
  timeout 200 python bench.py --device-only --batch $B --steps 4 --warmup 2 --gemm-rows $ROWS 2>/dev/null | tail -1 | tee "$OUT/b${B}_rows${ROWS}.json"
done; done
timeout 200 python bench.py --device-only --batch 1000 --steps 4 --warmup 2 --gemm-rows 128 --overlap 0 2>/dev/null | tail -1 | tee "$OUT/b1000_rows128_nooverlap.json"
