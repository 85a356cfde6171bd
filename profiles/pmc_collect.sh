#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (never combined with tracing),
# as MI355X_MICROARCH.md prescribes.  Run ON THE GPU BOX from the repo root:
#     bash profiles/pmc_collect.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
R=$PWD
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for PASS in \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PASS --output-format csv -d "$R/$OUT/pass$i" -o pmc -- \
      python "$R/bench.py" --steps 1 --warmup 0 --cpu-budget 0 "$@" > "$R/$OUT/pass$i.log" 2>&1
  echo "pass $i ($PASS) rc=$?"
done
cd "$R"
find "$OUT" -name "*counter_collection.csv" | head -20
