#!/bin/bash
# where does a step of the two-set recurrence go?  SQ counters of bench.py --device-only at B = 1000 / 2000, rec_sets 1 / 2
set -u
R=$PWD; OUT=$R/gpurun_out/r3_sets_pmc; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for B in 1000 2000; do for SETS in 1 2; do
  i=0
  for PASS in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
              "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/b${B}_s${SETS}/pass$i" -o pmc -- \
        python "$R/bench.py" --device-only --batch $B --steps 1 --warmup 0 --rec-sets $SETS > /dev/null 2>&1
  done
  python "$R/profiles/pmc_step.py" "$OUT/b${B}_s${SETS}" "$OUT/b${B}_s${SETS}.csv" > /dev/null
  grep "k_rec_mfma" "$OUT/b${B}_s${SETS}.csv" | grep -v "fallback" | sed "s/^/B=$B sets=$SETS /"
  rm -rf "$OUT/b${B}_s${SETS}"
done; done
cd "$R"
for B in 1000 2000; do timeout 200 python bench.py --device-only --batch $B --steps 3 --warmup 1 --rec-sets 2 2>/dev/null | tail -1; done
