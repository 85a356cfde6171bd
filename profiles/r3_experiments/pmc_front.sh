#!/bin/bash
# SQ counters of k_rl_front (LDS vs matrix pipe): bench.py --model rl128 --steps 1, front_ws 0 and 1
set -u
R=$PWD; OUT=$R/gpurun_out/r3_front_pmc; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for WS in 0 1; do
  i=0
  for PASS in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
              "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
              "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/ws$WS/pass$i" -o pmc -- \
        python "$R/bench.py" --model rl128 --front-ws $WS --steps 1 --warmup 0 --cpu-budget 0 > "$OUT/ws${WS}_pass$i.log" 2>&1
  done
  python "$R/profiles/pmc_step.py" "$OUT/ws$WS" "$OUT/ws$WS.csv" > /dev/null
  grep "k_rl_front" "$OUT/ws$WS.csv" | sed "s/^/ws=$WS /"
  rm -rf "$OUT/ws$WS"
done
