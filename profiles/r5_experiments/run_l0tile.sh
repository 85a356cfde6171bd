#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_l0tile; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_scan_split_gpu.py -x -q -m gpu -k "half" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -2 $O/tests.log
run() {
    local name=$1; shift; local args=$1; shift
    env "$@" timeout 300 python bench.py --device-only $args > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), d["scan_split"]["status"])' 2>&1)"
}
H="--half --steps 30 --warmup 8"
run t8_a "$H" MDK_L0_TILE16=0; run t16_a "$H" MDK_L0_TILE16=1; run t8_b "$H" MDK_L0_TILE16=0; run t16_b "$H" MDK_L0_TILE16=1
run t8_B100 "$H --batch 100" MDK_L0_TILE16=0; run t16_B100 "$H --batch 100" MDK_L0_TILE16=1
