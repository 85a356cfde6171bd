#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_overlap; mkdir -p $O
run() {
    local name=$1; shift; local args=$1; shift
    env "$@" timeout 300 python bench.py --device-only $args > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), " gi", round(d["gi_ms_per_step"],3), d["scan_split"]["status"])' 2>&1)"
}
S="--steps 30 --warmup 8"
run ov1_a "$S --overlap 1" X=1; run ov0_a "$S --overlap 0" X=1; run ov1_b "$S --overlap 1" X=1; run ov0_b "$S --overlap 0" X=1
run ov1_half "$S --overlap 1 --half" X=1; run ov0_half "$S --overlap 0 --half" X=1
