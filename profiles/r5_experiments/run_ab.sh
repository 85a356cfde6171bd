#!/bin/bash
# A/B of the working tree's library against medaka_amd/variants/lib_old.so (built from HEAD): bitwise test of the fused
# kernel first, then device-resident lines old / new / old / new at 200 x 10000 (split) and once each at 1000 x 10000.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r5_ab}; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fused_projection or three_layer or half_precision_mode" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -2 $O/tests.log
run() {
    local name=$1; shift; local args=$1; shift
    env "$@" timeout 300 python bench.py --device-only $args > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), d["scan_split"]["status"])' 2>&1)"
}
OLD="MDK_LIB=$PWD/medaka_amd/variants/lib_old.so MDK_SKIP_BUILD=1"
STD="--steps 30 --warmup 8"; BIG="--batch 1000 --scan-split 0 --steps 6 --warmup 2"
run old_a "$STD" $OLD; run new_a "$STD" X=1; run old_b "$STD" $OLD; run new_b "$STD" X=1
run old_big "$BIG" $OLD; run new_big "$BIG" X=1
run old_half "$STD --half" $OLD; run new_half "$STD --half" X=1
