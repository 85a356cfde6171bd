#!/bin/bash
# 1000 x 10000 without the split scan (250 work-groups, no certificate to trip over the ablations' garbage): layer-1 time per form
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_roll_abl; mkdir -p $O
run() {
    local name=$1; shift
    env "$@" timeout 300 python bench.py --device-only --batch 1000 --scan-split 0 --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), " rolled", d["rolled"])' 2>&1)"
}
run alt MDK_ROLL=0
run roll MDK_ROLL=1
for v in abl_noh abl_not abl_noht abl_nobload abl_noht_nobload abl_nopiece; do
    run $v MDK_ROLL=1 MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1
done
run alt_b MDK_ROLL=0
