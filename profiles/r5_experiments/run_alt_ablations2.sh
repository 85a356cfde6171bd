#!/bin/bash
# k_rec_fused staging, taken apart (timing only): requests alone / + conversion / + LDS stores
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_alt_abl2; mkdir -p $O
run() {
    local name=$1; shift
    env "$@" timeout 300 python bench.py --device-only --batch 1000 --scan-split 0 --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3))' 2>&1)"
}
run alt MDK_ROLL=0
for v in alt_nopiece alt_nopstore alt_noldsw; do run $v MDK_ROLL=0 MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1; done
run alt_b MDK_ROLL=0
