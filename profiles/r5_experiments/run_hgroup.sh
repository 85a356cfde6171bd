#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_hgroup; mkdir -p $O
run() {
    local name=$1; shift; local args=$1; shift
    env "$@" timeout 300 python bench.py --device-only $args > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), d["scan_split"]["status"])' 2>&1)"
}
S="--steps 30 --warmup 8"
run base_a "$S" X=1
for v in hg256 hg288; do run $v "$S" MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1; done
run base_b "$S" X=1
for v in hg256 hg288; do run ${v}_b "$S" MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1; done
run base_half "$S --half" X=1
for v in hg256 hg288; do run ${v}_half "$S --half" MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1; done
