#!/bin/bash
# Non-temporal streaming of the layer-1 kernel's activations in / h out (MDK_NT_STREAM bits) against the default policy:
# 1000 x 10000 unsplit, and the headline 200 x 10000 split.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_nt; mkdir -p $O
run() {
    local name=$1; shift; local args=$1; shift
    env "$@" timeout 300 python bench.py --device-only $args > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), d["scan_split"]["status"])' 2>&1)"
}
BIG="--batch 1000 --scan-split 0 --steps 6 --warmup 2"
STD="--steps 30 --warmup 8"
run big_base "$BIG" MDK_ROLL=0
for v in nt1 nt2 nt3; do run big_$v "$BIG" MDK_ROLL=0 MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1; done
run std_base "$STD" MDK_ROLL=0
for v in nt1 nt2 nt3; do run std_$v "$STD" MDK_ROLL=0 MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1; done
run std_base_b "$STD" MDK_ROLL=0
