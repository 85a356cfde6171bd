#!/bin/bash
# Builds the schedule variants of k_rec_roll (rec_roll.hpp knobs) as separate libraries under medaka_amd/variants/
# (git-ignored *.so; they travel to the GPU box with the tree).  Usage: build_roll_variants.sh
set -e
cd "$(dirname "$0")/../.."
build() {   # name flags...
    local name=$1; shift
    MDK_LIB_OUT=$PWD/medaka_amd/variants/lib_$name.so python medaka_amd/build.py "$@" > /dev/null
    echo "built $name: $*"
}
build tv2_pre2 -DMDK_ROLL_TV=2 -DMDK_ROLL_TPRE=2 &
build tv1_pre0 -DMDK_ROLL_TV=1 -DMDK_ROLL_TPRE=0 &
build tv3_pre0 -DMDK_ROLL_TV=3 -DMDK_ROLL_TPRE=0 &
wait
build tv2_pre4 -DMDK_ROLL_TV=2 -DMDK_ROLL_TPRE=4 &
build tv0_pre18 -DMDK_ROLL_TV=0 -DMDK_ROLL_TPRE=18 &
build tpost -DMDK_ROLL_TPOST=1 &
wait
