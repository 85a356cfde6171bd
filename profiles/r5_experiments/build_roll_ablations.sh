#!/bin/bash
# Timing-only ablations of k_rec_roll (wrong results): which block costs what.  Libraries under medaka_amd/variants/.
set -e
cd "$(dirname "$0")/../.."
build() { local name=$1; shift; MDK_LIB_OUT=$PWD/medaka_amd/variants/lib_$name.so python medaka_amd/build.py "$@" > /dev/null; echo "built $name: $*"; }
build abl_noh -DROLL_DBG_NOH &
build abl_not -DROLL_DBG_NOT &
build abl_noht -DROLL_DBG_NOH -DROLL_DBG_NOT &
wait
build abl_nobload -DROLL_DBG_NOBLOAD &
build abl_noht_nobload -DROLL_DBG_NOH -DROLL_DBG_NOT -DROLL_DBG_NOBLOAD &
build abl_nopiece -DROLL_DBG_NOPIECE &
wait
