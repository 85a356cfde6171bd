#!/bin/bash
# On the GPU box: correctness of k_rec_roll first (bit-identical to k_rec_fused), then the device-resident bench line of
# every schedule variant built by build_roll_variants.sh, against the alternating kernel (MDK_ROLL=0) in the same process order.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_roll; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rolled or fused_projection" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
run() {   # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --device-only --steps 30 --warmup 8 > $O/$name.json 2> $O/$name.err
    echo "$name: $(cat $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms  L0", round(d["rec_l0_ms"],3), " L1", round(d["rec_l1_ms"],3), " rolled", d["rolled"], d["scan_split"]["status"])' 2>&1)"
}
run alt_a MDK_ROLL=0
run roll_default MDK_ROLL=1
for v in tv1_pre0 tv3_pre0 tv2_pre4 tv0_pre18 tpost; do
    run roll_$v MDK_ROLL=1 MDK_LIB=$PWD/medaka_amd/variants/lib_$v.so MDK_SKIP_BUILD=1
done
run alt_b MDK_ROLL=0
run roll_default_b MDK_ROLL=1
