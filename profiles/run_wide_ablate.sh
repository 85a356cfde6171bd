# timing-only ablations of k_lstm_wide (library built with -DMDK_WIDE_ABLATE); results are garbage
# usage: bash profiles/run_wide_ablate.sh [--half]
for a in 0 1 2 8 9 4; do
  MDK_WIDE_ABL=$a MDK_LIB=$PWD/medaka_amd/libmedaka_amd_abl.so MDK_SKIP_BUILD=1 python profiles/bench_rl.py 100 2000 4 --wide $1 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('abl', $a, 'ms', round(r['ms'],2))"
done
