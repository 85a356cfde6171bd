# positions per front-end work-group = 16 * MT (libraries built with -DMDK_RL_MT=<mt>)
for mt in 4 6 7 8; do
  for mode in "" "--half"; do
    MDK_LIB=$PWD/medaka_amd/libmedaka_amd_mt$mt.so MDK_SKIP_BUILD=1 python profiles/bench_rl.py 100 10000 50 $mode 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('MT', $mt, '$mode', 'ms', round(r['ms'],2), 'err', r['max_abs_dp_vs_oracle'])"
  done
done
