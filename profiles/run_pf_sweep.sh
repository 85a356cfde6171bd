for pf in 5 9 13; do
  if [ $pf = 5 ]; then L=$PWD/medaka_amd/libmedaka_amd.so; else L=$PWD/medaka_amd/libmedaka_amd_pf$pf.so; fi
  for b in 200 1000 2000; do
    MDK_LIB=$L MDK_SKIP_BUILD=1 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --batch $b 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('pf', $pf, 'B', $b, round(r['value']/1e6,1), 'Mcol/s', 'rec_ms', round(r['roofline']['avg_launch_ms'],2), {k: round(v,2) for k,v in r['roofline']['kernel_ms_per_step'].items()})"
  done
done
