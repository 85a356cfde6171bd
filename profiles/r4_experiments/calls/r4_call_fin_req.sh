#!/bin/bash
# HEAD = 2: under which step of the strip to request the other direction's partial logits (device-resident, 200 x 10000)
export MDK_SKIP_BUILD=1
for v in ship req2 req6 ship combine; do
  unset MDK_LIB MDK_FINAL_HEAD
  case $v in req*) export MDK_LIB=$PWD/_variants/lib_$v.so;; combine) export MDK_FINAL_HEAD=0;; esac
  timeout 300 python bench.py --device-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,1), 'M col/s', round(d['ms_per_step'],3), 'ms rec', round(d['rec_ms_per_step'],3), 'head', round(d['head_ms_per_step'],3))"
done
