#!/bin/bash
# Layer-0 recurrence tuning knobs, rebuilt into side libraries (_variants/lib_*.so) and timed
# device-resident at BASELINE configs[1]; the shipped library first and last as the control.
mkdir -p gpurun_out/variants_out
export MDK_SKIP_BUILD=1
for v in ship pf3 pf4 pf7 prio0 ship2; do
  if [ "$v" = ship ] || [ "$v" = ship2 ]; then unset MDK_LIB; else export MDK_LIB=$PWD/_variants/lib_$v.so; fi
  timeout 300 python bench.py --device-only --steps 20 --warmup 5 > gpurun_out/variants_out/$v.json 2> gpurun_out/variants_out/$v.err
  echo "$v rc=$? $(python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/variants_out/$v.json").read().strip().splitlines()[-1])
    print(round(d["value"]/1e6,1), round(d["ms_per_step"],3), "rec", round(d["rec_ms_per_step"],3), d["scan_split"]["status"])
except Exception as e:
    print("parse", e)
PY
)"
done
