#!/bin/bash
# item 5a: what does doubling the per-lane gate math cost at a fixed MFMA count?  (half precision, unfused kernels:
# 8-window work-groups = 2 windows per lane, 16-window ones = 4, both 12 MFMAs per wave and step)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c17; mkdir -p $O
run() { name=$1; shift; MDK_FUSE_PROJ=0 timeout 200 python bench.py --device-only --steps 10 --warmup 3 "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s=d["scan_split"]; vw=s["chunks"]*d["batch_windows"]; steps=s["columns"]
print(f"{sys.argv[2]:28s} {vw} windows x {steps} steps: rec {d['rec_ms_per_step']:.3f} ms/forward (2 layers), gi {d['gi_ms_per_step']:.3f}, total {d['ms_per_step']:.3f}; rec per (window, step, layer): {1e6*d['rec_ms_per_step']/(2*vw*steps):.4f} ns")
PY
}
run half_tile8_S5   --half --tile 8
run half_tile16_S10 --half --tile 16 --scan-split 10
run half_tile16_S5  --half --tile 16
run fp32_tile8_S5   --tile 8
