#!/bin/bash
# read-level front end, round-3 rewrite (table conv1 + software-pipelined conv2) against the round-2 build
set -u
R=$PWD; OUT=$R/gpurun_out/r3_front; mkdir -p "$OUT"
OLD=$R/profiles/r3_experiments/libmedaka_amd_before_front.so
timeout 200 python profiles/r3_experiments/front_compare.py dump "$OUT/new.npz" 2>&1 | tail -1
MDK_LIB=$OLD MDK_SKIP_BUILD=1 timeout 200 python profiles/r3_experiments/front_compare.py dump "$OUT/old.npz" 2>&1 | tail -1
python profiles/r3_experiments/front_compare.py check "$OUT/old.npz" "$OUT/new.npz" | tee "$OUT/compare.txt"
rm -f "$OUT/old.npz" "$OUT/new.npz"
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "read_level or wide or swap" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
for M in rl384 rl128; do
  MDK_LIB=$OLD MDK_SKIP_BUILD=1 timeout 200 python bench.py --model $M --steps 3 --warmup 1 --cpu-budget 0 2>/dev/null | tail -1 > "$OUT/bench_${M}_old.json"
  timeout 300 python bench.py --model $M --steps 3 --warmup 1 --cpu-budget 40 2> "$OUT/bench_${M}.log" | tail -1 > "$OUT/bench_${M}.json"
  MDK_LIB=$OLD MDK_SKIP_BUILD=1 timeout 200 python bench.py --model $M --half --steps 3 --warmup 1 --cpu-budget 0 2>/dev/null | tail -1 > "$OUT/bench_${M}_half_old.json"
  timeout 200 python bench.py --model $M --half --steps 3 --warmup 1 --cpu-budget 0 2>/dev/null | tail -1 > "$OUT/bench_${M}_half.json"
done
python - "$OUT" <<'PY'
import glob, json, os, sys
for p in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        r = json.loads(open(p).read().strip().splitlines()[-1])
        k = r["roofline"]
        print(f"{os.path.basename(p):28s} {r['value']/1e6:7.2f} M positions/s  {r['ms_per_step']:7.2f} ms/batch  front {k['avg_launch_ms']:6.2f} ms  frac {k['frac']:.3f}")
    except Exception as e:
        print(os.path.basename(p), "no result", e)
PY
