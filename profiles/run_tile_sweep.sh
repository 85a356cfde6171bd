# 4- vs 8-window recurrence work-groups at medium batches (overlap auto)
python - <<'PY'
import sys, json, subprocess
for b in (320, 400, 512, 640):
    row = []
    for tile in (4, 8):
        out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--batch", str(b), "--tile", str(tile)],
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        r = json.loads(out)
        row.append(round(r["value"] / 1e6, 1))
    print("B", b, "4-window", row[0], "8-window", row[1], "M col/s", flush=True)
PY
