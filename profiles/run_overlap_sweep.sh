# where does the side-stream overlap stop paying?  (overlap_gemm: 0 off, 2 forced)
python - <<'PY'
import sys, json, subprocess
for b in (256, 320, 400, 512, 640, 800):
    row = []
    for ov in (0, 2):
        out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--batch", str(b), "--overlap", str(ov)],
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        r = json.loads(out)
        row.append(round(r["value"] / 1e6, 1))
    print("B", b, "off", row[0], "forced", row[1], "M col/s", flush=True)
PY
