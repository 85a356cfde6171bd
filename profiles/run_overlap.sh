# layer-1 projection under the tail of layer 0's recurrence: off / on
python - <<'PY'
import sys, json, subprocess
for ov in (0, 1):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--cpu-sample", "0", "--overlap", str(ov)],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    r = json.loads(out)
    print("overlap", ov, round(r["value"] / 1e6, 1), "M col/s", round(r["ms_per_step"], 2), "ms", r["roofline"]["kernel_ms_per_step"])
PY
