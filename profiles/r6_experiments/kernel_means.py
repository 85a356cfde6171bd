"""Mean duration per kernel (name, grid) over the last N dispatches of each, from a rocprofv3 kernel-trace CSV directory."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
groups = collections.defaultdict(list)
for r in rows:
    m = re.search(r"k_[a-z_0-9]+(I[A-Za-z0-9_]*E)?", r["Kernel_Name"])
    groups[(m.group(0) if m else r["Kernel_Name"][:40], r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0.0
for k, v in sorted(groups.items()):
    v = v[-last:]
    print(f"{k[0]:48s} grid {k[1]:>7s} x {k[2]:>2s}  n={len(v):4d}  mean {sum(v) / len(v) / 1e3:9.1f} us")
