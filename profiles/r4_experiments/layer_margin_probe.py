"""Which layer needs the margin?  Largest junction difference per (layer, direction, certificate point) at forced margins
(MDK_SPLIT_DEBUG output of the engine), 200 x 10000 i.i.d. pileups: the data behind "a narrower layer-1 window" (review
item 5b: layer 1 would get half the margin)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g; g.build()
from medaka_amd import engine, synth
name, margin = sys.argv[1], int(sys.argv[2])
if name == "trained":
    st = dict(np.load(%r + "/tests/golden/weights_trained.npz"))
else:
    z = np.load(%r + "/tests/golden/weights_zoo.npz")
    st = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
x = np.concatenate([synth.counts_windows(8, 10000, depth=50, seed=100 + s) for s in range(25)])
e = engine.GruEngine(st)
e.set_option("scan_split_audit", 0)
e.set_option("scan_split_margin", margin)
e.set_option("scan_split", 5)
e.forward_host(x)
''' % (ROOT, ROOT, ROOT)
for name in ("trained", "maj1", "depthmix", "hp"):
    for margin in (32, 64, 96, 128, 192):
        r = subprocess.run([sys.executable, "-c", CHILD, name, str(margin)], capture_output=True, text=True,
                           env=dict(os.environ, MDK_SPLIT_DEBUG="1"))
        worst = {}
        for m in re.finditer(r"layer (\d) direction (\d) point (\d): ([0-9.e+-]+|inf|nan)", r.stderr):
            key = (int(m.group(1)), int(m.group(3)))
            worst[key] = max(worst.get(key, 0.0), float(m.group(4)))
        print(f"{name:9s} margin {margin:4d}: " + "  ".join(f"L{l} point{p} {v:.1e}" for (l, p), v in sorted(worst.items())), flush=True)
