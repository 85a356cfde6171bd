"""Determinism soak of the LSTM(384) cluster exchange: repeated forwards must return identical bits
(a lost or stale granule would show up as a changed result or a time-out error).
    python profiles/soak_wide.py [repeats] [--compete]
--compete: a second PROCESS keeps 128 CUs busy with a compute-bound kernel (mdk_selftest_burn, 128 work-groups
of FMA loops) while the forwards run: late cluster members are then the rule; a time-out must end in the
engine's retry on the plain schedule (counted in timing()["wide_retries"]), never in an error or a changed bit."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as g  # noqa: E402

g.build()
from medaka_amd import engine  # noqa: E402
from oracle import rl_oracle  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 12
compete = "--compete" in sys.argv
burner = None
if compete:
    import subprocess
    burner = subprocess.Popen([sys.executable, "-c",
                               "import sys, ctypes; sys.path.insert(0, '.'); from medaka_amd import lib; L = lib.load();\n"
                               "import time; t0 = time.time()\n"
                               "while time.time() - t0 < 240: lib.check(L.mdk_selftest_burn(0, 128, 4000000), 'burn')"])
    time.sleep(3)
kw = dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
st = rl_oracle.synth_rl_state(seed=21, **kw)
e = engine.RlEngine(st, **kw)
t0 = time.time()
for B, P, D in ((40, 3000, 3), (300, 1200, 2)):
    x = rl_oracle.synth_reads(B, P, D, use_dwells=True, seed=B)
    for half in (False, True):
        e.set_precision(half)
        for wt in (0, 1):
            e.set_option("wide_write_through", wt)
            ref = e.forward_host(x)
            bad = sum(not np.array_equal(e.forward_host(x), ref) for _ in range(reps))
            print(f"B={B} P={P} half={half} write_through={wt}: {reps} repeats, {bad} differing", flush=True)
            assert bad == 0
e.enable_timing(True)
e.forward_host(x)
print("retries on the plain schedule so far:", e.timing()["wide_retries"])
if burner is not None:
    burner.terminate()
    burner.wait()
print(f"soak ok in {time.time() - t0:.0f}s")
