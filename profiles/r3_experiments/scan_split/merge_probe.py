"""How many columns of warm-up does a window that starts from h = 0 in the middle of a sequence need before the ENGINE's
results are bit-identical to the full scan?  (The premise of splitting the scan: profiles/r3_experiments/README.md.)

For each weight set: probs of the full (8, 6000) call against probs of the sub-window x[:, a-G : b+G] on its core
[a, b), bitwise.  Public API only (GruEngine.forward_host)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd import synth
from medaka_amd.engine import GruEngine

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")


def state(name, scale):
    w = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: (w[k] * scale if ("weight" in k and k.startswith("gru.")) else w[k]).astype(np.float32) for k in w.files}


def main():
    T, a, b = 6000, 2500, 3500
    x = synth.counts_windows(8, T, depth=60, seed=5).astype(np.float32)
    for half in (False, True):
        for name, scale in (("weights_init", 1.0), ("weights_trained", 1.0), ("weights_init", 3.0), ("weights_init", 5.0)):
            eng = GruEngine(state(name, scale))
            eng.set_precision(half)
            full = eng.forward_host(x)
            for G in (32, 64, 128, 256, 512, 1024, 2048):
                sub = eng.forward_host(np.ascontiguousarray(x[:, a - G:b + G]))[:, G:G + (b - a)]
                ref = full[:, a:b]
                bad = sub.view(np.uint32) != ref.view(np.uint32)
                print(f"{'half' if half else 'fp32'} {name} x{scale:g} margin {G:5d}: {int(bad.sum()):7d} of {bad.size} values differ, "
                      f"max |d| {np.abs(sub - ref).max():.2e}", flush=True)
                if not bad.any():
                    break
            eng.close()


if __name__ == "__main__":
    main()
