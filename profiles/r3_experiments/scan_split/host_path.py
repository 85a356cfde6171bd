"""Host-to-host time of a split call with the time-slab streaming of the host path on and off."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd.engine import GruEngine, PinnedArray

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")


def host_time(eng, xin, out, reps=8):
    for _ in range(2):
        eng.forward_host(xin, out=out)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward_host(xin, out=out)
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    w = np.load(os.path.join(GOLD, "weights_init.npz"))
    T = 10000
    for half in (False, True):
        for B in (200, 100):
            eng = GruEngine({k: w[k] for k in w.files})
            eng.set_precision(half)
            xin = PinnedArray((B, T, 10), np.float32).array
            xin[...] = np.random.default_rng(0).random((B, T, 10), dtype=np.float32)
            out = PinnedArray((B, T, 5), np.float32).array
            for split, margin in ((0, 256), (1, 256), (1, 128)):
                eng.set_option("scan_split", split)
                eng.set_option("scan_split_margin", margin)
                ms = host_time(eng, xin, out)
                print(f"{'half' if half else 'fp32'} B {B} scan_split {split} margin {margin}: {ms:6.2f} ms  {B * T / ms / 1e3:6.1f} M columns/s  {eng.split()}", flush=True)
            eng.close()


if __name__ == "__main__":
    main()
