"""Premise check for splitting the scan: device-resident time of the engine on the VIRTUAL batch (B*S windows of
T/S + 2G columns) against the real one (B windows of T columns).  No new engine code: the virtual batch is built with
numpy, so this measures only what the existing kernels do with it."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd import synth
from medaka_amd.engine import GruEngine

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")


def run(eng, B, T, reps=8):
    x = torch.rand(B, T, 10, device="cuda")
    out = torch.empty(B, T, 5, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        eng.forward_ptr(x.data_ptr(), B, T, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward_ptr(x.data_ptr(), B, T, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    w = np.load(os.path.join(GOLD, "weights_init.npz"))
    eng = GruEngine({k: w[k] for k in w.files})
    T = 10000
    for half in (False, True):
        eng.set_precision(half)
        for B in (200, 100, 10, 1000):
            base = run(eng, B, T)
            print(f"{'half' if half else 'fp32'} B {B:4d} T {T}: {base:7.2f} ms  ({B * T / base / 1e3:6.1f} M columns/s)", flush=True)
            for G in (128, 256):
                for S in (2, 3, 4, 5, 6, 8, 10, 16):
                    if B * S > 2400 or T // S < 4 * G:
                        continue
                    Tv = -(-(T // S + 2 * G) // 16) * 16
                    ms = run(eng, B * S, Tv)
                    print(f"      S {S:2d} margin {G}: {B * S:5d} x {Tv:5d}  {ms:7.2f} ms  x{base / ms:4.2f}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
