"""Split scan against the sequential scan of the same engine: differences, certificate, time (host path and
device-resident)."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd import synth
from medaka_amd.engine import GruEngine, PinnedArray

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")


def state(name, scale):
    w = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: (w[k] * scale if ("weight" in k and k.startswith("gru.")) else w[k]).astype(np.float32) for k in w.files}


def dev_time(eng, B, T, reps=8):
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


def host_time(eng, x, reps=6):
    xin = PinnedArray(x.shape, np.float32)          # what the engine's Batch.collate hands over
    xin.array[...] = x
    x = xin.array
    out = PinnedArray(x.shape[:2] + (5,), np.float32).array
    for _ in range(2):
        eng.forward_host(x, out=out)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward_host(x, out=out)
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    for half in (False, True):
        for name, scale in (("weights_init", 1.0), ("weights_trained", 1.0), ("weights_init", 3.0), ("weights_init", 5.0)):
            for B, T in ((8, 6000), (37, 10000), (200, 10000)):
                x = synth.counts_windows(B, T, depth=60, seed=B).astype(np.float32)
                eng = GruEngine(state(name, scale))
                eng.set_precision(half)
                eng.set_option("scan_split", 0)
                ref = eng.forward_host(x)
                eng.set_option("scan_split", 1)
                got = eng.forward_host(x)
                info = eng.split()
                xd = torch.from_numpy(x).cuda()
                od = torch.empty(B, T, 5, device="cuda")
                eng.set_option("scan_split", 1)
                eng.forward_ptr(xd.data_ptr(), B, T, od.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                dev = od.cpu().numpy()
                print(f"{'half' if half else 'fp32'} {name} x{scale:g} {B}x{T}: {info}  max|dp| vs sequential {np.abs(got - ref).max():.2e}  "
                      f"host path == device path: {np.array_equal(got, dev)}", flush=True)
                eng.close()
    w = state("weights_init", 1.0)
    for half in (False, True):
        for B in (200, 100, 10, 1, 128, 256, 300, 340):
            T = 10000
            eng = GruEngine(w)
            eng.set_precision(half)
            x = np.random.default_rng(0).random((B, T, 10), dtype=np.float32)
            row = []
            for opt in (0, 1):
                eng.set_option("scan_split", opt)
                row.append((dev_time(eng, B, T), host_time(eng, x), eng.split()))
            (d0, h0, _), (d1, h1, info) = row
            print(f"{'half' if half else 'fp32'} B {B:4d}: device {d0:6.2f} -> {d1:6.2f} ms (x{d0 / d1:4.2f}, {B * T / d1 / 1e3:6.1f} M columns/s)   "
                  f"host-to-host {h0:6.2f} -> {h1:6.2f} ms (x{h0 / h1:4.2f}, {B * T / h1 / 1e3:6.1f} M columns/s)   {info}", flush=True)
            eng.close()


if __name__ == "__main__":
    main()
