"""Random shapes through the split scan against the sequential scan of the same engine (host entry and device entry)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd import synth
from medaka_amd.engine import GruEngine, split_plan

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(2026)
    engines = {}
    for name in ("weights_trained", "weights_init"):
        w = np.load(os.path.join(GOLD, name + ".npz"))
        engines[name] = GruEngine({k: w[k] for k in w.files})
    worst = {False: 0.0, True: 0.0}
    stats = {}
    for i in range(n):
        name = ("weights_trained", "weights_init")[i % 2]
        half = bool(rng.integers(2))
        B = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 31, 50, 100, 128, 200, 255, 341, 400]))
        T = int(rng.choice([1024, 1025, 1500, 2047, 2048, 3000, 4097, 5000, 7777, 9999, 10000, 12000]))
        if B * T > 2_200_000:
            B = max(1, 2_200_000 // T)
        e = engines[name]
        e.set_precision(half)
        x = synth.counts_windows(B, T, depth=int(rng.integers(5, 80)), seed=int(rng.integers(1 << 30)))
        e.set_option("scan_split", 0)
        seq = e.forward_host(x)
        e.set_option("scan_split", 1)
        e.set_option("gpu_share", int(rng.choice([1, 1, 1, 2, 3])))
        out = e.forward_host(x)
        info = e.split()
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty(B, T, 5, device="cuda")
        e.forward_ptr(xd.data_ptr(), B, T, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        same_entry = np.array_equal(yd.cpu().numpy(), out)
        e.set_option("gpu_share", 1)
        d = float(np.abs(out - seq).max())
        stats[info["status"]] = stats.get(info["status"], 0) + 1
        ok = np.isfinite(out).all() and same_entry and d <= (4e-4 if half else 4e-6) and (info["chunks"] == 1) == (info["status"] in ("not used", "disabled"))
        if info["chunks"] > 1:
            worst[half] = max(worst[half], d)
        print(f"{i:3d} {name[8:]:8s} {'half' if half else 'fp32'} {B:4d} x {T:5d}: {info['chunks']:2d} chunks, margin {info['margin']:3d}, {info['status']:9s} "
              f"junction {info['max_delta']:.1e}  max|dp| vs sequential {d:.1e}  device entry identical {same_entry}  {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            raise SystemExit(1)
    print("statuses", stats, "worst max|dp| of split calls: fp32", worst[False], "half", worst[True])


if __name__ == "__main__":
    main()
