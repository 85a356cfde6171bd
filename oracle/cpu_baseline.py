"""One configuration of the host-CPU baseline, run as a child process of
bench.py so that a pathological thread count (OpenMP barriers of 256 spinning threads on a CPU-limited
container take minutes per pass) can be killed on a wall-clock limit instead of stalling the bench.

    python oracle/cpu_baseline.py --weights W.npz --input X.npy --batch B --threads N --out OUT.npz

Times `predict_on_batch` of the reference model on PyTorch-CPU fp32: the unmodified reference class
(medaka.architectures.GRUModel, models.py:303-313, gru.py:58-72) when /root/reference is present,
its three-call restatement (oracle.make_torch_oracle) otherwise.  1 short warm-up, up to 3 timed passes;
progress is printed as JSON lines after every pass so that the parent keeps what finished before a kill.
Test infrastructure: only bench.py's `cpu_baseline` leg executes this file.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", required=True)
    ap.add_argument("--input", required=True)
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--threads", type=int, required=True)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = str(a.threads)
    import numpy as np
    import torch
    from oracle import oracle, ref_shim
    torch.set_num_threads(a.threads)
    state = dict(np.load(a.weights))
    x = np.load(a.input, mmap_mode="r")[:a.batch]
    x = np.ascontiguousarray(x)
    kind = "port"
    if ref_shim.available():
        arch, _, te = ref_shim.reference_modules()
        m = arch.GRUModel(num_features=10, num_classes=5, gru_size=128).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        predict = lambda v: m.predict_on_batch(te.Batch(counts_matrix=torch.from_numpy(v)))
        kind = "reference"
    else:
        predict = oracle.make_torch_oracle(state).predict
    predict(x[:1, :500])                 # warm-up: thread pool, oneDNN primitives
    cols = x.shape[0] * x.shape[1]
    out = None
    for i in range(a.passes):
        t0 = time.perf_counter()
        out = predict(x)
        dt = time.perf_counter() - t0
        print(json.dumps({"pass": i, "seconds": dt, "columns_per_s": cols / dt, "kind": kind,
                          "batch": a.batch, "threads": a.threads}), flush=True)
        if i == 0 and a.out:
            np.save(a.out, out.numpy() if hasattr(out, "numpy") else np.asarray(out))


if __name__ == "__main__":
    main()
