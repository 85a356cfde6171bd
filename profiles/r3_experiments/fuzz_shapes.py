"""One-off randomized shape fuzz on the device: MFMA kernels vs the exact fp32 kernels (GRU, both host and device entry,
both precisions for finiteness) and the read-level engines vs the CPU oracle on small random shapes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from medaka_amd import engine, synth
from oracle import rl_oracle

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
e, ex = engine.GruEngine(st), engine.GruEngine(st)
ex.set_variant(True)
worst = 0.0
for i in range(40):
    B = int(rng.choice([1, 2, 3, 5, 8, 9, 17, 33, 64, 100]))
    T = int(rng.choice([1, 2, 7, 15, 16, 17, 100, 1000, 2047, 2048, 2064, 2500, 4096]))
    x = synth.counts_windows(B, T, seed=int(rng.integers(1 << 30)))
    e.set_option("overlap_gemm", int(rng.integers(0, 2))); e.set_option("stream_host", int(rng.integers(0, 2)))
    e.set_option("rec_windows_per_tile", int(rng.choice([0, 4, 8])))
    a, b = e.forward_host(x), ex.forward_host(x)
    d = float(np.abs(a - b).max())
    worst = max(worst, d)
    assert d <= 2e-5 and np.isfinite(a).all(), (B, T, d)
    e.set_precision(True); h = e.forward_host(x); e.set_precision(False)
    assert np.isfinite(h).all() and np.abs(h - b).max() < 5e-3, (B, T, "half", float(np.abs(h - b).max()))
print(f"GRU: 40 random shapes, MFMA vs exact kernels: worst max|dp| {worst:.2e}")
for name, kw, dw in (("rl128", dict(), False), ("rl384", dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False), True),
                     ("rl384nd", dict(lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False), False)):
    stt = synth.synth_rl_state(seed=9, **{**dict(lstm_size=128, cnn_size=128, use_dwells=False, bidirectional=True), **kw})
    er = engine.RlEngine(stt, **kw)
    worst = 0.0
    for i in range(8):
        B, P, D = int(rng.integers(1, 20)), int(rng.choice([1, 5, 95, 96, 97, 191, 193, 400])), int(rng.integers(1, 12))
        x = synth.synth_reads(B, P, D, use_dwells=dw, seed=int(rng.integers(1 << 30)), empty_tail=False)
        ref = rl_oracle.rl_forward(x, stt, use_dwells=dw, bidirectional=kw.get("bidirectional", True))
        out = er.forward_host(x)
        d = float(np.abs(out - ref).max())
        worst = max(worst, d)
        assert d <= 2e-5, (name, B, P, D, d)
    er.close()
    print(f"{name}: 8 random shapes vs the CPU oracle: worst max|dp| {worst:.2e}")
