"""Would the split scan carry over to the read-level models (open item 0)?  CPU-only premise check with
the restated reference arithmetic (oracle/rl_oracle.py): a sub-window with a margin of G positions on either side against
the full window, on the sub-window's own positions.  (The front end is per-position apart from the 17-tap conv's halo of
8 positions, so the margin serves both the conv halo and the LSTM stack's warm-up.)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from medaka_amd import synth
from oracle import rl_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")


def main():
    P, a, b = 1500, 600, 900
    for name, bi, dwells in (("rl_weights_trained", True, False), ("rl_weights_bi", True, False), ("rl_weights_uni", False, False),
                             ("rl_weights_bi_dwells", True, True)):
        st = dict(np.load(os.path.join(GOLD, name + ".npz")))
        x = synth.synth_reads(3, P, 12, use_dwells=dwells, seed=5)
        full = rl_oracle.rl_forward(x, st, use_dwells=dwells, bidirectional=bi)
        for G in (16, 32, 64, 128, 256):
            sub = rl_oracle.rl_forward(np.ascontiguousarray(x[:, a - G:b + G]), st, use_dwells=dwells, bidirectional=bi)[:, G:G + (b - a)]
            d = float(np.abs(sub - full[:, a:b]).max())
            print(f"{name:22s} margin {G:4d}: max|dp| on the sub-window's own positions {d:.2e}", flush=True)
            if d < 1e-7:
                break
    # the bundled rl_lstm384 architecture (4 alternating uni-directional LSTM(384)), weights from a seed
    st = synth.synth_rl_state(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False, seed=0)
    x = synth.synth_reads(2, 900, 8, use_dwells=True, seed=6)
    a, b = 350, 550
    full = rl_oracle.rl_forward(x, st, use_dwells=True, bidirectional=False)
    for G in (32, 64, 128, 256):
        sub = rl_oracle.rl_forward(np.ascontiguousarray(x[:, a - G:b + G]), st, use_dwells=True, bidirectional=False)[:, G:G + (b - a)]
        print(f"{'rl_lstm384 (seeded)':22s} margin {G:4d}: max|dp| on the sub-window's own positions {float(np.abs(sub - full[:, a:b]).max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
