"""Goldens for the read-level model from the UNMODIFIED reference `LatentSpaceLSTM`
(medaka/architectures/latent_space_lstm.py) on PyTorch-CPU.  Build container only.

    python oracle/make_golden_rl.py   ->  tests/golden/rl_*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, rl_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WIDE_SEED = 21
# name -> (windows, positions, padded depth, input seed): "two_groups" = one full + one ragged
# 8-window group; "many_groups" = 17 groups > 16 clusters, so clusters loop over groups
WIDE_CASES = {"two_groups": (11, 150, 7, 31), "many_groups": (130, 24, 3, 32)}
# rl_lstm384 WITHOUT dwells: 4 of the 8 bundled read-level models (reference options.py:175-182) take the
# `use_dwells=False` branch of latent_space_lstm.py:176-190 (4 features per read, 7 conv input channels)
WIDE_ND_SEED = 22
WIDE_ND_CASES = {"two_groups": (11, 150, 7, 41), "many_groups": (130, 24, 3, 42), "long": (3, 1100, 5, 43)}


def main():
    arch, models, te = ref_shim.reference_modules()
    out = {}
    for name, kw in (("bi", dict()), ("uni", dict(bidirectional=False)), ("bi_dwells", dict(use_dwells=True))):
        torch.manual_seed(7)
        m = arch.LatentSpaceLSTM(**kw).eval()
        # non-trivial batch-norm statistics and larger recurrent weights (default init is tame)
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if k.endswith("running_mean"):
                    v.copy_(torch.randn_like(v) * 0.3)
                elif k.endswith("running_var"):
                    v.copy_(torch.rand_like(v) + 0.5)
                elif "convs.2.weight" in k or "convs.5.weight" in k:
                    v.copy_(torch.rand_like(v) + 0.5)
                elif "lstm" in k and "weight" in k:
                    v.mul_(2.0)
        state = {k: v.numpy().copy() for k, v in m.state_dict().items() if "num_batches_tracked" not in k}
        np.savez(os.path.join(GOLD, f"rl_weights_{name}.npz"), **state)
        x = rl_oracle.synth_reads(3, 120, 9, use_dwells=kw.get("use_dwells", False), seed=11)
        y = m.predict_on_batch(te.Batch(read_level_features=torch.from_numpy(x))).numpy()
        out[f"{name}/x"] = x
        out[f"{name}/y"] = y
        print(name, y.shape, float(np.abs(rl_oracle.rl_forward(x, state, **{k: v for k, v in kw.items()}) - y).max()))
    np.savez_compressed(os.path.join(GOLD, "rl_cases.npz"), **out)

    # rl_lstm384 architecture (BASELINE config 4b): weights regenerated from a seed by the tests
    wide = {}
    kw = dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
    state = rl_oracle.synth_rl_state(seed=WIDE_SEED, **kw)
    m = arch.LatentSpaceLSTM(**kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    for name, (B, P, D, seed) in WIDE_CASES.items():
        x = rl_oracle.synth_reads(B, P, D, use_dwells=True, seed=seed)
        y = m.predict_on_batch(te.Batch(read_level_features=torch.from_numpy(x))).numpy()
        wide[f"{name}/x"] = x
        wide[f"{name}/y"] = y
        print("wide", name, y.shape, float(np.abs(rl_oracle.rl_forward(x, state, use_dwells=True, bidirectional=False) - y).max()))
    np.savez_compressed(os.path.join(GOLD, "rl_wide_cases.npz"), **wide)


def main_wide_no_dwells():
    arch, models, te = ref_shim.reference_modules()
    kw = dict(lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False)
    state = rl_oracle.synth_rl_state(seed=WIDE_ND_SEED, **kw)
    m = arch.LatentSpaceLSTM(**kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    out = {}
    for name, (B, P, D, seed) in WIDE_ND_CASES.items():
        x = rl_oracle.synth_reads(B, P, D, use_dwells=False, seed=seed)
        y = m.predict_on_batch(te.Batch(read_level_features=torch.from_numpy(x))).numpy()
        out[f"{name}/x"] = x
        out[f"{name}/y"] = y
        print("wide no-dwells", name, y.shape,
              float(np.abs(rl_oracle.rl_forward(x, state, use_dwells=False, bidirectional=False) - y).max()))
    np.savez_compressed(os.path.join(GOLD, "rl_wide_nd_cases.npz"), **out)


if __name__ == "__main__":
    if "--wide-no-dwells" in sys.argv:       # added in round 3; leaves the earlier fixtures untouched
        main_wide_no_dwells()
    else:
        main()
