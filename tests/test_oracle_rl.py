"""Read-level model oracle (oracle/rl_oracle.py) pinned against goldens produced by the UNMODIFIED
reference `LatentSpaceLSTM` (oracle/make_golden_rl.py), and against the reference live when present."""
import os

import numpy as np
import pytest

from conftest import GOLD
from oracle import ref_shim, rl_oracle

CONFIGS = {"bi": dict(), "uni": dict(bidirectional=False), "bi_dwells": dict(use_dwells=True)}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_rl_oracle_matches_reference_goldens(name):
    cases = np.load(os.path.join(GOLD, "rl_cases.npz"))
    state = dict(np.load(os.path.join(GOLD, f"rl_weights_{name}.npz")))
    out = rl_oracle.rl_forward(cases[f"{name}/x"], state, **CONFIGS[name])
    ref = cases[f"{name}/y"]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-6
    assert np.abs(out.sum(-1) - 1).max() <= 1e-5


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_rl_oracle_against_live_reference():
    import torch
    arch, models, te = ref_shim.reference_modules()
    torch.manual_seed(3)
    m = arch.LatentSpaceLSTM(bidirectional=False).eval()
    state = {k: v.numpy() for k, v in m.state_dict().items() if "num_batches" not in k}
    x = rl_oracle.synth_reads(2, 40, 5, seed=5)
    ref = m.predict_on_batch(te.Batch(read_level_features=torch.from_numpy(x))).numpy()
    assert np.abs(rl_oracle.rl_forward(x, state, bidirectional=False) - ref).max() <= 1e-6


@pytest.mark.parametrize("name", ["two_groups", "many_groups"])
def test_rl_oracle_matches_reference_goldens_lstm384(name):
    """rl_lstm384 architecture (LSTM 384, 4 x uni-directional, dwells): weights regenerated from the
    seed the golden script used, outputs from the unmodified reference."""
    from oracle.make_golden_rl import WIDE_SEED
    cases = np.load(os.path.join(GOLD, "rl_wide_cases.npz"))
    state = rl_oracle.synth_rl_state(seed=WIDE_SEED, lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
    out = rl_oracle.rl_forward(cases[f"{name}/x"], state, use_dwells=True, bidirectional=False)
    ref = cases[f"{name}/y"]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-5
    assert (out.argmax(-1) == ref.argmax(-1)).mean() >= 0.999


@pytest.mark.parametrize("name", ["two_groups", "many_groups", "long"])
def test_rl_oracle_matches_reference_goldens_lstm384_no_dwells(name):
    """The `use_dwells=False` flavour of rl_lstm384 (4 of the 8 bundled read-level models, reference
    options.py:175-182; branch latent_space_lstm.py:186-190): outputs from the unmodified reference."""
    from oracle.make_golden_rl import WIDE_ND_SEED
    cases = np.load(os.path.join(GOLD, "rl_wide_nd_cases.npz"))
    state = rl_oracle.synth_rl_state(seed=WIDE_ND_SEED, lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False)
    x = cases[f"{name}/x"]
    assert x.shape[-1] == 4
    out = rl_oracle.rl_forward(x, state, use_dwells=False, bidirectional=False)
    ref = cases[f"{name}/y"]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-5
    assert (out.argmax(-1) == ref.argmax(-1)).mean() >= 0.999
