"""Host logic of the split scan (medaka_amd/csrc/api.hip `plan_split_shape`, exported as `mdk_split_plan`): runs without a
GPU.  The properties the device side relies on:
  * the delivered ranges of the chunks tile [0, T) exactly, in order;
  * every chunk lies inside its window and has `columns` columns, a multiple of 16 (the overlap machinery's unit);
  * on every side where a chunk does not end at the window's own end it has at least `margin` columns of warm-up, and
    both certificate points of every junction (margin / 2 past it, both directions) lie inside both neighbours;
  * a chunk's own columns are at least twice its two margins; the virtual batch respects the chunk-window budget."""
import pytest
from hypothesis import given, settings, strategies as st

from medaka_amd import engine, lib


def check_plan(B, T, share, mode, margin):
    p = engine.split_plan(B, T, share, mode, margin)
    S = p["chunks"]
    if S == 1:
        assert p == {"chunks": 1, "columns": T, "margin": 0, "start": [0], "first": [0], "last": [T]}
        return p
    Tv, G = p["columns"], p["margin"]
    assert 2 <= S <= 16 and G == margin and Tv % 16 == 0 and Tv < T
    assert p["first"][0] == 0 and p["last"][-1] == T
    for k in range(S):
        s, a, b = p["start"][k], p["first"][k], p["last"][k]
        assert 0 <= s and s + Tv <= T and s <= a < b <= s + Tv
        if k:
            assert p["first"][k] == p["last"][k - 1]
            assert a - s >= G                      # warm-up of the upward scans
        else:
            assert s == 0
        if k < S - 1:
            assert s + Tv - b >= G                 # warm-up of the downward scans
        else:
            assert s + Tv == T
        assert b - a >= 4 * G - 1                  # own columns vs margins (T // S >= 4 G, cores differ by <= 1)
    for j in range(S - 1):                         # certificate points of junction j (scan_split.hpp k_split_verify)
        a = p["first"][j + 1]
        for t in (a - 1, a - 1 + G // 2, a, a - G // 2):
            for k in (j, j + 1):
                assert p["start"][k] <= t < p["start"][k] + Tv, (j, t, k, p)
    if mode == 1:
        budget = 1024 if share == 1 else 1600 // share
        assert S * B <= budget and S >= (3 if share == 1 and S == budget // B else 2)
    else:
        assert S <= mode
    assert S <= T // (4 * G)
    return p


@settings(max_examples=400, deadline=None)
@given(B=st.integers(1, 1200), T=st.integers(1, 40000), share=st.integers(1, 8), mode=st.integers(1, 16),
       margin=st.sampled_from([16, 32, 64, 128, 256, 512, 1024, 4096]))
def test_plan_properties(B, T, share, mode, margin):
    check_plan(B, T, share, mode, margin)


def test_reference_batch_shapes():
    """The shapes the reference produces (prediction.py: batch 100 by default, 200 in BASELINE configs[1], 10 000-column
    windows, B = 1 un-chunked remainders)."""
    assert check_plan(200, 10000, 1, 1, 128) == {
        "chunks": 5, "columns": 2256, "margin": 128, "start": [0, 1872, 3872, 5872, 7744],
        "first": [0, 2000, 4000, 6000, 8000], "last": [2000, 4000, 6000, 8000, 10000]}
    assert check_plan(100, 10000, 1, 1, 128)["chunks"] == 10
    assert check_plan(10, 10000, 1, 1, 128)["chunks"] == 16
    assert check_plan(1, 9999, 1, 1, 128)["chunks"] == 16
    assert check_plan(1, 777, 1, 1, 128)["chunks"] == 1                 # shorter than 8 margins
    assert check_plan(341, 10000, 1, 1, 128)["chunks"] == 3
    assert check_plan(342, 10000, 1, 1, 128)["chunks"] == 1             # two chunks alone are not worth the margins
    assert check_plan(1000, 10000, 1, 1, 128)["chunks"] == 1            # fills the chip by itself
    # processes that share the GPU: 1600 / K chunk-windows each, two chunks accepted
    assert check_plan(200, 10000, 2, 1, 128)["chunks"] == 4
    assert check_plan(200, 10000, 3, 1, 128)["chunks"] == 2
    assert check_plan(200, 10000, 4, 1, 128)["chunks"] == 2
    assert check_plan(200, 10000, 8, 1, 128)["chunks"] == 1
    # escalated margins (128 -> 256 -> 512) keep splitting a 10 000-column window
    assert check_plan(200, 10000, 1, 1, 256)["chunks"] == 5
    assert check_plan(200, 10000, 1, 1, 512)["chunks"] == 4
    assert check_plan(200, 10000, 1, 1, 4096)["chunks"] == 1


def test_bad_arguments_are_errors():
    for args in ((-1, 100, 1, 1, 128), (1, 100, 0, 1, 128), (1, 100, 9, 1, 128), (1, 100, 1, 17, 128), (1, 100, 1, 1, 100)):
        with pytest.raises(lib.EngineError):
            engine.split_plan(*args)
    assert engine.split_plan(0, 0)["chunks"] == 1
    assert engine.split_plan(5, 10000, scan_split=0)["chunks"] == 1


def test_premise_on_the_reference_arithmetic(gold):
    """The premise of the split, checked with the reference's own arithmetic (PyTorch-CPU nn.GRU -> Linear -> softmax,
    oracle.make_torch_oracle) and the engine's own plan: a chunk that starts from h = 0 one margin before its first
    delivered column, and ends one margin after its last, gives on its delivered columns what the full scan gives
    there -- to rounding for the bundled architecture's weights, and visibly NOT for a model that does not forget
    (weights x 5), which is why the engine certifies every junction instead of assuming this."""
    import numpy as np
    from medaka_amd import synth
    from oracle import oracle
    B, T = 3, 4096
    x = synth.counts_windows(B, T, depth=50, seed=11)
    plan = engine.split_plan(B, T, scan_split=4, margin=128)
    assert plan["chunks"] == 4

    def stitched(model):
        out = np.empty((B, T, 5), np.float32)
        for k in range(plan["chunks"]):
            s, a, b = plan["start"][k], plan["first"][k], plan["last"][k]
            out[:, a:b] = model.predict(np.ascontiguousarray(x[:, s:s + plan["columns"]])).numpy()[:, a - s:b - s]
        return out

    for name, scale, bound in (("trained", 1.0, 2e-6), ("init", 1.0, 2e-6), ("init", 5.0, None)):
        st = {k: (v * np.float32(scale) if k.startswith("gru.weight") else v) for k, v in gold["weights_" + name].items()}
        model = oracle.make_torch_oracle(st)
        d = float(np.abs(stitched(model) - model.predict(x).numpy()).max())
        if bound is not None:
            assert d <= bound, (name, scale, d)
        else:
            assert d > 1e-2, (name, scale, d)
