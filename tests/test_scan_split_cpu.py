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


# ---- the margin learner (api.hip MarginLearner, exported device-free as mdk_margin_sim) -----------------------------------
LADDER = (64, 96, 128, 192, 256, 384, 512)


def _sim(start, adapt, need, n):
    import ctypes
    from medaka_amd import lib
    m, f = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    lib.check(lib.load().mdk_margin_sim(start, adapt, need, n, m, f), "mdk_margin_sim")
    return list(m), list(f)


@pytest.mark.parametrize("need", LADDER)
def test_margin_learner_settles_at_what_the_model_needs(need):
    """A model that certifies iff the margin is >= `need`: from the default 128 the learner ends at exactly `need`, climbs one
    rung per rejection (never a doubling), shrinks one rung per `adapt` quiet calls, pays at most one wasted forward per
    rejected trial -- one, ever -- and every call is answered at a margin that certifies."""
    margins, forwards = _sim(128, 8, need, 200)
    assert all(g >= need and g in LADDER for g in margins), margins[:40]
    assert margins[-1] == need
    if need > 128:          # climbs during the FIRST call: one forward per rung
        rungs = [r for r in LADDER if 128 <= r <= need]
        assert forwards[0] == len(rungs) and margins[0] == need and set(forwards[1:]) == {1}
    else:                   # shrinks: need's rung reached after (rungs down) x 8 quiet calls, then ONE rejected trial below it (if there is a rung below)
        down = [r for r in LADDER if need <= r <= 128][::-1]
        assert margins[:8] == [128] * 8
        settle = 8 * (len(down) - 1)
        assert margins[settle] == need
        wasted = sum(f - 1 for f in forwards)
        assert wasted == (1 if need > 64 else 0), (wasted, forwards[:60])
        assert sorted(margins, reverse=True) == margins           # never back up
    # the steady state is one forward per call
    assert forwards[-50:] == [1] * 50


def test_margin_learner_never_retries_a_rejected_margin_and_gives_up_beyond_the_ladder():
    margins, forwards = _sim(128, 3, 96, 100)            # 64 is tried once (after 3 quiet calls at 96), rejected, never again
    assert margins[-1] == 96 and sum(f - 1 for f in forwards) == 1
    margins, forwards = _sim(128, 8, 0, 10)              # never certifies: 128, 192, 256, 384, 512 in the first call, then sequential
    assert forwards[0] == 5 and margins == [0] * 10 and forwards[1:] == [0] * 9
    margins, forwards = _sim(32, 8, 192, 5)              # a starting margin below the ladder joins it at the next rung
    assert margins[0] == 192 and forwards[0] == 5        # 32, 64, 96, 128 rejected, 192 certified
    margins, forwards = _sim(128, 0, 64, 50)             # adapt = 0: margins only grow
    assert set(margins) == {128} and set(forwards) == {1}
