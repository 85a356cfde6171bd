"""The CPU oracle pinned against outputs of the UNMODIFIED reference (tests/golden/, made by
oracle/make_golden.py) -- and, when the reference tree is present, against the reference live."""
import numpy as np
import pytest

from conftest import weight_set
from oracle import oracle, ref_shim

TOL = 2e-6   # fp32 re-association between torch/MKL and plain C


def _cases(gold):
    for key in sorted(gold["gru_outputs"]):
        wname, cname = key.split("/")
        yield key, wname, gold["gru_inputs"][cname], gold["gru_outputs"][key]


def test_c_oracle_matches_reference_goldens(gold):
    n = 0
    for key, wname, x, ref in _cases(gold):
        if x.shape[1] > 2000:   # keep the CPU suite short; long case covered below
            continue
        out = oracle.c_gru_forward(x, weight_set(gold, wname))
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() <= TOL, key
        assert (out.argmax(-1) == ref.argmax(-1)).all(), key
        n += 1
    assert n >= 20


def test_c_oracle_long_window(gold):
    x = gold["gru_inputs"]["synth60"]
    ref = gold["gru_outputs"]["trained/synth60"]
    out = oracle.c_gru_forward(x, gold["weights_trained"])
    assert np.abs(out - ref).max() <= TOL
    assert (out.argmax(-1) == ref.argmax(-1)).all()


def test_torch_oracle_matches_reference_goldens(gold):
    for key, wname, x, ref in _cases(gold):
        if x.shape[1] > 2000:
            continue
        m = oracle.make_torch_oracle(weight_set(gold, wname))
        out = m.predict(x).numpy()
        # same PyTorch ops as the reference: equal up to thread-count dependent blocking
        assert np.abs(out - ref).max() <= TOL, key


def test_majority_oracle(gold):
    for cname, ref in gold["majority_outputs"].items():
        out = oracle.c_majority_forward(gold["gru_inputs"][cname])
        assert np.abs(out - ref).max() <= 2e-7


def test_reference_test_vector_is_in_goldens(gold):
    # first window of `testcounts` is the exact (9,10) matrix of medaka/test/test_counts.py:92-102
    x = gold["gru_inputs"]["testcounts"][0, :9]
    assert x[3].tolist() == [0., 0.25, 0., 0.25, 0., 0., 0., 0.25, 0., 0.25]
    assert x.sum() == pytest.approx(8.25)


def test_consensus_decode_golden(gold):
    # argmax over '*ACGT' with '*' dropped (reference labels.py:1053-1085), on oracle output
    alphabet = np.array(list("*ACGT"))
    x = gold["gru_inputs"]["edge_B3"]
    out = oracle.c_gru_forward(x, gold["weights_trained"])
    for w in range(x.shape[0]):
        s = "".join(alphabet[out[w].argmax(-1)]).replace("*", "")
        assert s == str(gold["consensus_decode"]["trained/edge_B3"][w])


def test_edge_shapes():
    rng = np.random.default_rng(0)
    state = {k: rng.standard_normal(s).astype(np.float32) * 0.1 for k, s in
             zip(oracle.state_keys(), [(384, 10), (384, 128), (384,), (384,)] * 2 +
                 [(384, 256), (384, 128), (384,), (384,)] * 2 + [(5, 256), (5,)])}
    assert oracle.c_gru_forward(np.zeros((0, 5, 10), np.float32), state).shape == (0, 5, 5)
    assert oracle.c_gru_forward(np.zeros((2, 0, 10), np.float32), state).shape == (2, 0, 5)
    p = oracle.c_gru_forward(rng.random((1, 1, 10), dtype=np.float32), state)
    assert p.shape == (1, 1, 5) and abs(p.sum() - 1) < 1e-6


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_oracle_against_live_reference(gold):
    import torch
    arch, models, te = ref_shim.reference_modules()
    m = arch.GRUModel(num_features=10, num_classes=5, gru_size=128).eval()
    st = weight_set(gold, "x3")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    rng = np.random.default_rng(11)
    x = rng.random((3, 257, 10), dtype=np.float32)
    ref = m.predict_on_batch(te.Batch(counts_matrix=torch.from_numpy(x))).numpy()
    assert np.abs(oracle.c_gru_forward(x, st) - ref).max() <= TOL
    assert np.abs(oracle.make_torch_oracle(st).predict(x).numpy() - ref).max() <= TOL


@pytest.mark.parametrize("name", ["d60", "d300"])
def test_normalise_and_decode_restatements_match_reference(name):
    """f2/f3 oracles against the unmodified reference's `_post_process_pileup` (features.py:871-935)
    and `decode_consensus(with_qualities=True)` (labels.py:1053-1085), oracle/make_golden.py."""
    import os
    from conftest import GOLD
    d = np.load(os.path.join(GOLD, "pcie_diet.npz"))
    feats = oracle.normalise_counts(d[f"{name}/counts"], d[f"{name}/depth"])
    assert feats.dtype == np.float32 and np.array_equal(feats, d[f"{name}/features"])
    for w in range(feats.shape[0]):
        seq, qual = oracle.decode_consensus(d[f"{name}/probs"][w], with_qualities=True)
        assert seq == str(d[f"{name}/seq"][w]) and qual == str(d[f"{name}/qual"][w])
