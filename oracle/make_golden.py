"""Generate tests/golden/*.npz by running the UNMODIFIED reference on PyTorch-CPU.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

The reference pins nothing numeric at the `predict_on_batch` boundary
(medaka/test/test_architectures.py:58-64 assert shapes only), so the goldens are
outputs of the reference classes themselves:

    medaka.architectures.GRUModel(...).eval().predict_on_batch(medaka.torch_ext.Batch(counts_matrix=x))
    medaka.architectures.MajorityVoteModel().predict_on_batch(...)
    medaka.labels.HaploidLabelScheme().decode_consensus(...)      (argmax -> bases)

Weight sets:
  init     torch.manual_seed(0) default initialisation (near-uniform outputs)
  x3       init * 3 (saturating gates) -- derived in the tests, not stored
  trained  `init` after a short run of the reference's own training step
           (`TorchModel.process_batch`, models.py:315-345, RMSprop lr 1e-3 as training.py:125-133)
           on synthetic majority-vote data: outputs are saturated like a real model's, so
           the argmax-identity checks are not vacuous.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, oracle  # noqa: E402
from medaka_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# exact normalised features of the reference's own test (medaka/test/test_counts.py:92-102)
TEST_COUNTS_FEATURES = np.array([
    [0.5, 0., 0., 0., 0.5, 0., 0., 0., 0., 0.],
    [0., 0.5, 0., 0., 0., 0.5, 0., 0., 0., 0.],
    [0.5, 0., 0., 0., 0.5, 0., 0., 0., 0., 0.],
    [0., 0.25, 0., 0.25, 0., 0., 0., 0.25, 0., 0.25],
    [0.25, 0., 0., 0., 0., 0., 0., 0., 0., 0.],
    [0., 0., 0.5, 0., 0., 0., 0.5, 0., 0., 0.],
    [0.5, 0., 0., 0., 0.5, 0., 0., 0., 0., 0.],
    [0., 0., 0., 0.5, 0., 0., 0., 0.5, 0., 0.],
    [0., 0., 0.5, 0., 0., 0., 0.5, 0., 0., 0.]], dtype="float32")


def main():
    arch, models, te = ref_shim.reference_modules()
    import medaka.labels
    torch.set_num_threads(os.cpu_count())
    os.makedirs(GOLD, exist_ok=True)

    def predict(model, x):
        return model.predict_on_batch(te.Batch(counts_matrix=torch.from_numpy(x))).numpy()

    # ---- weights
    torch.manual_seed(0)
    model = arch.GRUModel(num_features=10, num_classes=5, gru_size=128).eval()
    init = oracle.state_to_numpy(model.state_dict())

    t0 = time.time()
    model.train()
    model.normalise = False          # as torch_ext.py:300 does for training
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-3)
    loss_fn = torch.nn.CrossEntropyLoss()
    for step in range(200):
        x, y = synth.counts_windows(32, 200, seed=10_000 + step, return_labels=True)
        batch = te.Batch(counts_matrix=torch.from_numpy(x), labels=torch.from_numpy(y))
        opt.zero_grad()
        loss, metrics = model.process_batch(batch, loss_fn)
        loss.backward()
        opt.step()
        if step % 25 == 0:
            acc = metrics["n_model_correct"] / metrics["n_positions"]
            print(f"train step {step} loss {loss.item():.4f} acc {acc:.3f}", flush=True)
    model.normalise = True
    model.eval()
    trained = oracle.state_to_numpy(model.state_dict())
    print(f"training took {time.time() - t0:.1f}s")

    np.savez(os.path.join(GOLD, "weights_init.npz"), **init)
    np.savez(os.path.join(GOLD, "weights_trained.npz"), **trained)

    def load(state, scale=1.0):
        m = arch.GRUModel(num_features=10, num_classes=5, gru_size=128).eval()
        m.load_state_dict({k: torch.from_numpy(v * np.float32(scale)) for k, v in state.items()})
        return m

    cases = {}
    # (i) the reference's own exact feature matrix, tiled to 3 windows x 90 columns
    x = np.tile(TEST_COUNTS_FEATURES, (3, 10, 1)).astype(np.float32)
    x[1] = np.roll(x[1], 3, axis=0)
    x[2] = x[2, ::-1]
    cases["testcounts"] = x
    # (ii) synthetic 60x counts at the production window length
    cases["synth60"] = synth.counts_windows(2, 10000, depth=60, seed=1234)
    # (iii) uniform noise like medaka/test/test_sample.py:50
    cases["uniform"] = synth.uniform_windows(5, 333, seed=7)
    # (iv) edge shapes: the B=1 any-T second pass of prediction.py:196-209
    for T in (1, 2, 7, 999):
        cases[f"edge_T{T}"] = synth.counts_windows(1, T, seed=100 + T)
    cases["edge_B3"] = synth.counts_windows(3, 1100, depth=40, seed=55)

    out = {}
    for wname, state, scale in (("init", init, 1.0), ("x3", init, 3.0), ("trained", trained, 1.0)):
        m = load(state, scale)
        for cname, x in cases.items():
            if wname != "trained" and cname == "synth60":
                continue  # keep the fixture small; long-T case on the trained set only
            out[f"{wname}/{cname}"] = predict(m, x)
    np.savez_compressed(os.path.join(GOLD, "gru_inputs.npz"), **cases)
    np.savez_compressed(os.path.join(GOLD, "gru_outputs.npz"), **out)

    # majority vote model + consensus decode through the reference label scheme
    mv = arch.MajorityVoteModel().eval()
    mv_out = {k: predict(mv, cases[k]) for k in ("testcounts", "edge_B3")}
    np.savez_compressed(os.path.join(GOLD, "majority_outputs.npz"), **mv_out)

    ls = medaka.labels.HaploidLabelScheme()

    class _S:  # decode_consensus reads only .label_probs (labels.py:1053-1085)
        pass
    dec = {}
    for key in ("trained/synth60", "trained/edge_B3", "trained/testcounts"):
        seqs = []
        for w in range(out[key].shape[0]):
            s = _S()
            s.label_probs = out[key][w]
            seqs.append(ls.decode_consensus(s))
        dec[key] = np.array(seqs)
    np.savez_compressed(os.path.join(GOLD, "consensus_decode.npz"), **dec)

    # f2 / f3: raw counts -> reference CountsFeatureEncoder._post_process_pileup
    # (features.py:871-935, normalise='total') -> reference model -> reference decode_consensus with
    # qualities (labels.py:1053-1085).  Depth 300 windows give non-trivial quotients.
    fe = medaka.features.CountsFeatureEncoder()
    diet = {}
    for name, (W, T, depth, seed) in {"d60": (3, 700, 60, 91), "d300": (2, 500, 300, 92)}.items():
        raw = synth.counts_windows(W, T, depth=depth, seed=seed, raw=True)
        feats, depths, seqs, quals = [], [], [], []
        for w in range(W):
            pos = np.empty(T, dtype=[("major", int), ("minor", int)])
            pos["major"], pos["minor"] = raw["major"][w], raw["minor"][w]
            region = medaka.common.Region("synth", int(pos["major"][0]), int(pos["major"][-1]) + 1)
            smp = fe._post_process_pileup(raw["counts"][w].astype(np.uint64), pos, region)
            feats.append(smp.features)
            depths.append(smp.depth)
        feats = np.stack(feats)
        assert feats.dtype == np.float32
        probs = predict(load(trained, 1.0), feats)
        for w in range(W):
            s = _S()
            s.label_probs = probs[w]
            seq, q = ls.decode_consensus(s, with_qualities=True)
            seqs.append(seq)
            quals.append(q)
        diet[f"{name}/counts"] = raw["counts"]
        diet[f"{name}/depth"] = np.stack(depths).astype(np.uint32)
        diet[f"{name}/features"] = feats
        diet[f"{name}/probs"] = probs
        diet[f"{name}/seq"] = np.array(seqs)
        diet[f"{name}/qual"] = np.array(quals)
        assert np.array_equal(diet[f"{name}/depth"], raw["depth"]), "synthetic depth != reference depth"
    np.savez_compressed(os.path.join(GOLD, "pcie_diet.npz"), **diet)

    # report how degenerate each weight set is
    for key in ("init/uniform", "x3/uniform", "trained/synth60"):
        p = out[key]
        print(key, "median max-prob", float(np.median(p.max(-1))),
              "class hist", np.bincount(p.argmax(-1).ravel(), minlength=5))
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
