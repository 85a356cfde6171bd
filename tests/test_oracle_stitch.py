"""The restated caller + stitcher (oracle/stitch_oracle.py) against goldens produced by the UNMODIFIED
reference (oracle/make_golden_stitch.py -> tests/golden/stitch_cases.npz): run_prediction's loop shape,
Sample.chunks, trim_samples' junctions and the stitched FASTQ.  CPU only; the model here is the
PyTorch-CPU restatement of reference GRUModel (oracle.make_torch_oracle), so a green run pins every
piece of test infrastructure the GPU end-to-end tests (tests/test_e2e_gpu.py) rely on."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from medaka_amd.torch_ext import Batch
from oracle import oracle, ref_shim
from oracle import stitch_oracle as so


@pytest.fixture(scope="module")
def sgold():
    return dict(np.load(os.path.join(GOLD, "stitch_cases.npz")))


class CpuModel:
    """predict_on_batch of the reference on PyTorch-CPU (models.py:303-313)."""

    def __init__(self, state):
        self.m = oracle.make_torch_oracle(state)
        self.shapes = []

    def predict_on_batch(self, batch):
        self.shapes.append(tuple(batch.counts_matrix.shape))
        return self.m.predict(batch.counts_matrix)


def run_case(sgold, gold, case, model):
    spec, sources = so.load_case(sgold, case)
    jitter = spec.get("jitter", ())
    store = so.predict(so.contig_regions(sources),
                       lambda r: so.pileups_in_region(sources, r, r.ref_name in jitter), model, Batch.collate,
                       spec["chunk_len"], spec["chunk_ovlp"], spec["batch_size"], spec["bam_chunk"])
    lengths = {r.ref_name: r.end for r in so.contig_regions(sources)}
    return store, lengths


@pytest.mark.parametrize("case", ["mini", "cfg1"])
def test_restated_pipeline_matches_reference_goldens(sgold, gold, case):
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    model = CpuModel(gold["weights_trained"])
    store, lengths = run_case(sgold, gold, case, model)
    # loop shape: the batches the model saw (order within the first pass depends on thread timing)
    want = [tuple(b) for b in sgold[f"{case}/batches"]]
    assert sorted(model.shapes) == sorted(want)
    n_first = int(sgold[f"{case}/n_batches_first_pass"])
    assert sorted(model.shapes[:n_first]) == sorted(want[:n_first])
    assert model.shapes[n_first:] == want[n_first:]          # the remainder pass is sequential: B = 1, any T
    assert set(store) == set(sgold[f"{case}/written"].tolist())
    # probabilities: per sample sum of the winning probability (a light pin; the strings below are the strict one)
    for name, ref in zip(sgold[f"{case}/written"], sgold[f"{case}/pmax_sum"]):
        assert abs(float(store[str(name)].label_probs.max(-1).sum(dtype=np.float64)) - ref) < 1e-2
    # trim_samples: every junction, the trimmed views and the contig breaks
    so.JUNCTION_LOG = []
    trimmed = []
    try:
        index = so.sorted_names(store.keys())
        for ref in index:
            for s, last, _ in so.trim_samples(store[n] for n in index[ref]):
                trimmed.append((s.name, last))
    finally:
        junctions, so.JUNCTION_LOG = so.JUNCTION_LOG, None
    assert np.array_equal(np.array(junctions).reshape(-1, 3), sgold[f"{case}/junctions"])
    assert [t[0] for t in trimmed] == sgold[f"{case}/trimmed"].tolist()
    assert [t[1] for t in trimmed] == sgold[f"{case}/trim_last"].tolist()
    # the stitched FASTQ, byte for byte
    assert so.fastq(store, lengths) == str(sgold[f"{case}/fastq"])


def test_window_starts_cover_the_reference_cases():
    """sliding_window (common.py:803-823): stepped windows + one right-aligned remainder."""
    assert so.window_starts(10, 4, 3) == [0, 3, 6]
    assert so.window_starts(11, 4, 3) == [0, 3, 6, 7]
    assert so.window_starts(4, 4, 3) == [0]
    assert so.window_starts(102345, 10000, 9000)[-2:] == [90000, 92345]


def test_region_split_semantics():
    """Region.split(fixed_size=False) (common.py:711-736), including its trailing contained piece."""
    R = so.Region
    assert so.split_region(R("c", 0, 10000), 4000, 200) == [R("c", 0, 4000), R("c", 3800, 7800), R("c", 7600, 10000)]
    assert so.split_region(R("c", 0, 7700), 4000, 200) == [R("c", 0, 4000), R("c", 3800, 7700), R("c", 7600, 7700)]
    assert so.split_region(R("c", 5, 900), 4000, 200) == [R("c", 5, 900)]


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present (GPU box)")
def test_goldens_regenerate_from_the_live_reference(sgold, gold):
    """The committed fixture is what the unmodified reference produces today."""
    from oracle import make_golden_stitch as mg
    out = mg.run_case("mini", mg.CASES["mini"], gold["weights_trained"])
    for key in ("mini/fastq", "mini/junctions", "mini/trimmed", "mini/trim_last"):
        assert np.array_equal(out[key], sgold[key]), key
    assert set(out["mini/written"].tolist()) == set(sgold["mini/written"].tolist())


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present (GPU box)")
def test_engine_collate_inside_the_unmodified_run_prediction(sgold, gold):
    """`integration.install()` replaces the counts branch of the reference's `Batch.collate`; the UNMODIFIED
    `run_prediction` (threaded DataLoader -> Batcher thread -> collate -> predict_on_batch, prediction.py:36-60,
    225-370) run with it must produce the committed FASTQ, junctions and samples byte for byte."""
    from medaka_amd import integration, torch_ext
    from oracle import make_golden_stitch as mg
    ref_shim.install()
    calls = []
    real = torch_ext.stack_counts

    def spy(feats, threads=None):
        calls.append(len(feats))
        return real(feats, threads)
    integration.install(collate=True)
    torch_ext.stack_counts, restore = spy, real
    try:
        import medaka.torch_ext as rte
        rte.Batch.collate = classmethod(integration._fast_collate(integration._ORIG["collate"].__func__))   # bind the spy
        out = mg.run_case("mini", mg.CASES["mini"], gold["weights_trained"])
    finally:
        torch_ext.stack_counts = restore
        integration.uninstall()
    assert sum(calls) > 0, "the patched collate was never called by the reference's Batcher thread"
    for key in ("mini/fastq", "mini/junctions", "mini/trimmed", "mini/trim_last"):
        assert np.array_equal(out[key], sgold[key]), key


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present (GPU box)")
def test_engine_model_class_inside_the_unmodified_run_prediction(sgold, gold, monkeypatch):
    """The engine-backed `medaka_amd.models.GRUModel` -- the object `integration.convert` hands to medaka -- driven by
    the UNMODIFIED `run_prediction` (batched pass + B = 1 remainder pass), its rows written through `Sample.amend` /
    `write_sample` and stitched by the reference's own code: with the device replaced by the CPU oracle (no GPU in the
    build container; the kernels' own parity is the GPU suite's job) the FASTQ must be the committed one.  What this
    pins is the plumbing: `Batch` handling, `get_model_input_features`, the host path's pointer / shape contract, the
    returned tensor's life in the writer."""
    import ctypes
    import torch
    from medaka_amd import models as amd_models
    from oracle import make_golden_stitch as mg
    from oracle import oracle as cpu_oracle
    cpu = cpu_oracle.make_torch_oracle(gold["weights_trained"])
    calls = []

    class OracleDevice:
        """Stands where `engine.GruEngine` stands: same `forward_ptr(x_ptr, B, T, out_ptr, host=True)` contract."""
        def set_precision(self, half): assert not half
        def set_variant(self, v): pass
        def set_normalise(self, n): assert n
        def set_option(self, k, v): pass
        def close(self): pass
        def take_promised(self, shape): return None        # (no batch is ever started ahead here: nothing was promised)
        def promise(self, tensor): pass
        def drop_pending(self): pass

        def forward_ptr(self, x_ptr, B, T, out_ptr, stream=None, host=False):
            assert host and stream is None
            x = np.ctypeslib.as_array(ctypes.cast(x_ptr, ctypes.POINTER(ctypes.c_float)), shape=(B, T, 10))
            out = np.ctypeslib.as_array(ctypes.cast(out_ptr, ctypes.POINTER(ctypes.c_float)), shape=(B, T, 5))
            out[...] = cpu.predict(x.copy()).numpy()
            calls.append((B, T))
    m = amd_models.GRUModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gold["weights_trained"].items()})
    m.eval()
    dev = OracleDevice()
    monkeypatch.setattr(amd_models.GRUModel, "engine", lambda self: dev)
    out = mg.run_case("mini", mg.CASES["mini"], gold["weights_trained"], model=m)
    assert len(calls) >= 3 and any(b == 1 for b, _ in calls)               # batched pass and B = 1 remainders
    for key in ("mini/fastq", "mini/junctions", "mini/trimmed", "mini/trim_last"):
        assert np.array_equal(out[key], sgold[key]), key
    assert out["mini/batches"].tolist() == sgold["mini/batches"].tolist()


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present (GPU box)")
def test_restated_relationship_and_overlap_against_live_reference(sgold):
    """Sample.relative_position / overlap_indices on pairs drawn from the golden pileups, including
    reversed, contained, abutting and gapped pairs."""
    ref_shim.install()
    import medaka.common as mc
    spec, sources = so.load_case(sgold, "mini")
    p = sources["m4"][0]
    cuts = [(0, 1000), (800, 1800), (1000, 2000), (200, 700), (2500, 3000), (0, 1000), (1800, 2600)]
    names = {"s2_within_s1": "within", "s1_within_s2": "within_rev", "forward_abutted": "abutted",
             "reverse_abutted": "abutted_rev", "forward_overlap": "overlap", "reverse_overlap": "overlap_rev",
             "forward_gapped": "gapped", "reverse_gapped": "gapped_rev"}
    mine = [p.cut(slice(a, b)) for a, b in cuts]
    theirs = [mc.Sample(ref_name=m.ref_name, features=m.features, labels=None, ref_seq=None, positions=m.positions,
                        label_probs=None, depth=m.depth) for m in mine]
    seen = set()
    for i in range(len(cuts)):
        for j in range(len(cuts)):
            if i == j:
                continue
            rel = mc.Sample.relative_position(theirs[i], theirs[j])
            assert so.relationship(mine[i], mine[j]) == names[rel.name], (cuts[i], cuts[j])
            seen.add(rel.name)
            if rel.name in ("forward_overlap", "forward_abutted"):
                assert so.overlap_indices(mine[i], mine[j]) == tuple(mc.Sample.overlap_indices(theirs[i], theirs[j]))
    assert len(seen) >= 7
