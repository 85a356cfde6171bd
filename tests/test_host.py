"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared
symbol, the reference-interface mirrors behave like the reference, sharding, synthetic data."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from medaka_amd import engine, lib, models, sharding, synth
from medaka_amd.torch_ext import Batch


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "medaka_amd.h")).read()
    release, _, debug = header.partition("#ifdef MDK_DEBUG_HOOKS")
    debug, _, tail = debug.partition("#endif")
    declared = set(re.findall(r"\b(mdk_[a-z0-9_]+)\s*\(", release + tail))
    declared -= {"mdk_gru_timing"}
    assert len(declared) >= 20
    L = lib.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/medaka_amd.h but not exported"
        assert name in lib.ABI, f"{name} missing from the ctypes ABI table"
    assert set(lib.ABI) <= declared
    # the test / profiling hooks live in the debug library only: the release library exports none of them
    hooks = set(re.findall(r"\b(mdk_[a-z0-9_]+)\s*\(", debug))
    assert hooks == set(lib.DEBUG_ABI) and len(hooks) == 3, hooks
    if not lib.is_debug_library():
        import subprocess
        from medaka_amd import build
        for path, want in ((build.LIB, False), (build.LIB_DEBUG, True)):
            if os.path.exists(path):
                syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
                for h in hooks:
                    assert (f" {h}" in syms) == want, (path, h)


def test_version_and_error_string():
    L = lib.load()
    assert b"gfx950" in L.mdk_version()
    assert isinstance(lib.last_error(), str)


def test_no_cpu_fallback_without_device():
    """Without a HIP device the engine must fail loudly, never compute on the host."""
    if lib.device_count() > 0:
        pytest.skip("a HIP device is visible")
    state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_init.npz")))
    with pytest.raises(lib.EngineError):
        engine.GruEngine(state)
    m = models.GRUModel()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.predict_on_batch(Batch(counts_matrix=torch.zeros(1, 4, 10)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        models.MajorityVoteModel().predict_on_batch(Batch(counts_matrix=torch.zeros(1, 4, 10)))


def test_engine_rejects_bad_weights():
    state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_init.npz")))
    bad = dict(state)
    bad["gru.weight_hh_l0"] = bad["gru.weight_hh_l0"][:, :64]
    with pytest.raises(ValueError):
        engine.GruEngine(bad)
    missing = {k: v for k, v in state.items() if k != "linear.bias"}
    with pytest.raises(KeyError):
        engine.GruEngine(missing)


def test_counts_entry_checks_its_arguments_before_any_device_call():
    """`forward_counts_host`: dtype, range, shape and the caller's result buffers are checked on the host --
    nothing below reaches the library (the handle is a dummy)."""
    import types
    from medaka_amd import engine
    eng = types.SimpleNamespace(num_features=10, num_classes=5, _h=None)
    call = engine.GruEngine.forward_counts_host
    counts, depth = np.ones((2, 7, 10), np.uint16), np.ones((2, 7), np.uint32)
    with pytest.raises(ValueError, match="must be integers"):
        call(eng, counts.astype(np.float32), depth)
    with pytest.raises(ValueError, match="65535"):
        call(eng, counts.astype(np.int64) * 70000, depth)
    with pytest.raises(ValueError, match="expected counts"):
        call(eng, counts[:, :, :9], depth)
    with pytest.raises(ValueError, match="expected counts"):
        call(eng, counts, depth[:, :6])
    with pytest.raises(ValueError, match="nothing requested"):
        call(eng, counts, depth, probs=False, decoded=False)
    good = (np.empty((2, 7), np.uint8), np.empty((2, 7), np.float32))
    for bad in ((good[0],), (good[1], good[0]), (good[0], np.empty((2, 7), np.float64)), (good[0][:, ::2], good[1]),
                (np.empty((2, 7, 5), np.float32),) + good):
        with pytest.raises(ValueError, match="out must be"):
            call(eng, counts, depth, probs=False, decoded=True, out=bad)


def test_pipelined_entry_checks_its_arguments_and_the_promise_queue():
    """`mdk_gru_forward_pipelined` / `mdk_gru_drop_pending`: null model, null buffer and token 0 are refused before any device
    call; the Python side of the promise (`GruEngine.promise / take_promised`): a buffer of another shape withdraws the promise
    and tells the engine to forget what it started."""
    import ctypes
    import types
    from medaka_amd import engine, lib
    L = lib.load()
    buf = (ctypes.c_float * 4)()
    assert L.mdk_gru_forward_pipelined(None, 1, 1, 4, buf, None) == lib.MDK_ERR_ARG and "null model" in lib.last_error()
    assert L.mdk_gru_drop_pending(None) == lib.MDK_ERR_ARG
    dropped = []
    eng = types.SimpleNamespace(_promised=None, drop_pending=lambda: dropped.append(1))
    take = lambda shape: engine.GruEngine.take_promised(eng, shape)
    assert take((2, 3, 5)) is None and not dropped                      # nothing promised: nothing to withdraw
    t = types.SimpleNamespace(shape=(2, 3, 5))
    engine.GruEngine.promise(eng, t)
    assert take((2, 3, 5)) is t and eng._promised is None and not dropped
    engine.GruEngine.promise(eng, t)
    assert take((4, 3, 5)) is None and dropped == [1] and eng._promised is None


def test_grumodel_mirrors_reference_interface():
    m = models.GRUModel(num_features=10, num_classes=5, gru_size=128)
    # state_dict keys are the stock nn.GRU / nn.Linear names
    assert list(m.state_dict().keys()) == engine.state_keys(2, True)
    assert m.state_dict()["gru.weight_ih_l1_reverse"].shape == (384, 256)
    d = m.to_dict()
    assert d["type"] == "GRUModel"
    assert d["kwargs"] == dict(num_features=10, num_classes=5, gru_size=128, n_layers=2,
                               bidirectional=True, time_steps=None, classify_activation=None)
    m2 = models.model_from_dict(d)
    assert isinstance(m2, models.GRUModel)
    assert m.device() == torch.device("cpu")
    assert m.normalise is True and m.half_precision is False
    m.half()
    assert m.half_precision is True and m.gru.weight_hh_l0.dtype == torch.float16
    with pytest.raises(NotImplementedError):
        m.process_batch(None, None)
    with pytest.raises(ValueError):
        models.model_from_dict({"type": "NoSuchModel", "kwargs": {}})
    with pytest.warns(UserWarning):
        models.GRUModel(time_steps=100)


def test_latent_space_lstm_mirrors_reference_interface():
    from medaka_amd import engine as E
    for bidir in (True, False):
        m = models.LatentSpaceLSTM(bidirectional=bidir)
        keys = [k for k in m.state_dict() if "num_batches_tracked" not in k and "expansion_layer" not in k
                or k.startswith("pre_pool")]
        assert keys == E.rl_state_keys(bidir)
    m = models.model_from_dict({"type": "LatentSpaceLSTM", "kwargs": {"use_dwells": True}})
    assert m.read_level_conv.convs[0].weight.shape == (128, 8, 1)
    assert m.to_dict()["kwargs"]["kernel_sizes"] == [1, 17]

    class ReadAlignmentFeatureEncoder:
        dtypes = ("",)
        include_dwells = False
    class CountsFeatureEncoder: pass
    models.LatentSpaceLSTM().check_feature_encoder_compatibility(ReadAlignmentFeatureEncoder())
    with pytest.raises(ValueError):
        models.LatentSpaceLSTM().check_feature_encoder_compatibility(CountsFeatureEncoder())
    with pytest.raises(ValueError):
        m.check_feature_encoder_compatibility(ReadAlignmentFeatureEncoder())   # model wants dwells
    b = Batch(read_level_features=torch.zeros(1, 3, 2, 4, dtype=torch.uint8))
    assert m.get_model_input_features(b) is b.read_level_features


def test_feature_encoder_compatibility():
    class CountsFeatureEncoder: pass
    class ReadAlignmentFeatureEncoder: pass
    class Sub(CountsFeatureEncoder): pass
    class Other: pass
    m = models.GRUModel()
    m.check_feature_encoder_compatibility(CountsFeatureEncoder())
    m.check_feature_encoder_compatibility(ReadAlignmentFeatureEncoder())
    m.check_feature_encoder_compatibility(Sub())
    with pytest.raises(ValueError):
        m.check_feature_encoder_compatibility(Other())


def test_batch_collate_counts_and_read_level():
    class S:
        def __init__(self, f, labels=None):
            self.features, self.labels = f, labels
    rng = np.random.default_rng(0)
    samples = [S(rng.random((7, 10)).astype(np.float64), np.arange(7)) for _ in range(3)]
    b = Batch.collate(samples)
    assert b.counts_matrix.shape == (3, 7, 10) and b.counts_matrix.dtype == torch.float32
    assert b.labels.shape == (3, 7) and b.read_level_features is None
    assert b.features is b.counts_matrix and b.majority_vote_probs is None
    rl = [S(rng.integers(0, 5, (6, d, 4)).astype(np.uint8)) for d in (3, 5)]
    b = Batch.collate(rl)
    assert b.read_level_features.shape == (2, 6, 5, 4) and b.read_level_features.dtype == torch.uint8
    assert int(b.read_level_features[0, :, 3:, :].sum()) == 0   # zero padded to max depth
    with pytest.raises(ValueError):
        Batch.collate([S(np.zeros(4))])


def test_sharding_lpt_on_the_reference_grid():
    contigs = [("chr1", 250_000_000), ("chr2", 40_000_000), ("chr3", 30_000_000), ("chrM", 16_569)]
    shards = sharding.shard_regions(contigs, 8)
    assert len(shards) == 8 and all(len(s) > 0 for s in shards)
    load = [sum(r.end - r.start for r in s) for s in shards]
    assert max(load) <= 1.02 * (sum(load) / 8)
    # the union is exactly what ONE `medaka inference` would cut for itself (prediction.py:100-110 ->
    # Region.split(bam_chunk, overlap=chunk_ovlp, fixed_size=False), restated in oracle/stitch_oracle.py
    # and pinned there against the reference)
    from oracle import stitch_oracle as so
    want = []
    for name, length in contigs:
        want.extend(so.split_region(so.Region(name, 0, length), 1_000_000, 1000))
    got = sorted((r for s in shards for r in s), key=lambda r: ([c[0] for c in contigs].index(r.ref_name), r.start))
    assert [tuple(r) for r in got] == [tuple(r) for r in want]
    assert all(r.end - r.start <= 1_000_000 for r in got)          # never cut again by the per-GPU process
    # whole small contigs are never cut
    assert sum(1 for s in shards for r in s if r.ref_name == "chrM") == 1
    one = sharding.shard_regions(contigs, 1)
    assert [r.ref_name for r in one[0]] == ["chr1", "chr2", "chr3", "chrM"]
    assert sharding.region_str(one[0][0]) == "chr1:0-250000000"
    lo_hi = [sharding.shard_windows(203, 8, r) for r in range(8)]
    assert lo_hi[0][0] == 0 and lo_hi[-1][1] == 203
    assert all(a[1] == b[0] for a, b in zip(lo_hi, lo_hi[1:]))
    assert sharding.shard_regions([], 4) == [[], [], [], []]


def test_synthetic_counts_contract():
    x, y = synth.counts_windows(3, 2000, depth=60, seed=5, return_labels=True)
    assert x.shape == (3, 2000, 10) and x.dtype == np.float32 and y.shape == (3, 2000)
    assert x.min() >= 0 and np.isfinite(x).all()
    rows = x.sum(-1)
    assert rows.max() <= 1 + 1e-5                  # counts / depth of the parent major column
    major = np.abs(rows - 1) < 1e-5                # major columns sum to 1
    assert 0.5 < major.mean() < 0.8                # ~30-45 % minor (insertion) columns
    assert set(np.unique(y)) == {0, 1, 2, 3, 4}
    x2 = synth.counts_windows(3, 2000, depth=60, seed=5)
    assert np.array_equal(x, x2)                   # seeded


# ---- integration hook against the UNMODIFIED reference (build container only) ----------------
from oracle import ref_shim  # noqa: E402


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_integration_hook_keeps_reference_model_on_cpu(tmp_path):
    """`integration.install()` wraps ModelStoreTGZ.load_model (datastore.py:135-157); a CPU load
    (`medaka inference --cpu`) must come back as the reference's own model, untouched."""
    import pickle
    import sys
    import tarfile
    import types
    import functools
    ref_shim.install()
    # datastore imports h5py lazily through module attributes only; the stub is enough
    import medaka.datastore as ds
    import medaka.models as ref_models
    from medaka_amd import integration

    # a model archive exactly as ModelMetaCheckpoint writes it (torch_ext.py:40-61):
    # <top>/weights.pt + <top>/meta.pkl with a partial(model_from_dict, {...})
    torch.manual_seed(1)
    import medaka.architectures as arch
    ref = arch.GRUModel(num_features=10, num_classes=5, gru_size=128)
    top = tmp_path / "model"
    top.mkdir()
    torch.save(ref.state_dict(), top / "weights.pt")
    meta = {"model_function": functools.partial(ref_models.model_from_dict, ref.to_dict())}
    with open(top / "meta.pkl", "wb") as fh:
        pickle.dump(meta, fh)
    tgz = tmp_path / "toy_model_pt.tar.gz"
    with tarfile.open(tgz, "w:gz") as tar:
        tar.add(top, arcname="model")

    integration.install()
    try:
        with ds.ModelStoreTGZ(str(tgz)) as store:
            model = store.load_model(device=torch.device("cpu"))
        assert type(model).__module__.startswith("medaka.architectures")   # reference class
        assert type(model).__name__ == "GRUModel"
        for k, v in ref.state_dict().items():
            assert torch.equal(model.state_dict()[k], v)
    finally:
        integration.uninstall()
    assert ds.ModelStoreTGZ.load_model.__name__ == "load_model"


def test_batch_assembly_matches_the_reference_expression():
    """`stack_counts` (mdk_gather_rows on a few host threads) == `torch.stack([...]).float()` of reference
    torch_ext.py:147-148, for overlapping row views as `Sample.chunks` makes them; odd inputs take the reference
    expression itself."""
    from medaka_amd import torch_ext as te
    rng = np.random.default_rng(3)
    big = rng.random((5000, 10), dtype=np.float32)
    for n, T, step, threads in ((1, 1, 1, 1), (7, 300, 250, 3), (33, 120, 100, 64), (5, 64, 64, 2)):
        feats = [big[i * step:i * step + T] for i in range(n)]
        want = torch.stack([torch.from_numpy(f) for f in feats]).float()
        got = te.stack_counts(feats, threads=threads)
        assert got.dtype == torch.float32 and got.shape == want.shape and got.is_contiguous()
        assert torch.equal(got, want)
    odd = [big[::2][:50], big[::2][50:100]]                       # not C-contiguous
    assert torch.equal(te.stack_counts(odd), torch.stack([torch.from_numpy(f) for f in odd]).float())
    f64 = [big[:40].astype(np.float64), big[40:80].astype(np.float64)]
    out = te.stack_counts(f64)
    assert out.dtype == torch.float32 and torch.equal(out, torch.stack([torch.from_numpy(f) for f in f64]).float())
    L = lib.load()
    assert L.mdk_gather_rows(None, None, 0, 16, 4) == lib.MDK_OK      # empty: nothing to do
    assert L.mdk_gather_rows(None, None, 3, 16, 4) == lib.MDK_ERR_ARG


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_install_patches_the_reference_collate_compatibly():
    """`integration.install()` replaces the counts-matrix branch of the reference's own `Batch.collate`
    (what the Batcher thread calls, prediction.py:356-370); every result equals the unpatched one."""
    ref_shim.install()
    import medaka.common as mc
    import medaka.torch_ext as rte
    from medaka_amd import integration
    rng = np.random.default_rng(0)
    big = rng.random((3000, 10), dtype=np.float32)

    def sample(feat, labels=None):
        return mc.Sample(ref_name="c", features=feat, labels=labels, ref_seq=None, positions=None,
                         label_probs=None, depth=None)
    counts = [sample(big[i * 90:i * 90 + 100]) for i in range(9)]
    labelled = [sample(big[i * 100:i * 100 + 100], np.arange(100)) for i in range(3)]
    reads = [sample(rng.integers(0, 5, (6, d, 4)).astype(np.uint8)) for d in (3, 5)]
    before = [rte.Batch.collate(x) for x in (counts, labelled, reads)]
    integration.install(collate=True)
    try:
        assert rte.Batch.collate.__func__ is not integration._ORIG["collate"].__func__
        after = [rte.Batch.collate(x) for x in (counts, labelled, reads)]
    finally:
        integration.uninstall()
    for a, b in zip(before, after):
        assert type(a) is type(b) is rte.Batch
        for field in ("counts_matrix", "read_level_features", "labels", "majority_vote_probs"):
            va, vb = getattr(a, field), getattr(b, field)
            assert (va is None) == (vb is None), field
            if va is not None:
                assert va.dtype == vb.dtype and torch.equal(va, vb), field
    assert rte.Batch.collate.__func__.__qualname__ == "Batch.collate"      # restored
    with pytest.raises(IndexError):
        rte.Batch.collate([])                                             # the reference's own error, unpatched


# ---- the model swap (integration.convert): every family, against the real classes and their stand-ins ----
import json  # noqa: E402

import ref_standins  # noqa: E402


def _ref_keys():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "ref_state_keys.json")))


@pytest.mark.parametrize("name", sorted(ref_standins.CONFIGS))
def test_standins_have_the_reference_surface(name):
    """tests/ref_standins.py (what the GPU box swaps) == the unmodified reference classes: same `to_dict()`, same
    state_dict keys / order / shapes / dtypes (golden from oracle/make_golden_keys.py; live when the tree is here)."""
    cls, kw = ref_standins.CONFIGS[name]
    want = _ref_keys()[name]
    assert ref_standins.describe(cls(**kw)) == want
    if ref_shim.available():
        arch, _, _ = ref_shim.reference_modules()
        live = getattr(arch, cls.__name__)(**kw)
        assert ref_standins.describe(live) == want


def _randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                v.copy_(torch.rand(v.shape, generator=g) * 0.4 + (0.5 if k.endswith("running_var") else -0.2))
            else:
                v.fill_(7)          # num_batches_tracked


def _no_device(monkeypatch):
    """No HIP device here: skip ONLY the move to the device and the engine build of `_finalise`."""
    from medaka_amd import integration
    monkeypatch.setattr(integration, "_finalise", lambda new, dev, ref, strict: new.eval())
    return integration


@pytest.mark.parametrize("source", ["reference", "standin"])
@pytest.mark.parametrize("name", sorted(ref_standins.CONFIGS))
def test_convert_swaps_every_model_family(name, source, monkeypatch):
    """`convert()` must hand back an ENGINE-backed class for GRUModel, LatentSpaceLSTM() and
    both rl_lstm384 variants built by the unmodified reference (datastore.py:135-157 -> models.py:392-400), with
    every tensor carried over -- `read_level_conv.expansion_layer.*` and `num_batches_tracked` included."""
    if source == "reference":
        if not ref_shim.available():
            pytest.skip("reference tree not present")
        arch, _, _ = ref_shim.reference_modules()
        cls, kw = getattr(arch, ref_standins.CONFIGS[name][0].__name__), ref_standins.CONFIGS[name][1]
    else:
        cls, kw = ref_standins.CONFIGS[name]
    integration = _no_device(monkeypatch)
    ref = cls(**kw).eval()
    _randomise(ref, 5)
    new = integration.convert(ref, device="cuda", strict=True)
    assert type(new).__module__ == "medaka_amd.models" and type(new).__name__ == type(ref).__name__
    assert hasattr(new, "engine")
    rs, ns = ref.state_dict(), new.state_dict()
    assert list(rs) == list(ns)
    for k in rs:
        assert torch.equal(rs[k], ns[k]), k
    assert new.to_dict() == ref.to_dict()
    assert not new.training and new.normalise is True and new.half_precision is False
    # half precision set before the swap survives it (prediction.py:164-168 calls half() after the load, but a
    # caller may convert an already-halved model)
    halved = cls(**kw).eval()
    halved.half()                     # (the reference's half() returns None, models.py:298-301)
    new_h = integration.convert(halved, device="cuda", strict=True)
    assert new_h.half_precision is True
    # `--cpu`: the reference model itself comes back, also in strict mode
    assert integration.convert(ref, device="cpu", strict=True) is ref
    assert integration.convert(new, device="cuda") is new          # idempotent


def test_convert_strict_mode_raises_instead_of_falling_back(monkeypatch, caplog):
    """`MEDAKA_AMD=strict` (set by medaka_amd.launch): anything that would keep PyTorch-ROCm's stock path on a
    HIP device is an error; the default mode logs a warning and returns the reference model."""
    integration = _no_device(monkeypatch)
    outside = [ref_standins.GRUModel(gru_size=64), ref_standins.LatentSpaceLSTM(lstm_size=384, bidirectional=True),
               ref_standins.LatentSpaceLSTM(cnn_size=64), ref_standins.LatentSpaceLSTM(kernel_sizes=[1, 9])]

    class SomethingElse(ref_standins._Base):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
    outside.append(SomethingElse())
    for m in outside:
        with caplog.at_level("WARNING", logger="medaka_amd"):
            caplog.clear()
            assert integration.convert(m, device="cuda", strict=False) is m
            assert "keeping the reference model" in caplog.text
        with pytest.raises(integration.EngineRequired):
            integration.convert(m, device="cuda", strict=True)
    monkeypatch.setenv("MEDAKA_AMD", "strict")
    assert integration.strict_from_env()
    with pytest.raises(integration.EngineRequired):
        integration.convert(outside[0], device="cuda")             # strict=None -> environment
    monkeypatch.setenv("MEDAKA_AMD", "1")
    assert not integration.strict_from_env()
    assert integration.convert(outside[0], device="cuda") is outside[0]


def test_convert_engine_rejection_is_loud_in_strict_mode(monkeypatch):
    """The real `_finalise`: an engine that cannot be built (no device here, or a rejected shape) keeps the
    reference model by default and raises in strict mode."""
    from medaka_amd import integration, lib as _lib, models as amd_models

    def boom(self):
        raise _lib.EngineError("mdk_gru_create: bad argument: simulated")
    monkeypatch.setattr(amd_models.GRUModel, "engine", boom)
    monkeypatch.setattr(torch.nn.Module, "to", lambda self, *a, **k: self)
    ref = ref_standins.GRUModel()
    assert integration.convert(ref, device="cuda", strict=False) is ref
    with pytest.raises(integration.EngineRequired, match="simulated"):
        integration.convert(ref, device="cuda", strict=True)


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
@pytest.mark.parametrize("name", ["GRUModel", "rl_lstm384_no_dwells"])
def test_install_swaps_the_model_inside_the_reference_load_model(name, tmp_path, monkeypatch):
    """The whole reference flow of `ModelStoreTGZ.load_model(device=cuda)` (datastore.py:135-157: unpack, the pickled
    `partial(model_from_dict, ...)`, `load_state_dict(torch.load(weights.pt))`, `.to(device)`, `.eval()`) with
    `integration.install()` active in strict mode: what comes back is the engine-backed class with the archive's
    weights.  Only the two steps that need a HIP device are stubbed (`Module.to`, `_finalise`)."""
    import functools
    import pickle
    import tarfile
    ref_shim.install()
    import medaka.architectures as arch
    import medaka.datastore as ds
    import medaka.models as ref_models
    from medaka_amd import integration
    cls, kw = ref_standins.CONFIGS[name]
    ref = getattr(arch, cls.__name__)(**kw)
    _randomise(ref, 9)
    top = tmp_path / "model"
    top.mkdir()
    torch.save(ref.state_dict(), top / "weights.pt")
    with open(top / "meta.pkl", "wb") as fh:
        pickle.dump({"model_function": functools.partial(ref_models.model_from_dict, ref.to_dict())}, fh)
    tgz = tmp_path / "toy_model_pt.tar.gz"
    with tarfile.open(tgz, "w:gz") as tar:
        tar.add(top, arcname="model")
    moved = []
    monkeypatch.setattr(torch.nn.Module, "to", lambda self, *a, **k: (moved.append(a), self)[1])
    monkeypatch.setattr(integration, "_finalise", lambda new, dev, ref_model, strict: new.eval())
    monkeypatch.setenv("MEDAKA_AMD", "strict")
    integration.install(collate=False)
    try:
        with ds.ModelStoreTGZ(str(tgz)) as store:
            model = store.load_model(device=torch.device("cuda"))
            assert store.model is model
    finally:
        integration.uninstall()
    assert type(model).__module__ == "medaka_amd.models" and type(model).__name__ == cls.__name__
    assert moved and str(moved[0][0]) == "cuda"                      # the reference moved its own model first
    for k, v in ref.state_dict().items():
        assert torch.equal(model.state_dict()[k], v), k
    assert model.to_dict() == ref.to_dict()


def test_validate_plan_only_on_a_state_dict_and_on_a_reference_archive(tmp_path, capsys):
    """`python -m medaka_amd.validate <model> --plan-only`: the device-free half of the one-command
    check for real model archives -- an .npz state dict, and a .tar.gz through the UNMODIFIED reference's
    `ModelStoreTGZ.load_model` (the flow a real archive takes)."""
    import functools
    import pickle
    import tarfile
    from medaka_amd import validate
    rep = validate.main([os.path.join(ROOT, "tests", "golden", "weights_trained.npz"), "--plan-only", "--batch", "100"])
    assert rep["model"]["class"] == "GRUModel" and rep["model"]["engine_covers_it"]
    assert rep["model"]["split_plan_at_margin_128"] == {"batch": 100, "columns": 10000, "chunks": 10, "virtual_columns": 1264, "margin": 128}
    plan = rep["model"]["pass_plan"]           # how those passes are launched (mdk_pass_plan): the split call and its audit without gi
    assert plan["forward"]["work_groups"] == 125 and plan["forward"]["fuse_projection"] and not plan["forward"]["needs_gi"]
    assert plan["audit_scan"]["fuse_projection"] and not plan["audit_scan"]["needs_gi"] and plan["workspace_GB"] < 3.0
    ref_shim.install()
    import medaka.architectures as arch
    import medaka.models as ref_models
    ref = arch.GRUModel(num_features=10, gru_size=128, n_layers=2, bidirectional=True)
    top = tmp_path / "model"
    top.mkdir()
    torch.save(ref.state_dict(), top / "weights.pt")
    with open(top / "meta.pkl", "wb") as fh:
        pickle.dump({"model_function": functools.partial(ref_models.model_from_dict, ref.to_dict())}, fh)
    tgz = tmp_path / "toy_model_pt.tar.gz"
    with tarfile.open(tgz, "w:gz") as tar:
        tar.add(top, arcname="model")
    rep = validate.main([str(tgz), "--plan-only", "--json", str(tmp_path / "r.json")])
    assert rep["model"]["source"] == "ModelStoreTGZ.load_model" and rep["model"]["class"] == "GRUModel" and rep["model"]["engine_covers_it"]
    assert rep["model"]["pass_plan"]["forward"]["windows_per_group"] == 8
    assert json.load(open(tmp_path / "r.json"))["plan_only"]
    capsys.readouterr()


def _build_c_host(tmp_path):
    import subprocess
    from medaka_amd import build as _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = _build.build()
    exe = os.path.join(str(tmp_path), "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "abi_smoke.c"), "-L", os.path.dirname(so),
                           "-lmedaka_amd", "-lm", "-Wl,-rpath," + os.path.dirname(so), "-o", exe])
    return exe


def test_c_host_compiles_and_links_against_the_header(tmp_path):
    """include/medaka_amd.h is plain C99 and the library's entry points resolve from a C program
    (the reference-side binding could equally be cffi, which medaka already uses for libmedaka)."""
    exe = _build_c_host(tmp_path)
    assert os.path.getsize(exe) > 0


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirror of every struct of include/medaka_amd.h (medaka_amd/lib.py) has the size and the field offsets
    gcc gives the C declaration: a field added on one side only cannot go unnoticed."""
    import ctypes
    import subprocess
    from medaka_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = {"mdk_gru_desc": lib.GruDesc, "mdk_rl_desc": lib.RlDesc, "mdk_rl_timing": lib.RlTiming,
             "mdk_gru_timing": lib.GruTiming, "mdk_gru_split": lib.GruSplit, "mdk_split_shape": lib.SplitShape,
             "mdk_pass_shape": lib.PassShape}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "medaka_amd.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf(" {fname}:%zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", exe])
    out = subprocess.check_output([exe], text=True).strip().split("\n")
    assert len(out) == len(pairs)
    for line in out:
        cname, size, *fields = line.split()
        cls = pairs[cname]
        assert int(size) == ctypes.sizeof(cls), cname
        assert len(fields) == len(cls._fields_), cname
        for item in fields:
            fname, off = item.split(":")
            assert getattr(cls, fname).offset == int(off), (cname, fname)
