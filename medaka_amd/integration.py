"""Swap the MI355X engine into an installed medaka without touching its sources.

Reference flow (medaka/prediction.py:113-168):
    model_store = models.open_model(args.model)
    model = model_store.load_model(device=device)     # datastore.py:135-157
    model.check_feature_encoder_compatibility(fenc); model.half(); run_prediction(...)

`install()` wraps `ModelStoreTGZ.load_model`: the reference builds and loads its own model
exactly as before; if the result is a `GRUModel` the engine supports AND the target device is
a HIP device, an engine-backed `medaka_amd.models.GRUModel` with the same `state_dict()` is
returned instead.  `medaka inference --cpu` and unsupported shapes keep
the reference implementation -- the engine itself never runs on the CPU.  `LatentSpaceLSTM`
models with cnn_size = 128 and lstm_size = 128, or lstm_size = 384 uni-directional (the bundled
`rl_lstm384`), are accelerated too.

Opt in with `MEDAKA_AMD=1` in the environment of `medaka inference` or by
calling `install()` before `medaka.prediction.predict(args)`.  `MEDAKA_AMD=strict` additionally turns every
"keeping the reference model" decision on a HIP device into an `EngineRequired` error -- the launcher
(`medaka_amd.launch`) uses it, so a multi-GPU job can never run PyTorch-ROCm's stock RNN path unnoticed.
"""
import functools
import logging
import os

_ORIG = {}
logger = logging.getLogger("medaka_amd")


def _gru_supported(model):
    """Every limit of mdk_gru_create and of the production kernels (include/medaka_amd.h), so that an
    unsupported archive keeps the reference model instead of failing at the first forward."""
    return (getattr(model, "gru_size", None) == 128 and 1 <= getattr(model, "num_features", 10) <= 16
            and 1 <= getattr(model, "n_layers", 2) <= 4)


def _rl_supported(model):
    lstm_size = getattr(model, "lstm_size", None)
    return (getattr(model, "cnn_size", None) == 128
            and (lstm_size == 128 or (lstm_size == 384 and not getattr(model, "bidirectional", True)))
            and list(getattr(model, "kernel_sizes", [])) == [1, 17]
            and getattr(model, "bases_embedding_size", 6) == 6 and getattr(model, "bases_alphabet_size", 6) <= 8
            and getattr(model, "num_classes", 5) == 5 and getattr(model, "pooler_type", "mean") == "mean")


class EngineRequired(RuntimeError):
    """Strict mode (`MEDAKA_AMD=strict`): the reference model would have been kept."""


def strict_from_env():
    """`MEDAKA_AMD=strict` (what `medaka_amd.launch` sets for its children): a model the engine cannot take
    is an error, not a quiet return to PyTorch-ROCm's stock RNN path."""
    return os.environ.get("MEDAKA_AMD", "0").strip().lower() == "strict"


def _keep_reference(model, why, strict):
    """The one place a reference model is handed back for a HIP device: loud in strict mode."""
    name = type(model).__name__
    if strict:
        raise EngineRequired(f"medaka_amd (strict): {name} stays on the reference implementation: {why}")
    logger.warning("medaka_amd: %s (%s), keeping the reference model", name, why)
    return model


def _finalise(new, dev, reference_model, strict):
    """Move the engine-backed model to the device and build the C-ABI engine NOW, so that anything the
    engine rejects shows up at load time and not at the first batch."""
    from medaka_amd import lib as _lib
    new = new.to(dev).eval()
    try:
        new.engine()
    except (_lib.EngineError, ValueError, KeyError) as e:
        return _keep_reference(reference_model, f"engine rejected it: {e}", strict)
    logger.info("medaka_amd: %s -> %s.%s on %s", type(reference_model).__name__, type(new).__module__,
                type(new).__name__, dev)
    return new


def _copy_state(new, model):
    """Every tensor of the reference model's state_dict, by name, strictly: the mirrors in
    medaka_amd.models declare the same parameters AND buffers as the reference classes (the unused
    `read_level_conv.expansion_layer` and the batch-norm `num_batches_tracked` counters included)."""
    new.load_state_dict(model.state_dict(), strict=True)
    new.normalise = getattr(model, "normalise", True)
    if getattr(model, "half_precision", False):
        new.half()
    return new


def convert(model, device=None, strict=None):
    """Return an engine-backed equivalent of a reference model (reference flow: the object
    `ModelStoreTGZ.load_model` returns, datastore.py:135-157, built by `model_from_dict`, models.py:392-400).

    Non-HIP devices (`medaka inference --cpu`) keep the reference model.  On a HIP device a model the engine
    does not cover keeps the reference implementation with a warning -- or, with `strict` (default: the
    `MEDAKA_AMD=strict` environment), raises `EngineRequired`."""
    import torch
    from medaka_amd import models as amd_models

    strict = strict_from_env() if strict is None else bool(strict)
    name = type(model).__name__
    dev = torch.device(device) if device is not None else model.device()
    if dev.type != "cuda":
        return model
    if isinstance(model, (amd_models.GRUModel, amd_models.MajorityVoteModel, amd_models.LatentSpaceLSTM)):
        return model
    if name == "GRUModel":
        if not _gru_supported(model):
            return _keep_reference(model, "outside the engine's envelope (gru_size 128, 1-4 layers, <= 16 features)", strict)
        kwargs = model.to_dict()["kwargs"]
        kwargs.pop("time_steps", None)
        kwargs.pop("classify_activation", None)
        try:
            new = _copy_state(amd_models.GRUModel(**kwargs), model)
        except (RuntimeError, TypeError, ValueError) as e:
            return _keep_reference(model, f"state_dict mismatch: {e}", strict)
        return _finalise(new, dev, model, strict)
    if name == "LatentSpaceLSTM":
        if not _rl_supported(model):
            return _keep_reference(model, "outside the engine's envelope (cnn_size 128, kernel_sizes [1, 17], mean pooling, "
                                          "lstm_size 128 or uni-directional 384)", strict)
        kwargs = model.to_dict()["kwargs"]
        kwargs.pop("time_steps", None)
        try:
            new = _copy_state(amd_models.LatentSpaceLSTM(**kwargs), model)
        except (RuntimeError, TypeError, ValueError) as e:
            return _keep_reference(model, f"state_dict mismatch: {e}", strict)
        return _finalise(new, dev, model, strict)
    if name == "MajorityVoteModel":
        return amd_models.MajorityVoteModel().to(dev).eval()
    return _keep_reference(model, "no engine for this architecture", strict)


def _fast_collate(orig):
    """`Batch.collate` of the reference (torch_ext.py:110-173) with its counts-matrix branch assembled by
    `medaka_amd.torch_ext.stack_counts`; every other case (read-level features, labels, odd dtypes) is the
    reference's own code.  Same class, same fields, same values."""
    from medaka_amd.torch_ext import stack_counts

    def collate(cls, samples, counts_matrix=False):
        first = samples[0] if len(samples) else None
        if first is not None and getattr(first.features, "ndim", 0) == 2 and getattr(first, "labels", None) is None:
            return cls(counts_matrix=stack_counts([s.features for s in samples]))
        return orig(cls, samples, counts_matrix)
    return collate


def install(collate=None):
    """Patch `medaka.datastore.ModelStoreTGZ.load_model` (the model swap) and -- unless `collate` is False or
    `MEDAKA_AMD_COLLATE=0` -- `medaka.torch_ext.Batch.collate` (batch assembly in the Batcher thread,
    prediction.py:356-370).  Idempotent."""
    import medaka.datastore as ds

    if "load_model" not in _ORIG:
        orig = ds.ModelStoreTGZ.load_model

        @functools.wraps(orig)
        def load_model(self, time_steps=None, device=None, *args, **kwargs):
            model = orig(self, time_steps=time_steps, device=device, *args, **kwargs)
            self.model = convert(model, device)
            return self.model

        _ORIG["load_model"] = orig
        ds.ModelStoreTGZ.load_model = load_model
    if collate is None:
        collate = os.environ.get("MEDAKA_AMD_COLLATE", "1").strip().lower() not in ("0", "off", "false")
    if collate and "collate" not in _ORIG:
        import medaka.torch_ext as rte
        _ORIG["collate"] = rte.Batch.__dict__["collate"]           # the classmethod object itself
        rte.Batch.collate = classmethod(_fast_collate(_ORIG["collate"].__func__))


def uninstall():
    if "load_model" in _ORIG:
        import medaka.datastore as ds
        ds.ModelStoreTGZ.load_model = _ORIG.pop("load_model")
    if "collate" in _ORIG:
        import medaka.torch_ext as rte
        rte.Batch.collate = _ORIG.pop("collate")


def install_from_env():
    """`MEDAKA_AMD=1` -> install(); used by the sitecustomize hook of INTEGRATION.md."""
    if os.environ.get("MEDAKA_AMD", "0").strip().lower() not in ("", "0", "off", "false"):
        install()
        return True
    return False
