"""Swap the MI355X engine into an installed medaka without touching its sources.

Reference flow (medaka/prediction.py:113-168):
    model_store = models.open_model(args.model)
    model = model_store.load_model(device=device)     # datastore.py:135-157
    model.check_feature_encoder_compatibility(fenc); model.half(); run_prediction(...)

`install()` wraps `ModelStoreTGZ.load_model`: the reference builds and loads its own model
exactly as before; if the result is a `GRUModel` the engine supports AND the target device is
a HIP device, an engine-backed `medaka_amd.models.GRUModel` with the same `state_dict()` is
returned instead.  `medaka inference --cpu`, read-level models and unsupported shapes keep
the reference implementation -- the engine itself never runs on the CPU.  `LatentSpaceLSTM`
models with cnn_size = 128 and lstm_size = 128, or lstm_size = 384 uni-directional (the bundled
`rl_lstm384`), are accelerated too.

Opt in with `MEDAKA_AMD=1` in the environment of `medaka inference` or by
calling `install()` before `medaka.prediction.predict(args)`.
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


def _engine_or_reference(new, reference_model):
    """Build the C-ABI engine now; anything it rejects keeps the reference model (docstring promise)."""
    from medaka_amd import lib as _lib
    try:
        new.engine()
    except (_lib.EngineError, ValueError, KeyError) as e:
        logger.warning("medaka_amd: engine rejected %s (%s), keeping the reference model",
                       type(reference_model).__name__, e)
        return reference_model
    return new


def convert(model, device=None):
    """Return an engine-backed equivalent of a reference model, or the model itself."""
    import torch
    from medaka_amd import models as amd_models

    name = type(model).__name__
    dev = torch.device(device) if device is not None else model.device()
    if dev.type != "cuda":
        return model
    if isinstance(model, (amd_models.GRUModel, amd_models.MajorityVoteModel, amd_models.LatentSpaceLSTM)):
        return model
    if name == "GRUModel" and _gru_supported(model):
        kwargs = model.to_dict()["kwargs"]
        kwargs.pop("time_steps", None)
        kwargs.pop("classify_activation", None)
        new = amd_models.GRUModel(**kwargs)
        new.load_state_dict(model.state_dict(), strict=True)      # same parameter names as the reference
        new.normalise = getattr(model, "normalise", True)
        if getattr(model, "half_precision", False):
            new.half()
        return _engine_or_reference(new.to(dev).eval(), model)
    if name == "LatentSpaceLSTM" and _rl_supported(model):
        kwargs = model.to_dict()["kwargs"]
        kwargs.pop("time_steps", None)
        new = amd_models.LatentSpaceLSTM(**kwargs)
        state = {k: v for k, v in model.state_dict().items()
                 if "num_batches_tracked" not in k and "read_level_conv.expansion_layer" not in k}
        missing = new.load_state_dict(state, strict=False)
        if missing.unexpected_keys or any("num_batches_tracked" not in k for k in missing.missing_keys):
            logger.warning("medaka_amd: state_dict mismatch (%s), keeping the reference model", missing)
            return model
        new.normalise = getattr(model, "normalise", True)
        if getattr(model, "half_precision", False):
            new.half()
        return _engine_or_reference(new.to(dev).eval(), model)
    if name == "MajorityVoteModel":
        return amd_models.MajorityVoteModel().to(dev).eval()
    logger.info("medaka_amd: %s is not accelerated, keeping the reference model", name)
    return model


def install():
    """Patch `medaka.datastore.ModelStoreTGZ.load_model` (idempotent)."""
    import medaka.datastore as ds

    if "load_model" in _ORIG:
        return
    orig = ds.ModelStoreTGZ.load_model

    @functools.wraps(orig)
    def load_model(self, time_steps=None, device=None, *args, **kwargs):
        model = orig(self, time_steps=time_steps, device=device, *args, **kwargs)
        self.model = convert(model, device)
        return self.model

    _ORIG["load_model"] = orig
    ds.ModelStoreTGZ.load_model = load_model


def uninstall():
    if "load_model" in _ORIG:
        import medaka.datastore as ds
        ds.ModelStoreTGZ.load_model = _ORIG.pop("load_model")


def install_from_env():
    """`MEDAKA_AMD=1` -> install(); used by the sitecustomize hook of INTEGRATION.md."""
    if os.environ.get("MEDAKA_AMD", "0") not in ("", "0"):
        install()
        return True
    return False
