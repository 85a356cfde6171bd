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
    if name == "GRUModel" and getattr(model, "gru_size", None) == 128:
        kwargs = model.to_dict()["kwargs"]
        kwargs.pop("time_steps", None)
        kwargs.pop("classify_activation", None)
        new = amd_models.GRUModel(**kwargs)
        new.load_state_dict(model.state_dict())
        new.normalise = getattr(model, "normalise", True)
        if getattr(model, "half_precision", False):
            new.half()
        return new.to(dev).eval()
    lstm_size = getattr(model, "lstm_size", None)
    if (name == "LatentSpaceLSTM" and getattr(model, "cnn_size", None) == 128
            and (lstm_size == 128 or (lstm_size == 384 and not getattr(model, "bidirectional", True)))
            and list(getattr(model, "kernel_sizes", [])) == [1, 17]):
        kwargs = model.to_dict()["kwargs"]
        kwargs.pop("time_steps", None)
        new = amd_models.LatentSpaceLSTM(**kwargs)
        new.load_state_dict(model.state_dict(), strict=False)
        new.normalise = getattr(model, "normalise", True)
        if getattr(model, "half_precision", False):
            new.half()
        return new.to(dev).eval()
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
