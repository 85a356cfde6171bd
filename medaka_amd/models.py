"""Model API of the hot path: mirrors of the reference model classes, backed by the HIP engine.

Reference interface kept (names, argument meaning, error behaviour):
  * `TorchModel`            medaka/models.py:277-365  (predict_on_batch, half, device, to_dict)
  * `CountsMatrixModel`     medaka/architectures/base_classes.py:6-25
  * `GRUModel`              medaka/architectures/gru.py:10-72
  * `MajorityVoteModel`     medaka/architectures/majority_vote_model.py:16-53
  * `model_from_dict`       medaka/models.py:392-400

`GRUModel` here is a `torch.nn.Module` with the SAME parameter names as the reference
(`gru.weight_ih_l0` ... `linear.bias`), so `ModelStoreTGZ.load_model`'s
`load_state_dict(torch.load(weights.pt))` / `.to(device)` / `.eval()` / `.half()` sequence
(medaka/datastore.py:135-157, medaka/prediction.py:150-168) works unchanged -- but its
`forward` is the hand-written HIP engine, not torch.nn.GRU.  There is no CPU path: a model
whose parameters are not on a HIP device refuses to predict (use the reference class for
`medaka inference --cpu`).
"""
import inspect
import logging
import warnings

import torch

from medaka_amd import engine as _engine
from medaka_amd import lib as _lib


class TorchModel(torch.nn.Module):
    """Base class mirroring reference `medaka.models.TorchModel` (models.py:277-365)."""

    def __init__(self):
        super().__init__()
        self.half_precision = False
        self.logger = logging.getLogger("TorchModel")

    def device(self):
        """Device where model has been loaded (models.py:291-296)."""
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def half(self):
        """Set model to half precision (models.py:298-301)."""
        super().half()
        self.half_precision = True
        return self

    def get_model_input_features(self, batch):
        raise NotImplementedError

    def predict_on_batch(self, batch):
        """Run inference on a feature batch; returns a cpu tensor (models.py:303-313)."""
        x = self.get_model_input_features(batch)
        with torch.inference_mode():
            return self._predict(x)

    def process_batch(self, batch, loss_fn):
        raise NotImplementedError(
            "training (TorchModel.process_batch, models.py:315-345) is outside the MI355X "
            "inference engine; train with the reference classes and load the weights here")

    def to_dict(self):
        """Return a dict of the model name and args (models.py:347-365)."""
        kwargs = inspect.signature(self.__class__.__init__).parameters
        out_kwargs = {}
        for k, v in kwargs.items():
            if k == "self":
                continue
            elif hasattr(self, k):
                out_kwargs[k] = getattr(self, k)
            elif v.default != inspect.Parameter.empty:
                out_kwargs[k] = v.default
            else:
                raise ValueError(f"Model parameter {k} not set, Cannot serialise model.")
        return {"type": self.__class__.__name__, "kwargs": out_kwargs}


_VALID_COUNTS_FENCS = ("CountsFeatureEncoder", "ReadAlignmentFeatureEncoder")


class CountsMatrixModel(TorchModel):
    """Models taking counts matrices (base_classes.py:6-25)."""

    def get_model_input_features(self, batch):
        """Return the counts matrix from the batch (base_classes.py:9-11)."""
        return batch.counts_matrix

    def check_feature_encoder_compatibility(self, fenc):
        """Check feature encoder is valid for this model (base_classes.py:13-25)."""
        names = {c.__name__ for c in type(fenc).__mro__}
        if not names.intersection(_VALID_COUNTS_FENCS):
            clsname = type(self).__name__
            raise ValueError(f"{type(fenc)} is not a valid feature encoder for {clsname}.")


def _hip_device_index(dev):
    if dev.type != "cuda":
        raise RuntimeError(
            f"medaka_amd model is on '{dev}': the MI355X engine has no CPU path. Move the model "
            "to a HIP device (`model.to('cuda')`) or use the reference model for --cpu runs.")
    return dev.index if dev.index is not None else torch.cuda.current_device()


class GRUModel(CountsMatrixModel):
    """Bidirectional GRU on counts matrix -- HIP engine behind the reference interface."""

    def __init__(self, num_features=10, num_classes=5, gru_size=128, n_layers=2,
                 bidirectional=True, time_steps=None, classify_activation=None):
        super().__init__()
        if time_steps is not None:
            warnings.warn("timesteps is no lnoger required to be specified")
        if classify_activation is not None:
            warnings.warn("classify_activation is no longer used")
        self.gru_size = gru_size
        self.num_classes = num_classes
        self.num_features = num_features
        self.n_layers = n_layers
        self.bidirectional = bidirectional
        # parameter containers only (names/shapes/initialisation as gru.py:46-55);
        # their torch forward is never called
        self.gru = torch.nn.GRU(num_features, gru_size, num_layers=n_layers,
                                bidirectional=bidirectional, batch_first=True)
        self.linear = torch.nn.Linear(2 * gru_size if bidirectional else gru_size, 5)
        self.normalise = True
        self._engine = None
        self._engine_key = None
        self.exact_kernels = False   # MDK_VARIANT_EXACT (debug cross-check kernels)
        self.kernel_variant = None   # explicit MDK_VARIANT_* override (A/B timing)

    # -- engine life cycle -----------------------------------------------------------------
    def _state_key(self, dev_index):
        return (dev_index, tuple((p.data_ptr(), p._version, p.dtype) for p in self.parameters()))

    def engine(self):
        """(Re)build the C-ABI engine from the current parameters, lazily (fork-safe: no
        device work happens at import or construction time)."""
        dev_index = _hip_device_index(self.device())
        key = self._state_key(dev_index)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            state = {k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()}
            self._engine = _engine.GruEngine(
                state, num_features=self.num_features, gru_size=self.gru_size,
                n_layers=self.n_layers, bidirectional=self.bidirectional, num_classes=5,
                normalise=bool(self.normalise), device=dev_index)
            self._engine_key = key
        self._engine.set_precision(self.half_precision)
        self._engine.set_variant(self.kernel_variant if self.kernel_variant is not None
                                 else int(self.exact_kernels))
        self._engine.set_normalise(bool(self.normalise))
        return self._engine

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x):
        """Model forward pass (gru.py:58-72): x (B, T, F) on the model's device ->
        (B, T, 5) float32 on the same device."""
        eng = self.engine()
        dev = self.device()
        if x.device != dev:
            raise RuntimeError(f"input on {x.device}, model on {dev}")
        if x.dim() != 3 or x.shape[2] != self.num_features:
            raise ValueError(f"expected (B, T, {self.num_features}) input, got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        B, T, _ = x.shape
        out = torch.empty((B, T, 5), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        eng.forward_ptr(x.data_ptr(), B, T, out.data_ptr(), stream=stream)
        return out

    def _predict(self, x):
        if x.device.type == "cpu":
            # host tensor in -> host tensor out through the engine's own staging
            # (models.py:309-312 does .to(device) ... .cpu())
            eng = self.engine()
            x = x.detach().to(torch.float32).contiguous()
            if x.dim() != 3 or x.shape[2] != self.num_features:
                raise ValueError(f"expected (B, T, {self.num_features}) input, got {tuple(x.shape)}")
            B, T, _ = x.shape
            out = torch.empty((B, T, 5), dtype=torch.float32)
            eng.forward_ptr(x.data_ptr(), B, T, out.data_ptr(), host=True)
            return out
        return self.forward(x).detach().cpu()


class MajorityVoteModel(CountsMatrixModel):
    """Majority vote of the pileup (majority_vote_model.py:16-53) on the device."""

    def __init__(self, time_steps=None, **kwargs):
        super().__init__()
        self.num_classes = 5
        if time_steps is not None:
            warnings.warn("timesteps is no longer required to be specified")
        self.dummy_parameter = torch.nn.Parameter(torch.zeros(1, requires_grad=True))
        self.config = {"model_type": "majority_vote", "model_args": {**kwargs}}

    def forward(self, pileup, **kwargs):
        dev_index = _hip_device_index(pileup.device)
        x = pileup.detach().to(torch.float32).contiguous()
        out = torch.empty(x.shape[:-1] + (5,), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.load().mdk_majority_forward_dev(
            x.data_ptr(), x.numel() // 10, out.data_ptr(), dev_index, stream),
            "mdk_majority_forward_dev")
        return out

    def _predict(self, x):
        if x.device.type == "cpu":
            dev_index = _hip_device_index(self.device())
            import numpy as np
            out = _engine.majority_forward_host(x.detach().float().contiguous().numpy(), dev_index)
            return torch.from_numpy(np.ascontiguousarray(out))
        return self.forward(x).cpu()


ARCHITECTURES = {"GRUModel": GRUModel, "MajorityVoteModel": MajorityVoteModel}


def model_from_dict(model_dict):
    """Create a model from a {"type", "kwargs"} dict (reference models.py:392-400)."""
    try:
        cls = ARCHITECTURES[model_dict["type"]]
    except KeyError as e:
        raise ValueError(f"unknown model type {model_dict.get('type')!r}; the MI355X engine "
                         f"provides {sorted(ARCHITECTURES)}") from e
    return cls(**model_dict["kwargs"])
