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

Interface glue that necessarily reads like the reference's: `TorchModel.device()` and `TorchModel.to_dict()` below follow
medaka/models.py:291-296 and :347-365 line by line (about twenty lines): `to_dict()` has to produce the very dict
`model_from_dict` and the model store round-trip, and subclassing the reference classes instead is impossible where
medaka is not installed (the GPU test box).  Everything else in this file is this repository's own.
"""
import inspect
import logging
import os
import warnings

import numpy as np
import torch

from medaka_amd import engine as _engine
from medaka_amd import lib as _lib


def _host_output(shape, dtype=torch.float32):
    """CPU tensor for `predict_on_batch` to return (models.py:312 `.cpu()`), page-locked: torch's
    caching host allocator recycles the block once every view the caller keeps (the HDF writer holds the
    per-sample rows, datastore.py:283-300) is gone, so steady-state batches neither allocate nor
    page-fault -- a fresh 40 MB pageable tensor costs 3.6 ms of first-touch faults as a DMA target
    (profiles/r2_host_path_probe.txt)."""
    try:
        return torch.empty(shape, dtype=dtype, pin_memory=True)
    except RuntimeError:
        return torch.empty(shape, dtype=dtype)


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


def _early_start():
    """MEDAKA_AMD_EARLY_START=0: `predict_on_batch` never promises the next call's result buffer, so the engine never starts the
    next batch's forward ahead of its call (include/medaka_amd.h `mdk_gru_forward_pipelined`)."""
    return os.environ.get("MEDAKA_AMD_EARLY_START", "1").strip().lower() not in ("0", "off", "false")


def gpu_share():
    """Processes sharing this process's GPU: `MEDAKA_AMD_PROCS_PER_GPU`, set by `medaka_amd.launch --procs-per-gpu`."""
    try:
        return max(1, min(8, int(os.environ.get("MEDAKA_AMD_PROCS_PER_GPU", "1"))))
    except ValueError:
        return 1


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
            self._engine.set_option("gpu_share", gpu_share())
            from medaka_amd import torch_ext as _te
            _te.set_stage_target(self._engine)       # batches collated from now on are handed over early
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
            from medaka_amd import torch_ext as _te
            staged = _te.staged_is_current(x)                 # (engine, token) left by the engine's Batch.collate, if x is untouched since
            x = x.detach().to(torch.float32).contiguous()
            if x.dim() != 3 or x.shape[2] != self.num_features:
                raise ValueError(f"expected (B, T, {self.num_features}) input, got {tuple(x.shape)}")
            B, T, _ = x.shape
            # the result buffer: the one promised to the engine at the end of the previous call, if any -- the engine may have
            # started THIS batch's forward ahead of this call and be streaming its probabilities into it (mdk_gru_forward_pipelined)
            out = eng.take_promised((B, T, 5))
            if out is None:
                out = _host_output((B, T, 5))
            if staged is not None and staged[0] is eng:
                nxt = _host_output((B, T, 5)) if _early_start() and out.is_pinned() else None
                if nxt is not None and not nxt.is_pinned():
                    nxt = None
                eng.promise(nxt)
                if eng.forward_staged(staged[1], B, T, out.data_ptr(), nxt.data_ptr() if nxt is not None else None):
                    return out                                # x was already on its way: no PCIe wait in this call
                eng.drop_pending()                            # (token no longer valid: nothing may still write to `nxt`)
                eng.promise(None)
            eng.forward_ptr(x.data_ptr(), B, T, out.data_ptr(), host=True)
            return out
        return self.forward(x).detach().cpu()

    # -- opt-in fast paths around the network (no reference counterpart) ----
    def predict_on_counts(self, counts, depth, decoded=False):
        """Raw pileup counts (B,T,10) uint16 + depth of the parent major column (B,T) uint32, host
        arrays/tensors, in place of `CountsFeatureEncoder(normalise='total')` features
        (features.py:907-911): 24 instead of 40 bytes per column cross PCIe.  Returns the (B,T,5)
        float32 probabilities as `predict_on_batch` does, or with `decoded` the pair (argmax class
        uint8 (B,T), its probability float32 (B,T)) that `decode_consensus` (labels.py:1061-1065)
        starts from -- 5 bytes per column back instead of 20 (see engine.decode_consensus)."""
        counts = np.asarray(counts.numpy() if isinstance(counts, torch.Tensor) else counts)
        depth = np.asarray(depth.numpy() if isinstance(depth, torch.Tensor) else depth)
        with torch.inference_mode():
            eng = self.engine()
        if counts.ndim != 3:
            raise ValueError(f"expected counts (B, T, {self.num_features}), got {counts.shape}")
        B, T = counts.shape[:2]
        # results land in page-locked tensors from torch's caching host allocator, as predict_on_batch's do (_host_output)
        if decoded:
            cls, pmax = _host_output((B, T), torch.uint8), _host_output((B, T))
            eng.forward_counts_host(counts, depth, probs=False, decoded=True, out=(cls.numpy(), pmax.numpy()))
            return cls, pmax
        p = _host_output((B, T, 5))
        eng.forward_counts_host(counts, depth, out=(p.numpy(),))
        return p


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


class ReadLevelFeaturesModel(TorchModel):
    """Models taking read level features (base_classes.py:28-34)."""

    def get_model_input_features(self, batch):
        """Return the read level features from the batch (base_classes.py:31-34)."""
        return batch.read_level_features


class _ReversibleLSTM(torch.nn.Module):
    """Parameter container with the reference's naming (latent_space_lstm.py:11-33)."""

    def __init__(self, *args, reverse=False, **kwargs):
        super().__init__()
        self.lstm = torch.nn.LSTM(*args, **kwargs)
        self.reverse = reverse


class _ReadLevelConv(torch.nn.Module):
    """Parameter container mirroring read_level_modules.ReadLevelConv (:45-78)."""

    def __init__(self, in_features, out_dim, kernel_sizes, channel_dim):
        super().__init__()
        layers, in_feat = [], in_features
        for k in kernel_sizes:
            assert k % 2 == 1, "kernel sizes must be odd (for equal & symmetric padding)"
            layers += [torch.nn.Conv1d(in_feat, channel_dim, kernel_size=k, padding=(k - 1) // 2),
                       torch.nn.ReLU(), torch.nn.BatchNorm1d(channel_dim)]
            in_feat = channel_dim
        self.convs = torch.nn.Sequential(*layers)
        self.expansion_layer = torch.nn.Linear(channel_dim, out_dim)   # unused by the reference forward


class LatentSpaceLSTM(ReadLevelFeaturesModel):
    """Read-level model (latent_space_lstm.py:35-207) -- HIP engine behind the reference interface.

    Same constructor, parameter names and `state_dict()` as the reference, so archives load
    unchanged; `forward` is the fused read-level front end + MFMA LSTM stack of the engine.
    Engine limits: cnn_size == 128, kernel_sizes == [1, 17], mean pooling; lstm_size == 128, or
    lstm_size == 384 with bidirectional=False (the bundled `rl_lstm384` models).
    """

    def __init__(self, num_classes=5, lstm_size=128, cnn_size=128, kernel_sizes=[1, 17],
                 pooler_type="mean", pooler_args={}, use_dwells=False, bases_alphabet_size=6,
                 bases_embedding_size=6, bidirectional=True, time_steps=None):
        super().__init__()
        if time_steps is not None:
            warnings.warn("timesteps is no lnoger required to be specified")
        if pooler_type != "mean":
            raise ValueError(f"Unknown PoolerType {pooler_type}")
        self.num_classes = num_classes
        self.lstm_size = lstm_size
        self.cnn_size = cnn_size
        self.kernel_sizes = kernel_sizes
        self.pooler_type = pooler_type
        self.pooler_args = pooler_args
        self.use_dwells = use_dwells
        self.bases_alphabet_size = bases_alphabet_size
        self.bases_embedding_size = bases_embedding_size
        self.base_embedder = torch.nn.Embedding(bases_alphabet_size, bases_embedding_size)
        self.strand_embedder = torch.nn.Embedding(3, bases_embedding_size)
        extra = 2 if use_dwells else 1
        self.read_level_conv = _ReadLevelConv(bases_embedding_size + extra, lstm_size, kernel_sizes, cnn_size)
        self.pre_pool_expansion_layer = torch.nn.Linear(cnn_size, lstm_size)
        self.bidirectional = bidirectional
        if bidirectional:
            self.lstm = torch.nn.LSTM(lstm_size, lstm_size, num_layers=2, bidirectional=True, batch_first=True)
        else:
            self.lstm = torch.nn.Sequential(*[
                _ReversibleLSTM(lstm_size, lstm_size, batch_first=True, reverse=not bool(i % 2))
                for i in range(4)])
        self.linear = torch.nn.Linear((1 + bidirectional) * lstm_size, self.num_classes)
        self.normalise = True
        self._engine = None
        self._engine_key = None

    def check_feature_encoder_compatibility(self, fenc):
        """Check feature encoder is valid for this model (latent_space_lstm.py:209-236)."""
        clsname = type(self).__name__
        if "ReadAlignmentFeatureEncoder" not in {c.__name__ for c in type(fenc).__mro__}:
            raise ValueError(f"{clsname} expects a ReadAlignmentFeatureEncoder.")
        if len(getattr(fenc, "dtypes", ("",))) > 1:
            raise NotImplementedError(f"{clsname} is currently only implemented for one dtype.")
        if self.use_dwells and not getattr(fenc, "include_dwells", False):
            raise ValueError("Model expects dwells, however include_dwells not set in the feature encoder.")

    def engine(self):
        dev_index = _hip_device_index(self.device())
        if self.lstm_size == 384 and gpu_share() > 1:
            raise _lib.EngineError(
                "LatentSpaceLSTM(lstm_size=384): the cluster recurrence keeps one work-group on each of 192 CUs for "
                f"a whole layer and cannot share its GPU (MEDAKA_AMD_PROCS_PER_GPU={gpu_share()}); run one process "
                "per GPU for the rl_lstm384 models")
        key = (dev_index, tuple((p.data_ptr(), p._version, p.dtype) for p in self.parameters()),
               tuple((b.data_ptr(), b._version) for b in self.buffers()))
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            state = {k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()
                     if "num_batches_tracked" not in k}
            self._engine = _engine.RlEngine(
                state, use_dwells=self.use_dwells, bidirectional=self.bidirectional,
                lstm_size=self.lstm_size, cnn_size=self.cnn_size, kernel_sizes=self.kernel_sizes,
                alphabet_size=self.bases_alphabet_size, embedding_size=self.bases_embedding_size,
                normalise=bool(self.normalise), device=dev_index, num_classes=self.num_classes)
            self._engine_key = key
        self._engine.set_precision(self.half_precision)
        self._engine.set_normalise(bool(self.normalise))
        return self._engine

    def forward(self, x):
        """x: uint8 (B, P, D, F) on the model's device -> (B, P, 5) float32 on the same device."""
        eng = self.engine()
        if x.device != self.device():
            raise RuntimeError(f"input on {x.device}, model on {self.device()}")
        if x.dim() != 4:
            raise ValueError(f"expected (B, P, D, F) input, got {tuple(x.shape)}")
        x = x.detach().to(torch.uint8).contiguous()
        B, P, D, F = x.shape
        out = torch.empty((B, P, 5), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        eng.forward_ptr(x.data_ptr(), B, P, D, F, out.data_ptr(), stream=stream)
        return out

    def _predict(self, x):
        if x.device.type == "cpu":
            eng = self.engine()
            if x.dim() != 4:
                raise ValueError(f"expected (B, P, D, F) input, got {tuple(x.shape)}")
            x = x.detach().to(torch.uint8).contiguous()
            B, P, D, F = x.shape
            out = _host_output((B, P, 5))          # page-locked, recycled by torch's host allocator
            eng.forward_ptr(x.data_ptr(), B, P, D, F, out.data_ptr(), host=True)
            return out
        return self.forward(x).detach().cpu()


ARCHITECTURES = {"GRUModel": GRUModel, "MajorityVoteModel": MajorityVoteModel,
                 "LatentSpaceLSTM": LatentSpaceLSTM}


def model_from_dict(model_dict):
    """Create a model from a {"type", "kwargs"} dict (reference models.py:392-400)."""
    try:
        cls = ARCHITECTURES[model_dict["type"]]
    except KeyError as e:
        raise ValueError(f"unknown model type {model_dict.get('type')!r}; the MI355X engine "
                         f"provides {sorted(ARCHITECTURES)}") from e
    return cls(**model_dict["kwargs"])
