"""Boundary object of the hot path: mirror of reference `medaka.torch_ext.Batch`.

Same field names and `collate` semantics as reference medaka/torch_ext.py:102-173 so that code
written against the reference (`model.predict_on_batch(Batch(counts_matrix=x))`) runs unchanged
when medaka itself is not installed (e.g. on the GPU test box).  When medaka is installed the
engine accepts the reference's own `Batch` -- only attribute access is used.
"""
import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch

# host threads of the batch assembly (MEDAKA_AMD_COLLATE_THREADS; the reference's Batcher is one thread)
def _collate_threads():
    try:
        return max(1, min(64, int(os.environ.get("MEDAKA_AMD_COLLATE_THREADS", "4"))))
    except ValueError:
        return 4


COLLATE_THREADS = _collate_threads()


def _batch_buffer(shape, dtype):
    """Page-locked when a HIP device is there (recycled by torch's caching host allocator once the batch is
    dropped: no page faults, DMA-able without staging), ordinary memory otherwise."""
    if torch.cuda.is_available():
        try:
            return torch.empty(shape, dtype=dtype, pin_memory=True)
        except RuntimeError:
            pass
    return torch.empty(shape, dtype=dtype)


# Early hand-over (include/medaka_amd.h `mdk_gru_stage_input`): the engine whose model will see the batches registers itself
# here (models.GRUModel.engine()); `stack_counts` then starts the batch's host -> device copy from the Batcher thread, so
# that it crosses PCIe while the main thread is still inside `predict_on_batch` of the previous batch.  The token rides on
# the tensor (`_mdk_stage`); `predict_on_batch` redeems it.  MEDAKA_AMD_STAGE=0 turns it off.
_stage_target = None


def set_stage_target(engine):
    """`engine`: a medaka_amd.engine.GruEngine (kept by weak reference) or None."""
    global _stage_target
    import weakref
    _stage_target = weakref.ref(engine) if engine is not None else None


def forget_stage_target(engine):
    """`engine` is being closed: no further batch is handed to it."""
    global _stage_target
    if _stage_target is not None and _stage_target() is engine:
        _stage_target = None


def _stage(out):
    ref = _stage_target               # read ONCE: close() on the main thread may clear the global under this (Batcher) thread
    if ref is None or os.environ.get("MEDAKA_AMD_STAGE", "1") == "0" or not out.is_pinned():
        return
    try:
        eng = ref()
        if eng is None or out.dim() != 3 or out.shape[2] != eng.num_features:
            return
        tok = eng.stage_input(out.data_ptr(), out.shape[0], out.shape[1])
        if tok:
            # what was copied: this tensor, these bytes.  `predict_on_batch` redeems the token only if the tensor is still the
            # one that was handed over -- same storage, same shape, no in-place edit since (staged_is_current)
            out._mdk_stage = (eng, tok, out._version, out.data_ptr(), tuple(out.shape))
    except Exception:        # staging is an optimisation: the ordinary path answers
        pass


def staged_is_current(x):
    """The (engine, token) a tensor carries from `_stage`, if the device copy still is the tensor's content: nothing wrote to
    it in place since the hand-over (torch's version counter), same storage, same shape.  None otherwise -- the ordinary
    host path then copies what the tensor holds NOW."""
    st = getattr(x, "_mdk_stage", None)
    if st is None:
        return None
    try:
        eng, tok, ver, ptr, shape = st
        if x._version == ver and x.data_ptr() == ptr and tuple(x.shape) == shape:
            return eng, tok
    except Exception:
        pass
    return None


def stack_counts(feats, threads=None):
    """`torch.stack([torch.from_numpy(f) for f in feats]).float()` (reference torch_ext.py:147-148) without the
    17 ms it costs per 200 x 10000 x 10 batch on one thread into fresh pageable memory: equal-shaped,
    C-contiguous float32 arrays are gathered into one (page-locked) buffer by `mdk_gather_rows` on a few
    host threads (GIL released).  Anything else takes the reference's own expression.  Same values, dtype and shape."""
    first = feats[0]
    plain = all(isinstance(f, np.ndarray) and f.dtype == np.float32 and f.shape == first.shape
                and f.flags.c_contiguous for f in feats)
    if not plain or first.size == 0:
        return torch.stack([torch.from_numpy(np.asarray(f)) for f in feats]).float()
    from medaka_amd import lib as _lib
    out = _batch_buffer((len(feats),) + first.shape, torch.float32)
    rows = (ctypes.c_void_p * len(feats))(*[f.ctypes.data for f in feats])
    _lib.check(_lib.load().mdk_gather_rows(out.data_ptr(), rows, len(feats), first.nbytes,
                                           threads or COLLATE_THREADS), "mdk_gather_rows")
    _stage(out)
    return out


@dataclass
class Batch:
    """Batch of samples (reference torch_ext.py:102-109)."""

    read_level_features: torch.Tensor = None
    counts_matrix: torch.Tensor = None
    labels: torch.Tensor = None
    majority_vote_probs: torch.Tensor = None

    @classmethod
    def collate(cls, samples, counts_matrix=False):
        """Construct a batch from `Sample`-like objects (reference torch_ext.py:110-173).

        2-d `features` (columns x 10) become the float32 `counts_matrix` (B, T, 10) -- assembled by
        `stack_counts` (multi-threaded gather into a page-locked buffer);
        3-d read-level features are zero-padded to the maximum depth as uint8 (B, P, D, F).
        """
        if len(samples) == 0:
            raise ValueError("cannot collate an empty list of samples")
        feats = [np.asarray(s.features) for s in samples]
        d = {}
        if feats[0].ndim == 3:
            depths = [f.shape[1] for f in feats]
            npos, _, nfeats = feats[0].shape
            out = np.zeros((len(samples), npos, max(depths), nfeats), dtype=np.uint8)
            for i, f in enumerate(feats):
                out[i, :, :depths[i], :] = f
            d["read_level_features"] = torch.from_numpy(out)
            if counts_matrix:
                d["counts_matrix"] = torch.stack(
                    [torch.from_numpy(np.asarray(s.counts_matrix)) for s in samples]).float()
        elif feats[0].ndim == 2:
            d["counts_matrix"] = stack_counts(feats)
        else:
            raise ValueError(
                f"Unknown feature dimension {feats[0].ndim}. Expect 3 for"
                "read level features or 2 for counts matrices.")
        # reference quirk kept: majority_vote_probs is never filled at collate time
        # (getattr on a dict, torch_ext.py:154-159)
        if getattr(samples[0], "labels", None) is not None:
            d["labels"] = torch.stack([torch.from_numpy(np.asarray(s.labels)) for s in samples])
        return cls(**d)

    @property
    def features(self):
        """Return the features tensor (reference torch_ext.py:167-172)."""
        if self.read_level_features is None:
            return self.counts_matrix
        return self.read_level_features
