"""Boundary object of the hot path: mirror of reference `medaka.torch_ext.Batch`.

Same field names and `collate` semantics as reference medaka/torch_ext.py:102-173 so that code
written against the reference (`model.predict_on_batch(Batch(counts_matrix=x))`) runs unchanged
when medaka itself is not installed (e.g. on the GPU test box).  When medaka is installed the
engine accepts the reference's own `Batch` -- only attribute access is used.
"""
from dataclasses import dataclass

import numpy as np
import torch


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

        2-d `features` (columns x 10) become the float32 `counts_matrix` (B, T, 10);
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
            d["counts_matrix"] = torch.stack([torch.from_numpy(f) for f in feats]).float()
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
