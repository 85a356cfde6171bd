"""Synthetic `medaka_counts` pileup windows.

The reference builds its network input in `CountsFeatureEncoder._post_process_pileup`
(reference medaka/features.py:871-935) from `calculate_pileup`
(reference src/medaka_counts.c:199-372): one row per pileup column (major
reference position or minor/insertion position), ten float32 channels in the
order ``a c g t A C G T d D`` (lower case = reverse strand,
reference src/medaka_counts.h:19-30), each row divided by the read depth at the
parent major position.  This module draws windows with that contract so that
parity tests and the bench see realistic, non-degenerate data without a BAM.

Not a port of any reference code: the reference has no synthetic generator for
normalised counts (its tests draw uniform noise, medaka/test/test_sample.py:50).
"""
import numpy as np

# class labels follow the reference HaploidLabelScheme alphabet '*ACGT'
# (reference medaka/labels.py:342): 0='*' (gap), 1..4 = A,C,G,T
NUM_FEATURES = 10
NUM_CLASSES = 5


def counts_windows(n_windows, n_cols, depth=60, seed=1234, p_sub=0.01,
                   p_del=0.01, p_ins=0.01, p_draft_err=0.02,
                   return_labels=False, raw=False):
    """Draw `n_windows` windows of `n_cols` pileup columns.

    Returns float32 ``(n_windows, n_cols, 10)`` (and int64 labels
    ``(n_windows, n_cols)`` when `return_labels`).  With `raw` the un-normalised
    pileup is returned instead, as the reference's `pileup_counts` hands it to
    `_post_process_pileup`: dict(counts uint16 (W,T,10), depth uint32 (W,T) = depth of
    the parent major column, major / minor int64 (W,T) position fields).
    """
    rng = np.random.default_rng(seed)
    out = np.zeros((n_windows, n_cols, NUM_FEATURES), dtype=np.float32)
    labels = np.zeros((n_windows, n_cols), dtype=np.int64)
    raw_depth = np.ones((n_windows, n_cols), dtype=np.uint32)
    raw_major = np.zeros((n_windows, n_cols), dtype=np.int64)
    raw_minor = np.zeros((n_windows, n_cols), dtype=np.int64)
    for w in range(n_windows):
        # --- column skeleton: every reference position gives a major column,
        # followed by a minor column when at least one read inserts there.
        n_pos = n_cols  # upper bound; trimmed below
        depth_pos = np.maximum(rng.poisson(depth, n_pos), 1)
        true_base = rng.integers(0, 4, n_pos)
        kind = rng.random(n_pos)
        # draft errors: real insertion missing from the draft / extra draft base
        real_ins = kind < p_draft_err / 2
        extra_base = (kind >= p_draft_err / 2) & (kind < p_draft_err)
        p_ins_pos = np.where(real_ins, 0.95, p_ins)
        n_ins = rng.binomial(depth_pos, p_ins_pos)
        has_minor = n_ins > 0
        n_per_pos = 1 + has_minor.astype(np.int64)
        start = np.cumsum(n_per_pos) - n_per_pos
        keep = start < n_cols
        # --- major columns
        n_fwd = rng.binomial(depth_pos, 0.5)
        p_del_pos = np.where(extra_base, 0.95, p_del)
        for strand, n_reads in ((1, n_fwd), (0, depth_pos - n_fwd)):
            n_d = rng.binomial(n_reads, p_del_pos)
            n_s = rng.binomial(n_reads - n_d, p_sub)
            n_ok = n_reads - n_d - n_s
            sub_base = (true_base + rng.integers(1, 4, n_pos)) % 4
            cols = start[keep]
            base_off = 4 * strand
            np.add.at(out[w], (cols, base_off + true_base[keep]), n_ok[keep])
            np.add.at(out[w], (cols, base_off + sub_base[keep]), n_s[keep])
            np.add.at(out[w], (cols, 8 + strand), n_d[keep])
        labels[w, start[keep]] = np.where(extra_base[keep], 0, 1 + true_base[keep])
        # --- minor columns: only inserting reads counted
        mk = keep & has_minor & (start + 1 < n_cols)
        ins_base = rng.integers(0, 4, n_pos)
        ins_fwd = rng.binomial(n_ins, 0.5)
        mcols = start[mk] + 1
        np.add.at(out[w], (mcols, 4 + ins_base[mk]), ins_fwd[mk])
        np.add.at(out[w], (mcols, ins_base[mk]), (n_ins - ins_fwd)[mk])
        labels[w, mcols] = np.where(real_ins[mk], 1 + ins_base[mk], 0)
        # --- normalise by depth of the parent major column
        col_depth = np.ones(n_cols, dtype=np.float32)
        col_depth[start[keep]] = depth_pos[keep]
        col_depth[mcols] = depth_pos[mk]
        raw_depth[w] = col_depth.astype(np.uint32)
        raw_major[w, start[keep]] = np.nonzero(keep)[0]
        raw_major[w, mcols] = np.nonzero(mk)[0]
        raw_minor[w, mcols] = 1
        if not raw:
            out[w] /= col_depth[:, None]
    if raw:
        return dict(counts=out.astype(np.uint16), depth=raw_depth, major=raw_major, minor=raw_minor)
    if return_labels:
        return out, labels
    return out


def uniform_windows(n_windows, n_cols, seed=1234):
    """Dense uniform [0,1) windows (timing is data independent for this path)."""
    rng = np.random.default_rng(seed)
    return rng.random((n_windows, n_cols, NUM_FEATURES), dtype=np.float32)
