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


# ---- read-level model (reference LatentSpaceLSTM, BASELINE config 4b) -------------------------------
def synth_reads(B, P, D, use_dwells=False, seed=0, empty_tail=True):
    """Random read-level features in the reference's layout [base, qual, strand, mapq(, dwell)]
    (src/medaka_read_matrix.c; strand kept in {0, 1} -- see DESIGN.md on the reference's uint8 cast)."""
    rng = np.random.default_rng(seed)
    Fd = 5 if use_dwells else 4
    x = np.zeros((B, P, D, Fd), dtype=np.uint8)
    x[..., 0] = rng.integers(0, 6, (B, P, D))
    x[..., 1] = rng.integers(0, 50, (B, P, D))
    x[..., 2] = rng.integers(0, 2, (B, P, D))
    x[..., 3] = rng.integers(0, 61, (B, P, D))
    if use_dwells:
        x[..., 4] = rng.integers(0, 30, (B, P, D))
    if empty_tail:   # padded (empty) reads as Batch.collate produces for shallower windows
        for b in range(B):
            n_empty = int(rng.integers(0, max(1, D // 2)))
            if n_empty:
                x[b, :, D - n_empty:, :] = 0
    return x


def synth_rl_state(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False, seed=0, lstm_gain=3.0,
                   head_gain=40.0):
    """A full `LatentSpaceLSTM.state_dict()` (reference key names and shapes, latent_space_lstm.py:
    93-150) drawn from a seeded numpy generator: PyTorch-default-like uniform(+-1/sqrt(fan_in)) ranges,
    non-trivial batch-norm statistics, LSTM weights x `lstm_gain` and head x `head_gain` so that the
    gates saturate and the output distribution is peaked (max prob ~0.6 on average) without making the recurrence chaotic.  Used where committing the
    weights themselves would be too large (rl_lstm384 is 20 MB): goldens store only x and y."""
    rng = np.random.default_rng(seed)
    nf = 6 + 1 + (1 if use_dwells else 0)
    H, C = lstm_size, cnn_size

    def uni(shape, fan):
        b = 1.0 / np.sqrt(fan)
        return rng.uniform(-b, b, shape).astype(np.float32)

    st = {"base_embedder.weight": rng.standard_normal((6, 6)).astype(np.float32),
          "strand_embedder.weight": rng.standard_normal((3, 6)).astype(np.float32)}
    for conv, bn, cin, k in ((0, 2, nf, 1), (3, 5, C, 17)):
        st[f"read_level_conv.convs.{conv}.weight"] = uni((C, cin, k), cin * k)
        st[f"read_level_conv.convs.{conv}.bias"] = uni((C,), cin * k)
        st[f"read_level_conv.convs.{bn}.weight"] = (rng.random(C) + 0.5).astype(np.float32)
        st[f"read_level_conv.convs.{bn}.bias"] = (rng.standard_normal(C) * 0.1).astype(np.float32)
        st[f"read_level_conv.convs.{bn}.running_mean"] = (rng.standard_normal(C) * 0.3).astype(np.float32)
        st[f"read_level_conv.convs.{bn}.running_var"] = (rng.random(C) + 0.5).astype(np.float32)
    st["read_level_conv.expansion_layer.weight"] = uni((H, C), C)
    st["read_level_conv.expansion_layer.bias"] = uni((H,), C)
    st["pre_pool_expansion_layer.weight"] = uni((H, C), C)
    st["pre_pool_expansion_layer.bias"] = uni((H,), C)
    if bidirectional:
        for layer in range(2):
            for sfx in ("", "_reverse"):
                kin = H if layer == 0 else 2 * H
                st[f"lstm.weight_ih_l{layer}{sfx}"] = uni((4 * H, kin), H) * np.float32(lstm_gain)
                st[f"lstm.weight_hh_l{layer}{sfx}"] = uni((4 * H, H), H) * np.float32(lstm_gain)
                st[f"lstm.bias_ih_l{layer}{sfx}"] = uni((4 * H,), H)
                st[f"lstm.bias_hh_l{layer}{sfx}"] = uni((4 * H,), H)
    else:
        for i in range(4):
            st[f"lstm.{i}.lstm.weight_ih_l0"] = uni((4 * H, H), H) * np.float32(lstm_gain)
            st[f"lstm.{i}.lstm.weight_hh_l0"] = uni((4 * H, H), H) * np.float32(lstm_gain)
            st[f"lstm.{i}.lstm.bias_ih_l0"] = uni((4 * H,), H)
            st[f"lstm.{i}.lstm.bias_hh_l0"] = uni((4 * H,), H)
    st["linear.weight"] = uni((5, (2 if bidirectional else 1) * H), H) * np.float32(head_gain)
    st["linear.bias"] = uni((5,), H)
    return st


# ---- structured pileups: what real drafts contain and i.i.d. sequence does not --------------------------------
STRUCTURED_KINDS = ("zero_run", "homopolymer", "dinucleotide", "tandem", "depth_cliff", "all_minor")


def _emit_columns(rng, true_base, depth_pos, n_cols, p_sub=0.01, p_del=0.01, p_ins=0.01):
    """One window from a per-position truth: major column per position (+ a minor column where a read inserts),
    counts / depth of the parent major column, channel order ``a c g t A C G T d D`` as `counts_windows`.
    depth 0 gives all-zero columns (counts / max(1, depth), features.py:907-911)."""
    n_pos = len(true_base)
    out = np.zeros((n_cols, NUM_FEATURES), dtype=np.float32)
    labels = np.zeros(n_cols, dtype=np.int64)
    n_ins = rng.binomial(depth_pos, p_ins)
    n_per_pos = 1 + (n_ins > 0).astype(np.int64)
    start = np.cumsum(n_per_pos) - n_per_pos
    keep = start < n_cols
    cols = start[keep]
    n_fwd = rng.binomial(depth_pos, 0.5)
    for strand, n_reads in ((1, n_fwd), (0, depth_pos - n_fwd)):
        n_d = rng.binomial(n_reads, p_del)
        n_s = rng.binomial(n_reads - n_d, p_sub)
        sub_base = (true_base + rng.integers(1, 4, n_pos)) % 4
        np.add.at(out, (cols, 4 * strand + true_base[keep]), (n_reads - n_d - n_s)[keep])
        np.add.at(out, (cols, 4 * strand + sub_base[keep]), n_s[keep])
        np.add.at(out, (cols, 8 + strand), n_d[keep])
    labels[cols] = np.where(depth_pos[keep] > 0, 1 + true_base[keep], 0)
    mk = keep & (n_ins > 0) & (start + 1 < n_cols)
    ins_base = rng.integers(0, 4, n_pos)
    ins_fwd = rng.binomial(n_ins, 0.5)
    mcols = start[mk] + 1
    np.add.at(out, (mcols, 4 + ins_base[mk]), ins_fwd[mk])
    np.add.at(out, (mcols, ins_base[mk]), (n_ins - ins_fwd)[mk])
    col_depth = np.ones(n_cols, dtype=np.float32)
    col_depth[cols] = np.maximum(depth_pos[keep], 1)
    col_depth[mcols] = np.maximum(depth_pos[mk], 1)
    out /= col_depth[:, None]
    return out, labels


def structured_windows(kind, n_windows, n_cols, depth=50, seed=1234, return_labels=False):
    """Windows with the long-range structure of real pileups (the i.i.d. `counts_windows` has none):

      zero_run     a run of >= 2000 zero-coverage columns (all-zero features: the network runs on its biases alone)
      homopolymer  runs of one base, 5-60 long
      dinucleotide long two-base repeats (hundreds of columns), short random spacers
      tandem       a 3-12 base unit repeated for hundreds of columns
      depth_cliff  coverage jumping between 5x and 500x every few hundred columns
      all_minor    every column an insertion column: only the inserting reads counted, divided by the full depth
    """
    if kind not in STRUCTURED_KINDS:
        raise ValueError(f"unknown kind {kind!r} (one of {STRUCTURED_KINDS})")
    rng = np.random.default_rng([seed, STRUCTURED_KINDS.index(kind)])
    out = np.zeros((n_windows, n_cols, NUM_FEATURES), dtype=np.float32)
    labels = np.zeros((n_windows, n_cols), dtype=np.int64)
    for w in range(n_windows):
        n_pos = n_cols
        depth_pos = np.maximum(rng.poisson(depth, n_pos), 1)
        base = rng.integers(0, 4, n_pos)
        if kind == "zero_run":
            run = min(n_pos, int(rng.integers(2000, 3001))) if n_pos > 2000 else max(1, n_pos // 2)
            at = int(rng.integers(0, n_pos - run + 1))
            depth_pos[at:at + run] = 0
        elif kind == "homopolymer":
            i = 0
            while i < n_pos:
                run = int(rng.integers(5, 61))
                base[i:i + run] = rng.integers(0, 4)
                i += run
        elif kind in ("dinucleotide", "tandem"):
            i = 0
            while i < n_pos:
                unit = rng.integers(0, 4, 2 if kind == "dinucleotide" else int(rng.integers(3, 13)))
                if kind == "dinucleotide" and unit[0] == unit[1]:
                    unit[1] = (unit[1] + 1) % 4
                run = int(rng.integers(200, 901))
                base[i:i + run] = np.resize(unit, run)[:max(0, min(run, n_pos - i))]
                i += run + int(rng.integers(0, 20))
        elif kind == "depth_cliff":
            i, deep = 0, bool(rng.integers(0, 2))
            while i < n_pos:
                run = int(rng.integers(300, 1101))
                depth_pos[i:i + run] = np.maximum(rng.poisson(500 if deep else 5, len(depth_pos[i:i + run])), 1)
                i, deep = i + run, not deep
        if kind == "all_minor":
            # insertion columns only: a few of `depth` reads carry a base there
            n_ins = rng.binomial(depth_pos, 0.08)
            ins_fwd = rng.binomial(n_ins, 0.5)
            cols = np.arange(n_cols)
            np.add.at(out[w], (cols, 4 + base), ins_fwd)
            np.add.at(out[w], (cols, base), n_ins - ins_fwd)
            out[w] /= depth_pos[:, None].astype(np.float32)
            continue
        out[w], labels[w] = _emit_columns(rng, base, depth_pos, n_cols)
    if return_labels:
        return out, labels
    return out
