"""Synthetic TRAINING tasks for the weight zoo (oracle/make_golden_zoo.py) -- test infrastructure.

The one trained weight set of rounds 1-3 learned a per-column majority vote: it needs no memory at all, so it says
little about how far back a real consensus model looks.  These tasks force memory of controlled length into the
weights, so that the split scan's certificate (medaka_amd/csrc/scan_split.hpp) is exercised by models that DO remember:

  majority     per-column majority (medaka_amd.synth.counts_windows): the round-1 task, other seeds, 15x longer
  depthmix     the same labels on pileups whose coverage jumps between 5x and 500x and drops to zero for long runs
  homopolymer  run-length correction: the draft's homopolymer is one base too long or too short and every read's
               deletions are smeared over the run, so no single column shows a majority -- the model has to sum the
               deletion fractions over the whole run (memory = run length, up to 40 columns)
  latch        a rare marker column switches a systematic-error mode on or off; between markers (up to ~600 columns)
               only the remembered mode tells the label -- a model that learns this carries one bit for hundreds of
               columns, further than the default margin of the split scan
"""
import numpy as np

from medaka_amd import synth

TASKS = ("majority", "depthmix", "homopolymer", "latch")


def _cols_from_counts(fwd, rev, dfwd, drev, depth):
    """(n,4) forward / reverse base counts + deletion counts -> (n,10) features in channel order acgtACGTdD."""
    out = np.zeros((len(depth), 10), dtype=np.float32)
    out[:, 0:4] = rev
    out[:, 4:8] = fwd
    out[:, 8] = drev
    out[:, 9] = dfwd
    return out / np.maximum(depth, 1)[:, None].astype(np.float32)


def homopolymer_window(rng, n_cols, depth):
    x = np.zeros((n_cols, 10), np.float32)
    y = np.zeros(n_cols, np.int64)
    i = 0
    while i < n_cols:
        base = int(rng.integers(0, 4))
        Lt = int(min(40, 1 + rng.geometric(0.18)))                 # true run length
        e = int(rng.choice([-1, 0, 1], p=[0.2, 0.6, 0.2])) if Lt > 2 else 0
        Ld = Lt + e                                                  # the draft's run length = columns of this run
        n = min(Ld, n_cols - i)
        dep = np.maximum(rng.poisson(depth, n), 4)
        # a read reports Lt - k bases, k in {0, 1, 2} more likely in long runs; over Ld draft columns it shows
        # max(0, Ld - (Lt - k)) deletions, each landing on a random column of the run
        pk = np.array([1.0, 0.10 + 0.02 * Lt, 0.01 * Lt])
        pk /= pk.sum()
        dels_per_read = np.maximum(0, Ld - (Lt - np.arange(3)))
        p_del_col = float((pk * dels_per_read).sum()) / Ld          # per column, per read
        nf = rng.binomial(dep, 0.5)
        cols = np.zeros((n, 4)), np.zeros((n, 4))
        dl = []
        for s, nr in enumerate((nf, dep - nf)):
            nd = rng.binomial(nr, min(0.95, p_del_col + 0.01))
            ns = rng.binomial(nr - nd, 0.01)
            cols[s][:, base] = nr - nd - ns
            cols[s][np.arange(n), (base + rng.integers(1, 4, n)) % 4] += ns
            dl.append(nd)
        x[i:i + n] = _cols_from_counts(cols[0], cols[1], dl[0], dl[1], dep)
        lab = np.full(Ld, 1 + base)
        if Ld > Lt:
            lab[Lt:] = 0                                             # the surplus draft bases are gaps, at the run's end
        y[i:i + n] = lab[:n]
        i += n
        if Lt > Ld and i < n_cols:                                   # the missing base: an insertion column after the run
            dep1 = max(4, int(rng.poisson(depth)))
            k = rng.binomial(dep1, 0.75)
            kf = rng.binomial(k, 0.5)
            f, r = np.zeros((1, 4)), np.zeros((1, 4))
            f[0, base], r[0, base] = kf, k - kf
            x[i] = _cols_from_counts(f, r, np.zeros(1), np.zeros(1), np.array([dep1]))
            y[i] = 1 + base
            i += 1
    return x, y


_SWAP = np.array([0, 2, 1, 4, 3])      # mode 1: A<->C, G<->T on the labels


def latch_window(rng, n_cols, depth, seg=(50, 600)):
    x, y = synth.counts_windows(1, n_cols, depth=depth, seed=int(rng.integers(0, 2 ** 31)), p_draft_err=0.0, return_labels=True)
    x, y = x[0], y[0]
    mode, i = int(rng.integers(0, 2)), 0
    first = True
    while i < n_cols:
        run = int(rng.integers(seg[0], seg[1] + 1))
        if not first or rng.random() < 0.5:
            # marker column: forward strand all deleted (ON) / reverse strand all deleted (OFF); never seen otherwise
            x[i] = 0.0
            x[i, 9 if mode else 8] = 0.5
            x[i, (0 if mode else 4) + int(rng.integers(0, 4))] = 0.5
            y[i] = 0
        first = False
        if mode:
            y[i + 1:i + run] = _SWAP[y[i + 1:i + run]]
        i += run
        mode ^= 1
    return x, y


def depthmix_window(rng, n_cols):
    kind = ("depth_cliff", "zero_run", "homopolymer")[int(rng.integers(0, 3))]
    x, y = synth.structured_windows(kind, 1, n_cols, depth=int(rng.choice([15, 50, 120])), seed=int(rng.integers(0, 2 ** 31)),
                                    return_labels=True)
    return x[0], y[0]


def make_pool(task, n_windows, n_cols, seed, seg=(50, 600)):
    """(n_windows, n_cols, 10) float32 features, (n_windows, n_cols) int64 labels ('*ACGT' -> 0..4).
    `seg`: latch task only, shortest / longest distance between two markers."""
    rng = np.random.default_rng([seed, TASKS.index(task)])
    if task == "majority":
        return synth.counts_windows(n_windows, n_cols, depth=50, seed=int(rng.integers(0, 2 ** 31)), return_labels=True)
    xs, ys = [], []
    for _ in range(n_windows):
        if task == "homopolymer":
            x, y = homopolymer_window(rng, n_cols, int(rng.choice([25, 50, 90])))
        elif task == "latch":
            x, y = latch_window(rng, n_cols, 50, seg)
        else:
            x, y = depthmix_window(rng, n_cols)
        xs.append(x)
        ys.append(y)
    return np.stack(xs).astype(np.float32), np.stack(ys)
