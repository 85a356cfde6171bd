"""CPU restatement of the reference's CALLER of the hot path and of its stitcher (test infrastructure).

The product never imports this file: north_star keeps `run_prediction`, `Sample.chunks`,
`trim_samples` and `stitch` on the host, unchanged, in the reference.  The GPU box has no reference
tree, so the end-to-end parity tests (engine probabilities -> stitched FASTQ == the reference's own
FASTQ) need the same logic restated; every function cites what it follows, and
tests/test_oracle_stitch.py pins the whole chain to tests/golden/stitch_cases.npz, which
oracle/make_golden_stitch.py produced by running the UNMODIFIED reference.

Restated:
    Region.split                       medaka/common.py:711-736
    sliding_window / Sample.chunks     medaka/common.py:803-823, 429-453
    SampleGenerator.samples            medaka/features.py:1283-1313  (quarantine of narrow pileups)
    predict() region handling          medaka/prediction.py:92-110, 176-209
    DataLoader + run_prediction loop   medaka/prediction.py:14-81, 225-370  (threads and queues)
    DataIndex._get_sorted_index        medaka/datastore.py:452-484
    Sample.relative_position           medaka/common.py:232-322
    Sample.overlap_indices             medaka/common.py:324-427
    Sample.trim_samples[_to_region]    medaka/common.py:495-600
    _stitch_samples                    medaka/stitch.py:33-83
    collapse_neighbours, fastq naming  medaka/stitch.py:172-197, 15-30, 260-275
    HaploidLabelScheme.decode_consensus  -> oracle.decode_consensus (labels.py:1053-1085)
"""
import collections
import itertools
import queue
import threading

import numpy as np

from oracle import oracle as _oracle

POS_DTYPE = [("major", int), ("minor", int)]
Region = collections.namedtuple("Region", "ref_name start end")


class Pileup(collections.namedtuple("Pileup", "ref_name features positions label_probs depth")):
    """The fields of reference `medaka.common.Sample` (common.py:59-62) that this path touches."""

    @property
    def size(self):
        return len(self.positions)

    @property
    def first_pos(self):
        return int(self.positions["major"][0]), int(self.positions["minor"][0])

    @property
    def last_pos(self):
        return int(self.positions["major"][-1]), int(self.positions["minor"][-1])

    @property
    def name(self):          # common.py:125-131
        return "{}:{}.{}-{}.{}".format(self.ref_name, *self.first_pos, *self.last_pos)

    def cut(self, key):      # Sample.slice, common.py:455-478
        pick = lambda a: None if a is None else a[key]
        return Pileup(self.ref_name, pick(self.features), self.positions[key], pick(self.label_probs),
                      pick(self.depth))

    def with_probs(self, probs):
        return self._replace(label_probs=probs)


def make_positions(major, minor):
    pos = np.empty(len(major), dtype=POS_DTYPE)
    pos["major"], pos["minor"] = major, minor
    return pos


# ---------------------------------------------------------------------------------------------
# windows
def split_region(region, size, overlap):
    """Region.split(size, overlap, fixed_size=False): common.py:711-736."""
    if size >= region.end - region.start:
        return [region]
    return [Region(region.ref_name, s, min(s + size, region.end))
            for s in range(region.start, region.end, size - overlap)]


def window_starts(n, window, step):
    """sliding_window (common.py:803-823): stepped windows, then one right-aligned remainder window."""
    starts = list(range(0, n - window + 1, step))
    end = starts[-1] + window if starts else 0
    if n > end:
        starts.append(n - window)
    return starts


def chunk_pileup(p, chunk_len, overlap):
    """Sample.chunks: common.py:429-453."""
    return [p.cut(slice(s, s + chunk_len)) for s in window_starts(p.size, chunk_len, chunk_len - overlap)]


def region_samples(pileups, chunk_len, overlap, enable_chunking=True):
    """SampleGenerator.samples (features.py:1283-1313): returns (samples, quarantined regions)."""
    out, quarantined = [], []
    for p in pileups:
        if p.size == 0:
            continue
        if not enable_chunking:
            out.append(p)
        elif p.size < chunk_len:
            quarantined.append((Region(p.ref_name, p.first_pos[0], p.last_pos[0] + 1), p.size))
        else:
            out.extend(chunk_pileup(p, chunk_len, overlap))
    return out, quarantined


def plan_regions(contig_regions, chunk_len, chunk_ovlp, bam_chunk):
    """predict(), prediction.py:92-110: (regions for the batched pass, short regions for the remainder pass)."""
    regions, remainder = [], []
    for r in contig_regions:
        if r.end - r.start < chunk_len:
            remainder.append(r)
        elif r.end - r.start > bam_chunk:
            regions.extend(split_region(r, bam_chunk, chunk_ovlp))
        else:
            regions.append(r)
    return regions, remainder


# ---------------------------------------------------------------------------------------------
# the loop around predict_on_batch
class Loader:
    """DataLoader of prediction.py:225-370: `workers` threads turn regions into samples, one batcher
    thread groups them (common.grouper, :903-916 -- the last batch is short, never padded) and
    collates; the main thread iterates.  `pileups_of(region)` plays `bam_to_sample`."""

    def __init__(self, regions, pileups_of, collate, batch_size, chunk_len, chunk_ovlp, enable_chunking=True,
                 workers=2, batch_cache=8):
        self.batch_size, self.workers = batch_size, workers
        self.remainders = []
        self._args = (chunk_len, chunk_ovlp, enable_chunking)
        self._pileups_of, self._collate = pileups_of, collate
        self._samples = queue.Queue(maxsize=batch_cache * batch_size)
        self._batches = queue.Queue(maxsize=batch_cache)
        self._regions = queue.Queue()
        for r in regions:
            self._regions.put(r)
        self.batches_made = 0          # progress counter read by the GIL test
        self._threads = [threading.Thread(target=self._region_worker, daemon=True) for _ in range(workers)]
        self._threads.append(threading.Thread(target=self._batch_worker, daemon=True))
        for t in self._threads:
            t.start()

    def _region_worker(self):
        while True:
            try:
                region = self._regions.get_nowait()
            except queue.Empty:
                self._samples.put(StopIteration)
                return
            samples, remain = region_samples(self._pileups_of(region), *self._args)
            for s in samples:
                self._samples.put(s)
            self.remainders.extend(remain)

    def _sample_stream(self):
        stops = 0
        while stops < self.workers:
            item = self._samples.get()
            if item is StopIteration:
                stops += 1
            else:
                yield item

    def _batch_worker(self):
        stream = self._sample_stream()
        while True:
            data = list(itertools.islice(stream, self.batch_size))
            if not data:
                break
            self._batches.put((data, self._collate(data)))
            self.batches_made += 1
        self._batches.put(StopIteration)

    def __iter__(self):
        while True:
            item = self._batches.get()
            if item is StopIteration:
                return
            yield item


def run_prediction(store, regions, pileups_of, model, collate, chunk_len, chunk_ovlp, batch_size=200,
                   enable_chunking=True, workers=2, on_batch=None):
    """prediction.py:14-81: every sample leaves with its row of `model.predict_on_batch(batch)`.
    `store` maps sample name -> Pileup (DataStore.write_sample keeps the first of a name, datastore.py:263-300)."""
    loader = Loader(regions, pileups_of, collate, batch_size, chunk_len, chunk_ovlp, enable_chunking, workers)
    for data, batch in loader:
        probs = model.predict_on_batch(batch)
        if on_batch is not None:
            on_batch(loader, data, batch, probs)
        for sample, p in zip(data, probs):
            if sample.name not in store:
                store[sample.name] = sample.with_probs(np.array(p.numpy() if hasattr(p, "numpy") else p))
    return loader.remainders


def predict(contig_regions, pileups_of, model, collate, chunk_len, chunk_ovlp, batch_size, bam_chunk, store=None,
            on_batch=None):
    """predict(), prediction.py:84-222 without BAM/HDF/argument handling: the batched pass over long
    regions, then the quarantined and short regions one by one, unchunked, batch_size 1."""
    store = {} if store is None else store
    regions, remainder = plan_regions(contig_regions, chunk_len, chunk_ovlp, bam_chunk)
    if regions:
        rem = run_prediction(store, regions, pileups_of, model, collate, chunk_len, chunk_ovlp, batch_size,
                             on_batch=on_batch)
        remainder.extend(r[0] for r in rem)
    if remainder:
        left = run_prediction(store, remainder, pileups_of, model, collate, chunk_len, chunk_ovlp, 1,
                              enable_chunking=False, on_batch=on_batch)
        assert not left
    return store


# ---------------------------------------------------------------------------------------------
# stitch
def sorted_names(names):
    """DataIndex._get_sorted_index (datastore.py:452-484): per contig, by start and then longest first."""
    def key(name):
        _, span = name.rsplit(":", 1)
        a, b = span.split("-")
        st, en = tuple(int(i) for i in a.split(".")), tuple(int(i) for i in b.split("."))
        return st + tuple(-i for i in en)
    by_ref = collections.defaultdict(list)
    for n in names:
        by_ref[n.rsplit(":", 1)[0]].append(n)
    return {ref: sorted(v, key=key) for ref, v in sorted(by_ref.items())}


def relationship(s1, s2):
    """Sample.relative_position (common.py:232-322) for samples of one contig; returns one of
    'within' (s2 inside s1), 'within_rev', 'abutted', 'overlap', 'gapped' and their '_rev' forms."""
    a, b = sorted((s1, s2), key=lambda s: (s.first_pos, -s.size))
    fwd = a.name == s1.name
    tag = lambda t: t if fwd else t + "_rev"
    (a_maj, a_min), (b_maj, b_min) = a.last_pos, b.first_pos
    if b.first_pos >= a.first_pos and b.last_pos <= a.last_pos:
        return tag("within")
    if (b_maj == a_maj + 1 and b_min == 0) or (b_maj == a_maj and b_min == a_min + 1):
        return tag("abutted")
    if b_maj < a_maj or (b_maj == a_maj and b_min < a_min + 1):
        return tag("overlap")
    return tag("gapped")


class OverlapError(Exception):
    pass


JUNCTION_LOG = None     # tests set this to a list to record every (end_1, start_2, heuristic) chosen


def overlap_indices(s1, s2):
    """Sample.overlap_indices (common.py:324-427): (end of s1, start of s2, heuristic used)."""
    r = _overlap_indices(s1, s2)
    if JUNCTION_LOG is not None:
        JUNCTION_LOG.append((-1 if r[0] is None else r[0], -1 if r[1] is None else r[1], int(r[2])))
    return r


def _overlap_indices(s1, s2):
    rel = relationship(s1, s2)
    if rel == "abutted":
        return None, None, False
    if rel != "overlap":
        raise OverlapError(f"cannot overlap {s1.name} and {s2.name}: {rel}")
    i1 = int(np.searchsorted(s1.positions, s2.positions[0]))
    j2 = int(np.searchsorted(s2.positions, s1.positions[-1], side="right"))
    p1, p2 = s1.positions[i1:], s2.positions[:j2]
    if np.array_equal(p1["minor"], p2["minor"]):       # columns line up: cut at the middle of the overlap
        n = len(p1)
        return i1 + n // 2, j2 - (n - n // 2), False
    # columns differ: nearest major position around the middle that both samples hold equally often
    if len(np.unique(p1["major"])) > 3 and len(np.unique(p2["major"])) > 3:
        lo, hi = int(p1["major"][0]), int(p1["major"][-1])
        mid, off = lo + (hi - lo) // 2, 1
        top, bottom = int(s1.positions["major"].max()), int(s2.positions["major"].min())
        while not (mid + off > top and mid - off < bottom):
            for test in (off, -off):
                left = np.where(s1.positions["major"] == mid + test)[0]
                right = np.where(s2.positions["major"] == mid + test)[0]
                if len(left) == len(right):
                    # (the reference indexes [0] of both and fails with IndexError when both are empty;
                    # synthetic pileups hold every major position, so that branch is not reachable here)
                    return int(left[0]), int(right[0]), True
            off += 1
    raise OverlapError(f"no junction for {s1.name} and {s2.name}")


def trim_samples(samples):
    """Sample.trim_samples (common.py:495-557): yields (trimmed view, is_last_in_contig, heuristic)."""
    it = iter(samples)
    try:
        s1 = next(it)
    except StopIteration:
        return
    start_1 = start_2 = None
    for s2 in itertools.chain(it, (None,)):
        heuristic, last = False, False
        if s2 is None:
            end_1, last = None, True
        else:
            rel = relationship(s1, s2)
            if rel == "within":
                continue
            if rel == "overlap":
                end_1, start_2, _ = overlap_indices(s1, s2)      # (the flag is dropped on this branch: :534-536)
            elif rel == "gapped":
                last, end_1, start_2 = True, None, None
            else:
                end_1, start_2, heuristic = overlap_indices(s1, s2)
        yield s1.cut(slice(start_1, end_1)), last, heuristic
        s1, start_1 = s2, start_2


def trim_to_region(stream, start, end):
    """Sample.trim_samples_to_region (common.py:559-600)."""
    for s, last, heur in stream:
        maj = s.positions["major"]
        if maj[-1] < start:
            continue
        if maj[0] < start:
            q = np.array([(start, 0)], dtype=s.positions.dtype)
            s = s.cut(slice(int(np.searchsorted(s.positions, q[0])), None))
            maj = s.positions["major"]
        if len(maj) == 0:
            continue
        if maj[0] >= end:
            return
        if maj[-1] >= end:
            s = s.cut(slice(None, int(np.searchsorted(maj, end))))
        if s.size > 0:
            yield s, last, heur


def stitch_region(samples, region, decode=_oracle.decode_consensus):
    """_stitch_samples (stitch.py:33-83) with min_depth 0: list of ((ref, start, stop), seq parts, qual parts)."""
    contigs, seqs, quals, start, s = [], [], [], None, None
    for s, last, _ in trim_to_region(trim_samples(samples), region.start, region.end):
        start = int(s.positions["major"][0]) if start is None else start
        seq, qual = decode(s.label_probs, with_qualities=True)
        seqs.append(seq)
        quals.append(qual)
        if last:
            contigs.append(((s.ref_name, start, int(s.positions["major"][-1])), seqs, quals))
            seqs, quals, start = [], [], None
    if seqs:
        contigs.append(((s.ref_name, start, int(s.positions["major"][-1])), seqs, quals))
    return contigs


def collapse(pieces):
    """collapse_neighbours (stitch.py:172-197)."""
    out = []
    for (ref, start, stop), seqs, quals in pieces:
        if out and out[-1][0][0] == ref and start == out[-1][0][2] + 1:
            (r0, s0, _), sq, ql = out[-1]
            out[-1] = ((r0, s0, stop), sq + seqs, ql + quals)
        else:
            out.append(((ref, start, stop), list(seqs), list(quals)))
    return out


def fastq(store, contig_lengths, decode=_oracle.decode_consensus):
    """`medaka sequence` without --fillgaps (stitch.py:199-275, serial path): samples of every file
    (`store`: name -> Pileup with label_probs) in index order, contig by contig in 1 Mb regions."""
    index = sorted_names(store.keys())
    pieces = []
    for ref in index:
        whole = Region(ref, 0, contig_lengths[ref])
        for region in split_region(whole, int(1e6), 0):
            names = index[ref]          # yield_from_feature_files keeps the samples overlapping the region
            keep = [store[n] for n in names
                    if store[n].first_pos[0] < region.end and store[n].last_pos[0] + 1 > region.start]
            pieces.extend(stitch_region(keep, region, decode))
    text, prev, counter = [], None, 0
    for (ref, start, stop), seqs, quals in collapse(pieces):
        counter = counter + 1 if ref == prev else 0
        text.append("@{}_{} {}-{}\n{}\n+\n{}\n".format(ref, counter, start, stop + 1, "".join(seqs), "".join(quals)))
        prev = ref
    return "".join(text)


# ---------------------------------------------------------------------------------------------
# golden inputs
def load_case(gold, case):
    """tests/golden/stitch_cases.npz -> (spec dict, contig -> [Pileup]) with features rebuilt by
    oracle.normalise_counts (bit-identical to the reference's _post_process_pileup: tests/test_oracle.py)."""
    from oracle.make_golden_stitch import CASES
    spec = CASES[case]
    sources = {}
    for ctg, pieces in spec["contigs"].items():
        sources[ctg] = []
        for k in range(len(pieces)):
            g = lambda f: gold[f"{case}/raw/{ctg}/{k}/{f}"]
            feats = _oracle.normalise_counts(g("counts"), g("depth"))
            sources[ctg].append(Pileup(ctg, feats, make_positions(g("major"), g("minor")), None, g("depth")))
    return spec, sources


def pileups_in_region(sources, region, jitter=False):
    """The synthetic `bam_to_sample` of oracle/make_golden_stitch.py::region_pileups."""
    out = []
    for s in sources[region.ref_name]:
        maj = s.positions["major"]
        lo, hi = int(np.searchsorted(maj, region.start, "left")), int(np.searchsorted(maj, region.end, "left"))
        if hi <= lo:
            continue
        piece = s.cut(slice(lo, hi))
        if jitter and region.start > 0:
            pos = piece.positions
            minors = np.nonzero((pos["minor"] > 0) & (pos["major"] < region.start + 150))[0]
            keep = np.ones(len(pos), dtype=bool)
            keep[minors[::3]] = False
            piece = piece.cut(keep)
        out.append(piece)
    return out


def contig_regions(sources):
    return [Region(c, 0, int(max(p.positions["major"][-1] for p in ps)) + 1) for c, ps in sources.items()]
