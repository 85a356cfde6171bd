"""Region sharding for multi-GPU consensus.

Pileup windows never exchange state (each window starts from h_0 = 0, reference
medaka/common.py:429-453, gru.py:66), so N GPUs are N independent `medaka inference`
processes on disjoint `--regions`, joined by `medaka sequence out_0.hdf ... out_{N-1}.hdf`
(reference README.md:294-330, stitch.py:202).  No collective is on the data path.

This module only decides WHO takes WHICH region, and it cuts exactly where the reference cuts:
`medaka inference` splits every region longer than `bam_chunk` into pieces of `bam_chunk` bases that
start every `bam_chunk - chunk_ovlp` bases (prediction.py:100-110 -> common.Region.split with
fixed_size=False, common.py:711-736).  `shard_regions` hands out THOSE pieces -- each is at most
`bam_chunk` long, so the per-GPU process does not cut it again -- which makes the set of regions, hence
of pileups, windows and samples, identical to a single-process run.  The PROBABILITIES of a sample are
the single-process run's to ~1e-7, not bit for bit: with the split scan on (the default) they depend at that
level on how many windows share a batch and on the margin a model has escalated to, and a sharded run
batches its windows differently; the stitched consensus is the same wherever the model separates its top
two classes by more than that (tests/test_e2e_gpu.py checks a sharded run against the reference's own
FASTQ).  Where bit-reproducibility across shardings matters, run the children with MDK_SCAN_SPLIT=0
(`medaka_amd.launch --reproducible`): the sequential scan's bits do not depend on the batch.  One exception is handled explicitly: a trailing piece SHORTER than `chunk_len` would, as a
region of its own, take the child's `region.size < chunk_len` branch (prediction.py:97-98: un-chunked
remainder pass) whereas a single process keeps it in the batched pass (it is only the *contig* that is tested
there).  Such a tail therefore travels together with its predecessor as ONE region [prev.start, end): the
child re-cuts it on the same grid into the same two pieces and the tail stays in the batched pass.
Pieces are assigned longest-first to the least loaded shard (LPT), ties by input order.
"""
from collections import namedtuple

Region = namedtuple("Region", "ref_name start end")


def region_str(r):
    """samtools-style string accepted by `medaka inference --regions`."""
    return f"{r.ref_name}:{r.start}-{r.end}"


def split_region(region, chunk, overlap):
    """The pieces reference `Region.split(chunk, overlap, fixed_size=False)` yields, trailing short
    piece included (it is narrower than chunk_len and ends in the reference's remainder pass)."""
    if chunk <= overlap:
        raise ValueError("chunk must exceed overlap")
    if chunk >= region.end - region.start:
        return [region]
    return [Region(region.ref_name, s, min(s + chunk, region.end))
            for s in range(region.start, region.end, chunk - overlap)]


def shardable_pieces(region, bam_chunk, chunk_ovlp, chunk_len):
    """`split_region`, with a trailing piece shorter than `chunk_len` joined to its predecessor (module
    docstring): every returned region is cut by the child exactly as a single process cuts the contig."""
    pieces = split_region(region, bam_chunk, chunk_ovlp)
    if len(pieces) > 1 and pieces[-1].end - pieces[-1].start < chunk_len:
        prev, tail = pieces[-2], pieces[-1]
        if tail.end - prev.start > bam_chunk:          # the child will re-cut [prev.start, end) into prev + tail
            pieces[-2:] = [Region(region.ref_name, prev.start, tail.end)]
        # else: the tail lies inside its predecessor (fewer than chunk_ovlp bases were left); it is redundant for
        # the consensus and is kept as the reference's grid has it
    return pieces


def shard_regions(contigs, n_shards, bam_chunk=1_000_000, chunk_ovlp=1000, chunk_len=10000):
    """contigs: iterable of (name, length) or Region.  Returns list[n_shards] of list[Region], every
    region one piece of the reference's own bam_chunk grid (or a last piece + its short tail).  Deterministic."""
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    pieces = []
    for c in contigs:
        r = c if isinstance(c, Region) else Region(c[0], 0, int(c[1]))
        if r.end > r.start:      # (a single process cuts its regions itself)
            pieces.extend(shardable_pieces(r, bam_chunk, chunk_ovlp, chunk_len) if n_shards > 1 else [r])
    order = sorted(range(len(pieces)), key=lambda i: (-(pieces[i].end - pieces[i].start), i))
    shards = [[] for _ in range(n_shards)]
    load = [0] * n_shards
    for i in order:
        k = min(range(n_shards), key=lambda s: (load[s], s))
        shards[k].append((i, pieces[i]))
        load[k] += pieces[i].end - pieces[i].start
    return [[r for _, r in sorted(s)] for s in shards]     # input order inside a shard


def shard_windows(n_windows, n_shards, rank):
    """Contiguous split of a window list across ranks (synthetic bench / in-memory runs):
    rank r takes windows [lo, hi)."""
    base, rem = divmod(n_windows, n_shards)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi
