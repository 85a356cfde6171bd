"""Region sharding for multi-GPU consensus.

Pileup windows never exchange state (each window starts from h_0 = 0, reference
medaka/common.py:429-453, gru.py:66), so N GPUs are N independent `medaka inference`
processes on disjoint `--regions`, joined by `medaka sequence out_0.hdf ... out_{N-1}.hdf`
(reference README.md:294-330, stitch.py:202).  No collective is on the data path.

This module only decides WHO takes WHICH region:
  * whole contigs are assigned longest-first to the least loaded shard (LPT);
  * contigs longer than a quarter of `total/n_shards` are first cut on multiples of `bam_chunk` with
    `chunk_ovlp` bases of overlap, the same cut `prediction.predict` itself applies
    (prediction.py:100-110 -> common.Region.split), so that stitch sees overlapping samples.
"""
from collections import namedtuple

Region = namedtuple("Region", "ref_name start end")


def region_str(r):
    """samtools-style string accepted by `medaka inference --regions`."""
    return f"{r.ref_name}:{r.start}-{r.end}"


def split_region(region, chunk, overlap):
    """Cut [start, end) into pieces of <= chunk bases overlapping by `overlap`
    (semantics of reference common.Region.split, fixed-size chunks, last piece short)."""
    if chunk <= overlap:
        raise ValueError("chunk must exceed overlap")
    out = []
    pos = region.start
    while True:
        end = min(pos + chunk, region.end)
        out.append(Region(region.ref_name, pos, end))
        if end >= region.end:
            break
        pos = end - overlap
    return out


def shard_regions(contigs, n_shards, bam_chunk=1_000_000, chunk_ovlp=1000):
    """contigs: iterable of (name, length) or Region.  Returns list[n_shards] of list[Region].

    Deterministic: ties broken by input order.
    """
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    regions = []
    for c in contigs:
        r = c if isinstance(c, Region) else Region(c[0], 0, int(c[1]))
        if r.end > r.start:
            regions.append(r)
    total = sum(r.end - r.start for r in regions)
    target = max(1, -(-total // n_shards))
    # pieces of about a quarter of a shard keep the greedy packing within a few percent
    per = max(bam_chunk, -(-(-(-target // 4)) // bam_chunk) * bam_chunk)
    pieces = []
    for r in regions:
        if n_shards > 1 and (r.end - r.start) > per:
            pieces.extend(split_region(r, per, chunk_ovlp))
        else:
            pieces.append(r)
    order = sorted(range(len(pieces)), key=lambda i: (-(pieces[i].end - pieces[i].start), i))
    shards = [[] for _ in range(n_shards)]
    load = [0] * n_shards
    for i in order:
        k = min(range(n_shards), key=lambda s: (load[s], s))
        shards[k].append(pieces[i])
        load[k] += pieces[i].end - pieces[i].start
    for s in shards:
        s.sort(key=lambda r: (r.ref_name, r.start))
    return shards


def shard_windows(n_windows, n_shards, rank):
    """Contiguous split of a window list across ranks (synthetic bench / in-memory runs):
    rank r takes windows [lo, hi)."""
    base, rem = divmod(n_windows, n_shards)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi
