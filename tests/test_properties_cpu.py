"""Property tests (hypothesis) of the host-side logic around the hot path: region sharding, region strings,
batch assembly.  CPU only."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from medaka_amd import launch, sharding, torch_ext
from oracle import stitch_oracle as so

contig_lists = st.lists(st.integers(min_value=1, max_value=3_000_000), min_size=1, max_size=12)


@settings(max_examples=60, deadline=None)
@given(lengths=contig_lists, n_shards=st.integers(1, 9), bam_chunk=st.sampled_from([50_000, 250_000, 1_000_000]),
       ovlp=st.sampled_from([0, 100, 1000]), chunk_len=st.sampled_from([1000, 10_000]))
def test_sharding_covers_exactly_what_one_process_cuts(lengths, n_shards, bam_chunk, ovlp, chunk_len):
    """Whatever the contigs: every shard region, cut again by the child as `medaka inference` cuts its regions
    (prediction.py:100-110), gives back exactly the pieces ONE process would have cut -- nothing lost, nothing twice
    -- and no shard region is a sub-`chunk_len` tail of a longer contig."""
    contigs = [(f"c{i}", n) for i, n in enumerate(lengths)]
    shards = sharding.shard_regions(contigs, n_shards, bam_chunk=bam_chunk, chunk_ovlp=ovlp, chunk_len=chunk_len)
    assert len(shards) == n_shards
    got = []
    for regs in shards:
        for r in regs:
            pieces = [r] if r.end - r.start <= bam_chunk else so.split_region(so.Region(*r), bam_chunk, ovlp)
            got.extend(tuple(p) for p in pieces)
    want = []
    for name, n in contigs:
        whole = so.Region(name, 0, n)
        want.extend(tuple(p) for p in ([whole] if n <= bam_chunk or n_shards == 1 else so.split_region(whole, bam_chunk, ovlp)))
    if n_shards == 1:
        assert sorted(got) == sorted(tuple(p) for name, n in contigs
                                     for p in ([so.Region(name, 0, n)] if n <= bam_chunk else so.split_region(so.Region(name, 0, n), bam_chunk, ovlp)))
    else:
        assert sorted(got) == sorted(want)
        for regs in shards:
            for r in regs:
                n = dict(contigs)[r.ref_name]
                tail_of_longer = r.end == n and r.start > 0 and r.end - r.start < chunk_len
                covered = r.start > 0 and any(o.ref_name == r.ref_name and o.start < r.start and o.end >= r.end
                                              for s in shards for o in s if o is not r)
                assert not tail_of_longer or covered     # only the reference's redundant in-predecessor tail may stay
    loads = [sum(r.end - r.start for r in regs) for regs in shards]
    if n_shards > 1 and sum(loads):
        biggest = max((r.end - r.start for regs in shards for r in regs), default=0)
        assert max(loads) - min(loads) <= biggest          # LPT: never worse than one piece apart


names = st.text(alphabet="abcXYZ019_.:|-", min_size=1, max_size=8).filter(lambda s: not s.startswith("-"))


@settings(max_examples=80, deadline=None)
@given(name=names, length=st.integers(1, 10**7), a=st.integers(0, 10**7), b=st.integers(0, 10**7))
def test_region_strings_round_trip(name, length, a, b):
    lengths = {name: length}
    lo, hi = min(a, b), max(a, b)
    assert launch.parse_region(name, lengths) == sharding.Region(name, 0, length)
    if f"{name}:{lo}-{hi}" not in lengths:
        assert launch.parse_region(f"{name}:{lo}-{hi}", lengths) == sharding.Region(name, lo, min(hi, length))
        assert launch.parse_region(f"{name}:{lo}-", lengths) == sharding.Region(name, lo, length)
        assert launch.parse_region(f"{name}:-{hi}", lengths) == sharding.Region(name, 0, min(hi, length))


@settings(max_examples=40, deadline=None)
@given(n=st.integers(1, 20), T=st.integers(1, 300), step=st.integers(1, 300), threads=st.integers(1, 9), seed=st.integers(0, 10))
def test_stack_counts_equals_torch_stack(n, T, step, threads, seed):
    rng = np.random.default_rng(seed)
    big = rng.random(((n - 1) * step + T, 10), dtype=np.float32)
    feats = [big[i * step:i * step + T] for i in range(n)]
    assert torch.equal(torch_ext.stack_counts(feats, threads=threads), torch.stack([torch.from_numpy(f) for f in feats]).float())
