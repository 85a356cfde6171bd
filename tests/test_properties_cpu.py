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


def test_bench_kernel_table_arithmetic():
    """bench.py's roofline bookkeeping (pure arithmetic, no device): MFMA counts per kernel family of the split forward at
    200 x 10000, algorithmic FLOP on the real columns, fractions of the 2.5 PFLOP/s fp16 peak."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    split = {"chunks": 5, "columns": 2256}
    k, step = bench.kernel_table(([2.0], [4.0], [0.01], [0.04], [6.2]), 2 | 256, split, 200, 10000, False)
    by = {e["kernel"].split(" ")[0]: e for e in k}
    wg_steps = 1000 * 2256 / 8 * 2                       # (8-window work-group, step) pairs of one layer, both directions
    assert step["windows_per_work_group"] == 8 and step["virtual_columns_scanned"] == 2256000
    assert abs(by["k_rec_mfma<XIN>"]["issued_gflop"] - wg_steps * 8 * 30 * 16384 / 1e9) < 1e-6
    assert abs(by["k_rec_fused"]["issued_gflop"] - wg_steps * 8 * (24 + 36 + 1) * 16384 / 1e9) < 1e-6
    assert abs(by["k_rec_fused"]["algorithmic_gflop"] - 2 * (98304 + 196608 + 1280) * 2e6 / 1e9) < 1e-6
    assert abs(step["algorithmic_gflop"] - 804352 * 2e6 / 1e9) < 1e-6
    assert abs(step["frac_issued_of_fp16_peak"] - step["issued_gflop"] / 6.2 / 2500.0) < 1e-12
    assert 0.4 < step["frac_issued_of_fp16_peak"] < 0.45
    assert [e["kernel"].split(" ")[0] for e in k] == ["k_rec_mfma<XIN>", "k_rec_fused", "k_head_combine"]
    # the scan's second half writes the probabilities itself (timing bit 9): no head kernel on the list, its bytes on the scan's
    k1, _ = bench.kernel_table(([2.0], [4.0], [0.01], [0.005], [6.2]), 2 | 256 | 512, split, 200, 10000, False)
    assert [e["kernel"].split(" ")[0] for e in k1] == ["k_rec_mfma<XIN>", "k_rec_fused"] and "softmax" in k1[1]["kernel"]
    assert abs(k1[1]["hbm_algorithmic_gb"] - by["k_rec_fused"]["hbm_algorithmic_gb"] - 2e6 * 20 / 1e9) < 1e-9
    assert k1[1]["issued_gflop"] == by["k_rec_fused"]["issued_gflop"]
    # the unfused latency regime: 4-window work-groups, GEMM and head as kernels of their own
    k2, step2 = bench.kernel_table(([6.6], [6.3], [3.6], [0.4], [13.0]), 0, {"chunks": 1, "columns": 10000}, 200, 10000, False)
    names = [e["kernel"].split(" ")[0] for e in k2]
    assert names == ["k_rec_mfma<XIN>", "k_rec_mfma", "k_gi_gemm", "k_head_tiled"] and step2["windows_per_work_group"] == 4
