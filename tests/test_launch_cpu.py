"""The N-GPU launcher (medaka_amd/launch.py) on CPU: two children of a stub `medaka inference`."""
import json
import os
import sys

from medaka_amd import launch, sharding
from oracle import stitch_oracle as so

STUB = r'''
import json, os, sys, time
args = sys.argv[1:]
bam, hdf = args[0], args[1]
bed = args[args.index("--regions") + 1]
regions = [l.split() for l in open(bed)]
if os.environ.get("STUB_FAIL") == os.environ["HIP_VISIBLE_DEVICES"]:
    sys.exit(3)
if os.environ.get("STUB_SLOW"):
    time.sleep(float(os.environ["STUB_SLOW"]))
json.dump({"gpu": os.environ["HIP_VISIBLE_DEVICES"], "amd": os.environ.get("MEDAKA_AMD"), "bam": bam,
           "share": os.environ.get("MEDAKA_AMD_PROCS_PER_GPU"), "scan_split": os.environ.get("MDK_SCAN_SPLIT"),
           "regions": regions, "args": args, "cuda_visible": os.environ.get("CUDA_VISIBLE_DEVICES")}, open(hdf, "w"))
'''


def _setup(tmp_path):
    draft = tmp_path / "draft.fa"
    draft.write_text(">chrA desc\n" + "ACGT" * 2600 + "\n>chrB\n" + "A" * 3100 + "\n>tiny\nACGTAC\n")
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    return str(draft), f"{sys.executable} {stub}"


def test_two_gpu_launch_partitions_the_reference_grid(tmp_path):
    draft, stub = _setup(tmp_path)
    out = tmp_path / "out"
    assert launch.contig_lengths(draft) == [("chrA", 10400), ("chrB", 3100), ("tiny", 6)]
    rc = launch.main(["calls.bam", draft, str(out), "--gpus", "2", "--model", "m", "--bam_chunk", "4000",
                      "--chunk_ovlp", "200", "--chunk_len", "100", "--inference-cmd", stub, "--", "--full_precision"])
    assert rc == 0
    jobs = [json.load(open(out / f"shard_{i}.hdf")) for i in range(2)]
    assert [j["gpu"] for j in jobs] == ["0", "1"] and all(j["amd"] == "strict" and j["share"] == "1" and j["cuda_visible"] is None for j in jobs)
    assert all(j["args"][-1] == "--full_precision" and "--model" in j["args"] for j in jobs)
    got = sorted((n, int(a), int(b)) for j in jobs for n, a, b in j["regions"])
    want = []
    for name, length in (("chrA", 10400), ("chrB", 3100), ("tiny", 6)):      # what ONE medaka inference cuts
        want.extend(tuple(r) for r in so.split_region(so.Region(name, 0, length), 4000, 200))
    assert got == sorted(want)
    assert all(len(j["regions"]) > 0 for j in jobs)
    loads = [sum(int(b) - int(a) for _, a, b in j["regions"]) for j in jobs]
    assert max(loads) <= 1.35 * min(loads)
    assert os.path.exists(out / "shard_0.log")


def test_failing_child_stops_the_others(tmp_path, monkeypatch):
    draft, stub = _setup(tmp_path)
    monkeypatch.setenv("STUB_FAIL", "1")
    monkeypatch.setenv("STUB_SLOW", "30")
    import time
    t0 = time.time()
    rc = launch.main(["calls.bam", draft, str(tmp_path / "out"), "--gpus", "2", "--bam_chunk", "4000",
                      "--chunk_ovlp", "200", "--inference-cmd", stub])
    assert rc == 1 and time.time() - t0 < 20
    assert not os.path.exists(tmp_path / "out" / "shard_0.hdf")


def test_dry_run_and_region_subset(tmp_path, capsys):
    draft, stub = _setup(tmp_path)
    rc = launch.main(["calls.bam", draft, str(tmp_path / "o"), "--gpus", "4", "--regions", "chrB", "chrA:100-900",
                      "--dry-run", "--inference-cmd", stub])
    assert rc == 0
    text = capsys.readouterr().out
    assert "medaka sequence" in text and text.count("HIP_VISIBLE_DEVICES=") == 2      # two regions -> two jobs
    beds = sorted(f for f in os.listdir(tmp_path / "o") if f.endswith(".bed"))
    regs = [open(tmp_path / "o" / b).read().split() for b in beds]
    assert sorted(map(tuple, regs)) == [("chrA", "100", "900"), ("chrB", "0", "3100")]


def test_procs_per_gpu_shares_each_gpu_between_k_children(tmp_path, monkeypatch):
    """`--procs-per-gpu 2` on 2 GPUs: four children on four disjoint shards of the same grid, two per GPU, each told
    that it shares its GPU; a launcher that is itself restricted to GPUs 5,7 sends its children there."""
    import pytest
    draft, stub = _setup(tmp_path)
    out = tmp_path / "out"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,7")
    rc = launch.main(["calls.bam", draft, str(out), "--gpus", "2", "--procs-per-gpu", "2", "--bam_chunk", "3000",
                      "--chunk_ovlp", "200", "--chunk_len", "100", "--inference-cmd", stub])
    assert rc == 0
    jobs = [json.load(open(out / f"shard_{i}.hdf")) for i in range(4)]
    assert [j["gpu"] for j in jobs] == ["5", "7", "5", "7"]
    assert all(j["share"] == "2" and j["amd"] == "strict" for j in jobs)
    got = sorted((n, int(a), int(b)) for j in jobs for n, a, b in j["regions"])
    want = []
    for name, length in (("chrA", 10400), ("chrB", 3100), ("tiny", 6)):
        want.extend(tuple(r) for r in so.split_region(so.Region(name, 0, length), 3000, 200))
    assert got == sorted(want)                       # the union is still what ONE medaka inference cuts
    assert all(os.path.exists(out / f"shard_{i}.log") for i in range(4))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")
    with pytest.raises(ValueError):                  # two GPUs asked for, one visible
        launch.main(["calls.bam", draft, str(tmp_path / "o2"), "--gpus", "2", "--inference-cmd", stub, "--dry-run"])
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    with pytest.raises(SystemExit):                  # the cluster recurrence of rl_lstm384 cannot share its GPU
        launch.parse(["b.bam", draft, "o", "--procs-per-gpu", "2", "--model", "r1041_e82_400bps_sup_v5.2.0_rl_lstm384_dwells"])
    assert launch.parse(["b.bam", draft, "o", "--lenient"]).lenient


def test_reproducible_children_run_sequential_scans(tmp_path, monkeypatch):
    """`--reproducible`: children get MDK_SCAN_SPLIT=0 (the sequential scan's bits do not depend on how windows are
    batched, so shard HDFs equal a single-process run's bit for bit); without it the variable is left alone."""
    monkeypatch.delenv("MDK_SCAN_SPLIT", raising=False)
    draft, stub = _setup(tmp_path)
    for flag, want in ((["--reproducible"], "0"), ([], None)):
        out = tmp_path / ("out" + "".join(flag))
        assert launch.main(["calls.bam", draft, str(out), "--gpus", "2", "--inference-cmd", stub] + flag) == 0
        assert [json.load(open(out / f"shard_{i}.hdf"))["scan_split"] for i in range(2)] == [want, want]


def test_region_strings_follow_the_reference(tmp_path):
    """`Region.from_string` semantics (reference common.py:670-710), clipped to the sequence."""
    import pytest
    lengths = {"chrA": 10400, "A:B:c": 900, "odd:5-7": 50}
    R = sharding.Region
    assert launch.parse_region("chrA", lengths) == R("chrA", 0, 10400)
    assert launch.parse_region("chrA:1000-2000", lengths) == R("chrA", 1000, 2000)
    assert launch.parse_region("chrA:1000", lengths) == R("chrA", 1000, 10400)
    assert launch.parse_region("chrA:1000-", lengths) == R("chrA", 1000, 10400)
    assert launch.parse_region("chrA:-5000", lengths) == R("chrA", 0, 5000)
    assert launch.parse_region("chrA:9000-99999", lengths) == R("chrA", 9000, 10400)
    assert launch.parse_region("A:B:c:500-", lengths) == R("A:B:c", 500, 900)
    assert launch.parse_region("odd:5-7", lengths) == R("odd:5-7", 0, 50)      # a whole name wins
    with pytest.raises(KeyError):
        launch.parse_region("nope:1-2", lengths)
    with pytest.raises(ValueError):
        launch.parse_region("chrA:x-y", lengths)


def test_short_tail_travels_with_its_predecessor():
    """a trailing piece shorter than chunk_len must not become a region of its own (the child would
    run it un-chunked, prediction.py:97-98); joined to its predecessor the child re-cuts the same two pieces."""
    R = sharding.Region
    # 2 500 000 bases, bam_chunk 1 000 000, overlap 1000: pieces start every 999 000; the last is [1998000, 2500000)
    # -> long enough, nothing changes
    a = sharding.shardable_pieces(R("c", 0, 2_500_000), 1_000_000, 1000, 10000)
    assert a == sharding.split_region(R("c", 0, 2_500_000), 1_000_000, 1000)
    # 2 003 000 bases: the last piece [1998000, 2003000) is 5000 < chunk_len
    b = sharding.shardable_pieces(R("c", 0, 2_003_000), 1_000_000, 1000, 10000)
    assert b == [R("c", 0, 1_000_000), R("c", 999_000, 2_003_000)]
    # ... and the child, cutting [999000, 2003000) for itself, gets exactly the single-process pieces back
    child = [tuple(x) for x in so.split_region(so.Region("c", 999_000, 2_003_000), 1_000_000, 1000)]
    assert child == [tuple(x) for x in sharding.split_region(R("c", 0, 2_003_000), 1_000_000, 1000)[1:]]
    assert child == [("c", 999_000, 1_999_000), ("c", 1_998_000, 2_003_000)]
    # fewer than chunk_ovlp bases left over: the reference's redundant tail lies inside its predecessor and stays
    c = sharding.shardable_pieces(R("c", 0, 1_998_500), 1_000_000, 1000, 10000)
    assert c == sharding.split_region(R("c", 0, 1_998_500), 1_000_000, 1000) and c[-1] == R("c", 1_998_000, 1_998_500)
    shards = sharding.shard_regions([("c", 2_003_000)], 2)
    assert sorted(r for s in shards for r in s) == b
