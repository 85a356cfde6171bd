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
                      "--chunk_ovlp", "200", "--inference-cmd", stub, "--", "--full_precision"])
    assert rc == 0
    jobs = [json.load(open(out / f"shard_{i}.hdf")) for i in range(2)]
    assert [j["gpu"] for j in jobs] == ["0", "1"] and all(j["amd"] == "1" and j["cuda_visible"] is None for j in jobs)
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
