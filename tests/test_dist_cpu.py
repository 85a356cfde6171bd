"""N>1 path on CPU: two gloo ranks run the bench's sharding + barrier + max-over-ranks timing."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, os.environ["MDK_ROOT"])
from medaka_amd import dist, sharding
ranks = dist.Ranks(backend="gloo")
lo, hi = sharding.shard_windows(11, ranks.world, ranks.rank)
calls = []
def step():
    calls.append(1)
    time.sleep(0.01 * (ranks.rank + 1))      # rank 1 is slower: max must pick it
elapsed_max, mine = dist.timed_steps(ranks, step, lambda: None, steps=3, warmup=1)
total = ranks.sum_over_ranks(hi - lo)
if ranks.rank == 0:
    print(json.dumps({"max": elapsed_max, "mine": mine, "calls": len(calls), "total": total,
                      "world": ranks.world, "lo_hi": [lo, hi]}))
ranks.close()
"""


def test_two_rank_gloo_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MDK_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["calls"] == 4 and r["total"] == 11 and r["lo_hi"] == [0, 6]
    assert r["max"] >= 0.055          # slow rank: 3 * 0.02 s
    assert r["mine"] < r["max"]       # rank 0 is the fast one


FALLBACK_WORKER = r"""
import json, os, sys
sys.path.insert(0, os.environ["MDK_ROOT"])
from medaka_amd import dist
ranks = dist.Ranks(backend="nccl")          # no GPU here: the RCCL attempt must fail on every rank and all of them move to gloo
ranks.barrier()
seen = ranks.ranks_seen()
mx = ranks.max_over_ranks(10.0 + ranks.rank)
if ranks.rank == 0:
    print(json.dumps({"backend": ranks.barrier_backend, "reason": ranks.fallback_reason, "seen": seen, "max": mx}))
ranks.close()
"""


def test_nccl_failure_falls_back_to_gloo_on_every_rank(tmp_path):
    """`dist.Ranks` tries RCCL where it is asked to (or where there are GPUs); when that cannot work -- here: no device at
    all -- every rank must end up in the SAME gloo group, with a working barrier / MAX / SUM (what bench.py --gpus N needs)."""
    script = tmp_path / "worker.py"
    script.write_text(FALLBACK_WORKER)
    env = dict(os.environ, MDK_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29543", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "gloo" and r["seen"] == 2 and r["max"] == 11.0 and "no HIP device" in r["reason"], r
    assert "RCCL process group not usable" in out.stderr


def test_bench_dry_ranks_line():
    """`bench.py --gpus N --dry-ranks`: process-group set-up, barrier and reductions only -- what a first 8-GPU launch can be
    checked with before any model is built."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29553", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["dry_ranks"] and r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["barrier_backend"] == "gloo", r


def test_bench_dry_ranks_world_8():
    """The driver's scaling run is `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`: eight ranks must meet (here on
    gloo: no device in the build container, so every rank takes the RCCL-failed branch together), count each other and agree
    on a MAX -- the whole of the N > 1 control path short of a device."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29563", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-ranks"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                  # rank 0 alone reports
    r = json.loads(lines[0])
    assert r["dry_ranks"] and r["n_gpus"] == 8 and r["ranks_seen"] == 8 and r["barrier_backend"] == "gloo", r


def test_sharding_config3_one_long_contig_on_8_shards():
    """BASELINE config 3's shape -- ONE 250 Mb contig -- on 8 GPUs: the shards' loads are within one `bam_chunk` piece of each
    other, every piece is a piece of the single-process grid (prediction.py:100-110 -> Region.split, common.py:711-736), and
    their union IS that grid (nothing dropped, nothing doubled; only a short tail may ride with its predecessor)."""
    from medaka_amd import sharding
    L, chunk, ovlp = 250_000_000, 1_000_000, 1000
    grid = sharding.split_region(sharding.Region("chr1", 0, L), chunk, ovlp)
    shards = sharding.shard_regions([("chr1", L)], 8, bam_chunk=chunk, chunk_ovlp=ovlp, chunk_len=10000)
    assert len(shards) == 8 and all(shards)
    loads = [sum(r.end - r.start for r in s) for s in shards]
    assert max(loads) - min(loads) <= chunk, loads
    flat = sorted((r for s in shards for r in s), key=lambda r: r.start)
    starts = [r.start for r in flat]
    assert starts == sorted(set(starts))                                   # no piece twice
    grid_starts = [r.start for r in grid]
    assert set(starts) <= set(grid_starts)
    # re-cut every shard region the way its child process does: the union is the single-process grid
    recut = sorted((p.start, p.end) for r in flat for p in sharding.split_region(r, chunk, ovlp))
    assert recut == sorted((p.start, p.end) for p in grid)
    # deterministic, and input order inside a shard
    assert shards == sharding.shard_regions([("chr1", L)], 8, bam_chunk=chunk, chunk_ovlp=ovlp, chunk_len=10000)
    assert all([r.start for r in s] == sorted(r.start for r in s) for s in shards)
