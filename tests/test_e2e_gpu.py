"""End-to-end parity (-m gpu), row a9 of SURVEY.md section 8: the engine sits where the reference's
model sits -- inside `run_prediction`'s loop (prediction.py:44-52) -- and the samples it labels go
through trim/stitch (stitch.py:33-83) to a FASTQ that must equal, byte for byte, the FASTQ the
UNMODIFIED reference produced from the same pileups on PyTorch-CPU (tests/golden/stitch_cases.npz,
oracle/make_golden_stitch.py).  The caller and stitcher used here are the restatements of
oracle/stitch_oracle.py, pinned to the same goldens on the CPU (tests/test_oracle_stitch.py).

  (i)   single process: identical record names and bases, qualities within the documented bound, identical loop shape;
  (ii)  the same regions split across two shards by medaka_amd.sharding and the two stores joined
        as `medaka sequence a.hdf b.hdf` does: identical FASTQ;
  (iii) loop shape at production batch size: batches of 200 from a producer thread, a short last
        batch, B = 1 remainders of arbitrary length; every row bit-identical to a single-window call;
        other Python threads keep running while a forward is in flight (GIL released by ctypes)."""
import os
import threading
import time

import numpy as np
import pytest
import torch

from conftest import GOLD
from medaka_amd import models, sharding
from medaka_amd.torch_ext import Batch
from oracle import stitch_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sgold():
    return dict(np.load(os.path.join(GOLD, "stitch_cases.npz")))


@pytest.fixture(scope="module")
def model(gold):
    m = models.GRUModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gold["weights_trained"].items()})
    return m.to("cuda").eval()


class Spy:
    def __init__(self, model):
        self.model, self.shapes = model, []

    def predict_on_batch(self, batch):
        self.shapes.append(tuple(batch.counts_matrix.shape))
        out = self.model.predict_on_batch(batch)
        assert out.device.type == "cpu" and out.dtype == torch.float32
        return out


def _check_qualities(got, want):
    """Phred characters are floor(-10 log10(1 - p)) (labels.py:388-402): not protected by the 1e-4
    tolerance on p where p -> 1 ('Precision'), so they are held to: same length,
    >= 97 % identical characters, never more than 3 apart (one fp32 ulp of p at Q60+ moves Q by 2)."""
    a = np.frombuffer("".join(got).encode(), dtype=np.uint8).astype(int)
    b = np.frombuffer("".join(want).encode(), dtype=np.uint8).astype(int)
    assert a.shape == b.shape
    same = float((a == b).mean())
    print(f"quality characters identical: {same:.4%}, max |dQ| = {np.abs(a - b).max()}")
    assert same >= 0.97 and np.abs(a - b).max() <= 3


def _pileups(sources, spec):
    jitter = spec.get("jitter", ())
    return lambda r: so.pileups_in_region(sources, r, r.ref_name in jitter)


@pytest.mark.parametrize("case", ["mini", "cfg1"])
def test_stitched_fastq_identical_to_reference(sgold, model, case):
    spec, sources = so.load_case(sgold, case)
    spy = Spy(model)
    store = so.predict(so.contig_regions(sources), _pileups(sources, spec), spy, Batch.collate,
                       spec["chunk_len"], spec["chunk_ovlp"], spec["batch_size"], spec["bam_chunk"])
    assert sorted(spy.shapes) == sorted(tuple(b) for b in sgold[f"{case}/batches"])
    assert set(store) == set(sgold[f"{case}/written"].tolist())
    lengths = {r.ref_name: r.end for r in so.contig_regions(sources)}
    got, want = so.fastq(store, lengths), str(sgold[f"{case}/fastq"])
    gl, wl = got.split("\n"), want.split("\n")
    assert gl[0::4] == wl[0::4]                       # record names: contig breaks and coordinates
    assert gl[1::4] == wl[1::4], "consensus bases differ from the reference's"
    _check_qualities(gl[3::4], wl[3::4])


@pytest.mark.parametrize("case", ["mini", "cfg1"])
def test_sharded_run_joins_to_the_same_fastq(sgold, model, case):
    """Two `medaka inference --regions <shard>` processes, then `medaka sequence a b` (README.md:309-330)."""
    spec, sources = so.load_case(sgold, case)
    contigs = so.contig_regions(sources)
    shards = sharding.shard_regions([(r.ref_name, r.end) for r in contigs], 2, bam_chunk=spec["bam_chunk"],
                                    chunk_ovlp=spec["chunk_ovlp"])
    assert all(len(s) > 0 for s in shards)
    stores = []
    for regs in shards:
        stores.append(so.predict([so.Region(*r) for r in regs], _pileups(sources, spec), model, Batch.collate,
                                 spec["chunk_len"], spec["chunk_ovlp"], spec["batch_size"], spec["bam_chunk"]))
    joined = {}
    for st in stores:                                  # DataIndex over several files: first file wins a name
        for k, v in st.items():
            joined.setdefault(k, v)
    assert set(joined) == set(sgold[f"{case}/written"].tolist())
    assert len(stores[0]) > 0 and len(stores[1]) > 0
    lengths = {r.ref_name: r.end for r in contigs}
    got, want = so.fastq(joined, lengths).split("\n"), str(sgold[f"{case}/fastq"]).split("\n")
    assert got[0::4] == want[0::4] and got[1::4] == want[1::4]
    _check_qualities(got[3::4], want[3::4])


def test_half_precision_end_to_end_identity_rate(sgold, gold, model):
    """BASELINE config 5: `model.half()` end to end through stitch.  fp16 operands are outside the fp32 parity
    contract; what is reported and bounded is the consensus identity against the fp32 run (whose FASTQ is the
    reference's, see above): per-column argmax identity over every sample, and the stitched records."""
    mh = models.GRUModel()
    mh.load_state_dict({k: torch.from_numpy(v) for k, v in gold["weights_trained"].items()})
    mh = mh.to("cuda").eval().half()
    spec, sources = so.load_case(sgold, "cfg1")
    run = lambda mm: so.predict(so.contig_regions(sources), _pileups(sources, spec), mm, Batch.collate,
                                spec["chunk_len"], spec["chunk_ovlp"], spec["batch_size"], spec["bam_chunk"])
    half, full = run(mh), run(model)
    same = sum(int((half[k].label_probs.argmax(-1) == full[k].label_probs.argmax(-1)).sum()) for k in full)
    cols = sum(full[k].size for k in full)
    dp = max(float(np.abs(half[k].label_probs - full[k].label_probs).max()) for k in full)
    lengths = {r.ref_name: r.end for r in so.contig_regions(sources)}
    got, want = so.fastq(half, lengths).split("\n"), str(sgold["cfg1/fastq"]).split("\n")
    recs_same = sum(a == b for a, b in zip(got[1::4], want[1::4]))
    print(f"half precision end to end: argmax identity {same / cols:.6f} over {cols} columns, max|dp| {dp:.2e}, "
          f"{recs_same}/{len(want[1::4])} stitched records identical to the reference's fp32 consensus")
    assert got[0::4] == want[0::4]
    assert same / cols >= 0.9995


def test_production_loop_shape_bitwise_and_gil(gold, model):
    from medaka_amd import synth
    chunk_len, ovlp, B = 1000, 200, 200
    # 437 windows -> batches of 200, 200, 37; two narrow contigs -> B = 1 remainders of 777 and 13 columns
    n_cols = (437 - 1) * (chunk_len - ovlp) + chunk_len
    srcs = {}
    for name, cols, seed in (("big", n_cols, 71), ("r1", 777, 72), ("r2", 13, 73)):
        raw = synth.counts_windows(1, cols, depth=50, seed=seed, raw=True)
        feats = (raw["counts"][0] / np.maximum(1, raw["depth"][0])[:, None]).astype(np.float32)
        srcs[name] = [so.Pileup(name, feats, so.make_positions(raw["major"][0], raw["minor"][0]), None, raw["depth"][0])]
    seen = []

    def on_batch(loader, data, batch, probs):
        p = probs.numpy()
        seen.append(p.shape)
        for i in sorted({0, len(data) // 2, len(data) - 1}):          # single-window calls: bit-identical rows
            one = model.predict_on_batch(Batch.collate([data[i]])).numpy()
            assert np.array_equal(one[0], p[i]), (p.shape, i)
    store = so.predict(so.contig_regions(srcs), lambda r: so.pileups_in_region(srcs, r), model, Batch.collate,
                       chunk_len, ovlp, B, 10**9, on_batch=on_batch)
    assert seen[:3] == [(200, 1000, 5), (200, 1000, 5), (37, 1000, 5)]
    assert sorted(seen[3:]) == [(1, 13, 5), (1, 777, 5)]
    assert len(store) == 437 + 2
    # GIL: a pure-Python ticker thread must keep its free-running pace while forwards are in flight
    x = torch.from_numpy(synth.counts_windows(8, 10000, seed=5)).repeat(25, 1, 1)      # 200 x 10000
    batch = Batch(counts_matrix=x)
    model.predict_on_batch(batch)
    ticks, stop = [0], threading.Event()

    def ticker():
        while not stop.is_set():
            ticks[0] += 1
    t = threading.Thread(target=ticker, daemon=True)
    t.start()
    time.sleep(0.2)
    t0, a = time.perf_counter(), ticks[0]
    time.sleep(0.3)
    free_rate = (ticks[0] - a) / (time.perf_counter() - t0)
    t0, a = time.perf_counter(), ticks[0]
    for _ in range(10):
        model.predict_on_batch(batch)
    busy_rate = (ticks[0] - a) / (time.perf_counter() - t0)
    stop.set()
    t.join()
    print(f"ticker: {free_rate:,.0f}/s free, {busy_rate:,.0f}/s during forwards")
    assert busy_rate > 0.5 * free_rate


def test_writer_thread_holds_rows_of_earlier_batches(model):
    """The `DataStore` pattern (reference datastore.py:196, 263-299, 323-329): `run_prediction` hands every
    row of `predict_on_batch`'s output -- a VIEW of the returned tensor -- to a one-thread executor and goes
    on to the next batch at once (prediction.py:44-52), so rows of several earlier batches are still queued
    while new forwards run.  `predict_on_batch` returns page-locked tensors that torch's host allocator
    recycles (medaka_amd/models.py: _host_output): a block may only be reused once every view is gone.
    Here the writer is held back until six batches have been produced; every held row must then still carry
    exactly the bits of its own batch, and live outputs never share storage."""
    from concurrent.futures import ThreadPoolExecutor
    from medaka_amd import synth
    n_batches, B, T = 6, 24, 2304
    batches = [Batch(counts_matrix=torch.from_numpy(synth.counts_windows(B, T, depth=50, seed=900 + b)))
               for b in range(n_batches)]
    gate, written, futures = threading.Event(), {}, []
    executor = ThreadPoolExecutor(1)                 # DataStore.write_executor

    def write_dataset(location, data):               # DataStore._write_dataset: reads the tensor when its turn comes
        gate.wait()
        written[location] = data.numpy().copy()
    storages = []
    for b, batch in enumerate(batches):
        class_probs = model.predict_on_batch(batch)
        storages.append(class_probs.untyped_storage().data_ptr())
        for i, prob in enumerate(class_probs):       # prediction.py:47
            futures.append(executor.submit(write_dataset, (b, i), prob))
        del class_probs, prob
    assert len(set(storages)) == n_batches           # six outputs alive in the writer queue: six distinct blocks
    gate.set()
    for f in futures:
        f.result()
    executor.shutdown()
    del futures
    for b, batch in enumerate(batches):
        again = model.predict_on_batch(batch).numpy()
        for i in range(B):
            assert np.array_equal(written[(b, i)], again[i]), (b, i)
    # with the views gone the allocator hands the blocks out again: no growth in steady state
    later = {model.predict_on_batch(batches[0]).untyped_storage().data_ptr() for _ in range(4)}
    print(f"held blocks {len(set(storages))}, blocks seen after release {len(later)}")
    assert later & set(storages) or len(later) <= 2


def test_processes_sharing_the_gpu_return_the_same_bits(gold, model, tmp_path):
    """`launch.py --procs-per-gpu K`: K independent processes on ONE GPU, each told its share
    (MEDAKA_AMD_PROCS_PER_GPU -> engine option gpu_share -> 8-window work-groups where 4-window ones of all K
    would not fit the chip).  Three concurrent children must each return, batch after batch, exactly the bits
    a process that has the GPU to itself returns."""
    import subprocess
    import sys
    from conftest import ROOT
    from medaka_amd import synth
    x = synth.counts_windows(8, 1024, depth=50, seed=321)
    x = np.concatenate([x] * 30)                                   # 240 windows: 120 work-groups of 4 alone, 60 of 8 under sharing
    np.save(tmp_path / "x.npy", x)
    want = model.predict_on_batch(Batch(counts_matrix=torch.from_numpy(x))).numpy()
    child = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from medaka_amd import models\n"
        "from medaka_amd.torch_ext import Batch\n"
        f"st = dict(np.load({os.path.join(GOLD, 'weights_trained.npz')!r}))\n"
        "m = models.GRUModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}); m = m.to('cuda').eval()\n"
        "assert models.gpu_share() == 3\n"
        "x = torch.from_numpy(np.load(sys.argv[1]))\n"
        "outs = [m.predict_on_batch(Batch(counts_matrix=x)).numpy().copy() for _ in range(6)]\n"
        "assert all(np.array_equal(o, outs[0]) for o in outs)\n"
        "np.save(sys.argv[2], outs[-1])\n")
    env = dict(os.environ, MEDAKA_AMD_PROCS_PER_GPU="3")
    procs = [subprocess.Popen([sys.executable, "-c", child, str(tmp_path / "x.npy"), str(tmp_path / f"y{k}.npy")], env=env)
             for k in range(3)]
    assert [p.wait(timeout=300) for p in procs] == [0, 0, 0]
    for k in range(3):
        assert np.array_equal(np.load(tmp_path / f"y{k}.npy"), want), k


def test_two_rank_bench_on_one_gpu(tmp_path):
    """The N > 1 path of bench.py on hardware (what the driver's 8-GPU scaling run executes: torch.distributed.run, one
    rank per device, barrier + MAX over ranks, aggregate value) with both ranks on the one visible GPU (`--shared-gpu`:
    gloo for the barrier, every engine told it shares the device).  rc 0, one JSON line from rank 0, both ranks' split
    scans certified, and an aggregate in the band two processes on one MI355X can reach -- not a scaling figure."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MDK_SCAN_SPLIT", None)
    common = ["--steps", "3", "--warmup", "1", "--cpu-budget", "0", "--loop-batches", "0", "--host-reps", "3", "--extra-rl", "0"]
    # (stdout carries the compact driver line only; everything else the run measured is in the side file)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--shared-gpu", "--full-out", str(tmp_path / "one.json")] + common,
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    r1 = json.load(open(tmp_path / "one.json"))
    assert l1["value"] == r1["value"] and l1["roofline"]["frac"] == r1["roofline"]["frac"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29677", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--full-out", str(tmp_path / "two.json")] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                  # rank 0 alone reports
    l2 = json.loads(lines[0])
    assert len(lines[0]) <= 6144 and l2["n_gpus"] == 2 and l2["roofline"] and "cpu_baseline" in l2
    r2 = json.load(open(tmp_path / "two.json"))
    assert l2["value"] == r2["value"]
    assert r2["n_gpus"] == 2 and r2["scaling"] == "weak" and r2["steps"] == 3
    assert r2["scan_split"]["ranks_certified"] == 2 and r2["scan_split"]["chunks"] >= 2, r2["scan_split"]
    ratio = r2["value"] / r1["value"]
    print(f"two ranks sharing one MI355X: {r2['value'] / 1e6:.1f} M columns/s together against {r1['value'] / 1e6:.1f} M for one "
          f"({ratio:.2f} x); host-to-host {r2['host_to_host']['value'] / 1e6:.1f} M against {r1['host_to_host']['value'] / 1e6:.1f} M")
    assert 0.6 <= ratio <= 1.6, ratio
    assert r2["host_to_host"]["value"] > 0 and r2["roofline"]["kernels"]
    assert r2["barrier_backend"] == "gloo" and r2["ranks_seen"] == 2


def test_rccl_that_cannot_work_falls_back_on_hardware():
    """`bench.py --gpus 2 --dry-ranks` with two ranks on the ONE visible GPU and NO `--shared-gpu`: `dist.Ranks` takes the RCCL
    branch (what it does on the 8-GPU node), both ranks land on device 0, RCCL refuses -- or times out on -- duplicate
    devices, and every rank must meet again on gloo: rc 0, `barrier_backend` gloo, a reason, both ranks seen.  The only way a
    1-GPU box can execute the nccl branch of the N > 1 path at all."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29687", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"dry ranks on one GPU: {line}")
    assert line["dry_ranks"] and line["ranks_seen"] == 2, line
    # RCCL may also accept two ranks on one device on some stacks: then the line says nccl and has no reason
    assert (line["barrier_backend"] == "gloo") == bool(line["fallback_reason"]), line


def test_validate_tool_on_a_state_dict(gold, tmp_path):
    """`python -m medaka_amd.validate weights.npz` end to end on the device (small shape): both precisions, the margin
    table, the learned margin, every input structure against PyTorch-CPU."""
    from medaka_amd import validate
    rep = validate.main([os.path.join(GOLD, "weights_trained.npz"), "--batch", "24", "--chunk-len", "6000", "--sample-windows", "2",
                         "--json", str(tmp_path / "v.json")])
    # (half precision against the FP32 CPU result: the structured inputs reach 6e-4 on depth cliffs -- the band a CPU fp16
    # emulation of the reference itself deviates by, DESIGN.md 5.2 -- with every argmax identical)
    for prec, tol in (("fp32", 2e-5), ("half", 2e-3)):
        r = rep[prec]
        assert r["margin_table_iid"][128]["status"] == "certified" and r["smallest_certified_margin"] in (64, 96, 128), r["margin_table_iid"]
        assert r["learned"]["status"] == "certified" and r["learned"]["settled_at"] in (64, 96, 128)
        assert r["device_resident"]["columns_per_s"] > r["sequential_scan"]["columns_per_s"] > 0 and r["host_to_host"]["columns_per_s"] > 0
        # the loop the reference runs: a loader thread ahead of predict_on_batch, whose forwards are then started ahead of their calls
        assert r["fed_loop"]["columns_per_s"] > 0 and r["fed_loop"]["forwards_started_ahead"] >= r["fed_loop"]["batches"] // 2, r["fed_loop"]
        if prec == "half":
            assert r["learned"]["fp32_parity_probes"] >= 1, r["learned"]
        for kind, k in r["inputs"].items():
            assert k["status"] in ("certified", "rejected", "disabled", "not used")
            assert k["max_abs_dp_vs_cpu"] <= tol and k["argmax_identical"] >= 0.999 * k["columns_checked"], (prec, kind, k)
