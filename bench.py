#!/usr/bin/env python
"""Headline benchmark: pileup columns / second through the consensus network forward pass.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Metric and workload are BASELINE.json's: configs[1] = r1041_e82_400bps_sup-architecture
consensus (GRUModel 10 -> 2 x bi-GRU(128) -> Linear(5) -> softmax, reference
medaka/architectures/gru.py:46-72), synthetic 50x-depth pileup windows, batch 200 x 10000.
One "step" = one pass of the hot path (`GRUModel.forward`) over one batch whose input tensor
is already resident in HBM; `value` = all columns processed by all ranks / max-over-ranks wall
time of K steps.  Multi-GPU = independent replicas on disjoint window shards, no collective on
the data path (weak scaling: per-GPU batch fixed).

Also reported on the same JSON line:
  roofline      dominant kernel (k_rec_mfma, the GRU recurrence): algorithmic FLOP per launch
                / hipEvent-measured launch duration vs the fp16 dense MFMA peak / 4 (the fp32-parity
                split issues 4 fp16 MACs per algorithmic MAC); the native-fp32 fraction is kept beside it
  cpu_baseline  the reference's own CPU path (PyTorch-CPU nn.GRU/Linear/softmax, restated in
                oracle/oracle.py) timed on this box's host cores on a bounded sample
  parity        max |dp| and argmax identity of the engine vs that CPU result on the sample
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per pileup column
FLOP_PER_COLUMN = 804_352            # whole network
REC_FLOP_PER_COLUMN_LAYER = 196_608  # one layer's h->h recurrence, both directions (98 304 MAC)
PEAK_F32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32-in/fp32-acc MFMA = fp32 vector peak
PEAK_F16_DENSE_TFLOPS = 2500.0       # the pipe the fp16x2-split kernels actually issue on


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=200, help="windows per step per GPU")
    ap.add_argument("--chunk-len", type=int, default=10000, help="pileup columns per window")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--overlap", type=int, default=1, help="layer-1 projection GEMM under the tail of the layer-0 recurrence (0 off, 1 on)")
    ap.add_argument("--cpu-sample", type=int, default=8, help="windows in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--variant", type=int, default=None, help="MDK_VARIANT_* override (1 = exact fp32 kernels)")
    ap.add_argument("--tile", type=int, default=0, help="recurrence windows per work-group (0 auto, 4, 8, 16 = half precision only)")
    ap.add_argument("--half", action="store_true", help="model.half() path (reference GPU default)")
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(state, x_sample, probs_sample):
    """Reference CPU path on a bounded sample; also the parity check of the same run.

    PyTorch-CPU GRU scales poorly with threads (reference README.md:332-336 advises <= 2), so
    a short sweep over thread counts is timed and the fastest is reported with its count."""
    import numpy as np
    import torch
    from oracle import oracle
    cores = os.cpu_count() or 1
    m = oracle.make_torch_oracle(state)
    cols = x_sample.shape[0] * x_sample.shape[1]
    best, ref, sweep = None, None, {}
    for nthreads in sorted({t for t in (2, 8, 32) if t <= cores} | {min(cores, 2)}):
        torch.set_num_threads(nthreads)
        m.predict(x_sample[:1, :500])                # warm-up
        t0 = time.perf_counter()
        out = m.predict(x_sample)
        dt = time.perf_counter() - t0
        sweep[nthreads] = cols / dt
        log(f"cpu baseline: {nthreads} threads -> {cols / dt:,.0f} columns/s ({dt:.1f}s)")
        if best is None or cols / dt > best[1]:
            best = (nthreads, cols / dt)
        ref = out.numpy() if ref is None else ref
        if dt > 25:
            break
    base = {"value": best[1], "unit": "pileup columns/s", "cores": best[0], "kind": "port",
            "host_cores_available": cores, "thread_sweep_columns_per_s": sweep,
            "sample": f"{x_sample.shape[0]} windows x {x_sample.shape[1]} columns, PyTorch-CPU fp32 "
                      f"nn.GRU+Linear+softmax (the ops of reference gru.py:66-71), one timed pass "
                      f"per thread count, best reported"}
    parity = {"max_abs_dp": float(np.abs(probs_sample - ref).max()),
              "argmax_identical": bool((probs_sample.argmax(-1) == ref.argmax(-1)).all()),
              "tolerance": 1e-4, "columns_checked": int(cols)}
    return base, parity


def main():
    args = parse()
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.build()
    from medaka_amd import dist, models, synth

    log('start')
    ranks = dist.Ranks()
    if ranks.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ranks.world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback of the engine")
    dev = torch.device("cuda", ranks.local_rank)
    torch.cuda.set_device(dev)

    B, T = args.batch, args.chunk_len
    state = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trained.npz")))
    model = models.GRUModel()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    if args.half:
        model.half()
    model.kernel_variant = args.variant

    # synthetic 50x pileup windows, one distinct shard per rank (seeded by rank); generated in
    # tiles of 8 windows and repeated to B to keep host set-up time small
    base_tiles = min(B, 40)
    x_host = synth.counts_windows(base_tiles, T, depth=args.depth, seed=1234 + ranks.rank)
    reps = -(-B // base_tiles)
    x_host = np.concatenate([x_host] * reps)[:B]
    log('synthetic input ready')
    x_dev = torch.from_numpy(x_host).to(dev)
    eng = model.engine()
    eng.set_option("rec_windows_per_tile", args.tile)
    eng.set_option("overlap_gemm", args.overlap)
    eng.enable_timing(True)

    out_holder = {}

    def step():
        with torch.inference_mode():
            out_holder["y"] = model.forward(x_dev)

    rec_ms, total_ms, gi_ms, head_ms = [], [], [], []

    def step_timed():
        step()
        t = eng.timing()
        rec_ms.extend(t["rec_ms"])
        gi_ms.append(sum(t["gi_ms"]))
        head_ms.append(t["head_ms"])
        total_ms.append(t["total_ms"])

    log('engine ready, timing')
    elapsed, mine = dist.timed_steps(ranks, step_timed, lambda: torch.cuda.synchronize(dev),
                                     steps=args.steps, warmup=args.warmup)
    log(f'timed region done: {elapsed:.3f}s for {args.steps} steps')
    # keep only the timed steps' kernel records
    n_layers = len(eng.timing()["rec_ms"])
    rec_ms = rec_ms[-args.steps * n_layers:]
    cols_per_step = B * T
    value = ranks.world * cols_per_step * args.steps / elapsed

    result = {
        "metric": "pileup columns/sec (consensus bi-GRU inference)",
        "value": value, "unit": "pileup columns/s", "n_gpus": ranks.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.half else "f32",
        "data": "synthetic",
        "config": {"workload": f"r1041_e82_400bps_sup-architecture consensus (GRUModel 10->2x biGRU128->5), "
                               f"synthetic {args.depth}x pileup windows, batch {B} x {T} columns per GPU, "
                               f"input resident in HBM, probabilities left in HBM",
                   "batch_windows": B, "chunk_len": T, "columns_per_step_per_gpu": cols_per_step,
                   "weights": "tests/golden/weights_trained.npz (reference-trained on synthetic data; "
                              "published model archives are git-LFS stubs offline)",
                   "parallelism": f"{ranks.world} independent replicas, window-sharded, no collective"},
    }
    if ranks.rank == 0:
        rec_avg_ms = statistics.mean(rec_ms)
        rec_flop = REC_FLOP_PER_COLUMN_LAYER * cols_per_step
        achieved = rec_flop / (rec_avg_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_rec_mfma_bytes_per_launch")
            except Exception:
                traffic = None
        # fp32 parity is bought with an fp16 hi/lo split: 4 fp16 MACs are issued per algorithmic MAC,
        # so the pipe that bounds this kernel is the fp16 dense MFMA pipe at a quarter of its rate
        # (half precision mode issues 1 MAC per MAC and is priced against the full rate).
        issue_factor = 1 if args.half else 4
        peak = PEAK_F16_DENSE_TFLOPS / issue_factor
        result["roofline"] = {
            "kernel": "k_rec_mfma (GRU recurrence; figures are per LAYER PASS = all windows, both directions, T steps; "
                      "with the layer-1 projection overlapped, layer 0's pass is 7 resumable launches of the same "
                      "kernel, whose rocprof durations add up to this span)",
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic,
            "avg_launch_ms": rec_avg_ms, "launches_timed": len(rec_ms),
            "kernel_launches_per_step": eng.timing()["rec_launches"],
            "algorithmic_flop_per_launch": rec_flop,
            "note": f"peak = fp16 dense MFMA {PEAK_F16_DENSE_TFLOPS:.0f} TFLOP/s / {issue_factor} fp16 MACs issued per "
                    "algorithmic MAC; a native fp32-MFMA kernel would be capped at "
                    f"{PEAK_F32_MATRIX_TFLOPS} TFLOP/s, of which this launch reaches {achieved / PEAK_F32_MATRIX_TFLOPS:.3f}",
            "frac_of_f32_matrix_peak": achieved / PEAK_F32_MATRIX_TFLOPS,
            "whole_network_tflops": FLOP_PER_COLUMN * cols_per_step / (statistics.mean(total_ms) * 1e-3) / 1e12,
            "kernel_ms_per_step": {"rec": sum(rec_ms) / args.steps, "gi": statistics.mean(gi_ms[-args.steps:]),
                                   "head": statistics.mean(head_ms[-args.steps:]),
                                   "device_total": statistics.mean(total_ms[-args.steps:])},
        }
        if args.cpu_sample > 0 and ranks.world == 1:   # the CPU baseline is a single-GPU-run figure
            n = min(args.cpu_sample, B)
            probs = out_holder["y"][:n].cpu().numpy()
            result["cpu_baseline"], result["parity"] = cpu_baseline(state, x_host[:n], probs)
            result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
        # host tensor in -> host tensor out (what predict_on_batch does), for DESIGN.md only
        t0 = time.perf_counter()
        eng.enable_timing(False)
        from medaka_amd.torch_ext import Batch
        xb = Batch(counts_matrix=torch.from_numpy(x_host))
        model.predict_on_batch(xb)                      # first call sizes the engine's staging buffers
        t0 = time.perf_counter()
        model.predict_on_batch(xb)
        result["pcie_inclusive_columns_per_s"] = cols_per_step / (time.perf_counter() - t0)
        # the same with the PCIe diet (SURVEY 8f f2 + f3): uint16 counts + uint32 depth in (24 B/column),
        # argmax class + its probability out (5 B/column)
        import numpy as np
        cnt = np.minimum(np.rint(x_host * 60.0), 65535).astype(np.uint16)
        dep = np.full(x_host.shape[:2], 60, dtype=np.uint32)
        model.predict_on_counts(cnt, dep, decoded=True)
        t0 = time.perf_counter()
        model.predict_on_counts(cnt, dep, decoded=True)
        result["pcie_diet_columns_per_s"] = cols_per_step / (time.perf_counter() - t0)
        print(json.dumps(result), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
