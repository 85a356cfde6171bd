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
  host_to_host  (= metric_8d) the rate SURVEY.md section 8d defines: host tensor in -> host tensor out through
                `model.predict_on_batch` (reference models.py:303-313): median of >= 5 timed batches after
                0.6 s of untimed ones (the first six are on the line as they came: `first_calls_ms`)
  fed_loop      the same inside the thread structure of the reference's inference loop
  roofline      dominant kernel (k_rec_fused: layer 1's projection + recurrence + classifier head in one
                kernel; k_rec_mfma on the sequential scan): algorithmic FLOP per layer pass / hipEvent-measured
                duration vs the fp16 dense MFMA peak / the fp16 MACs the fp32-parity split issues per
                algorithmic MAC (3.33; 4 for k_rec_mfma); one entry per hot kernel under `kernels`, the whole
                forward under `step`; the native-fp32 fraction is kept beside it
  cpu_baseline  the reference's own CPU path (B in {10, 100, 200} x threads in
                {1, 2, cpu_count}, 1 warm-up + median of 3, under a wall-clock cap) -- the unmodified
                reference class when /root/reference is present, its PyTorch-CPU restatement otherwise
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
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="wall-clock cap of the CPU baseline in seconds (0 = skip)")
    ap.add_argument("--variant", type=int, default=None, help="MDK_VARIANT_* override (1 = exact fp32 kernels)")
    ap.add_argument("--tile", type=int, default=0, help="recurrence windows per work-group (0 auto, 4, 8, 16 = half precision only)")
    ap.add_argument("--half", action="store_true", help="model.half() path (reference GPU default)")
    ap.add_argument("--parity-ref", default=None,
                    help="path of a .npy with reference probabilities (n, T, 5) of the first n windows of this run's batch: `parity` is "
                         "then this run's host-to-host result against it (how the fp32 line hands its CPU result to its --half child)")
    ap.add_argument("--extra-half", type=int, default=1,
                    help="1 (default): the fp32-parity line also carries `extra.half`, the same batch with model.half() -- what the "
                         "reference CLI selects on a GPU unless --full_precision (prediction.py:164-168); 0 = skip")
    ap.add_argument("--deferred-store", type=int, default=None, help="recurrence: store h_t from inside step t+1")
    ap.add_argument("--scan-split", type=int, default=None,
                    help="split scan (include/medaka_amd.h \"scan_split\"): default = the engine's (1, auto); 0 = the sequential scan "
                         "only; n >= 2 = force n chunks per window")
    ap.add_argument("--scan-split-margin", type=int, default=None, help="warm-up columns on either side of a chunk (default: the engine's, 128)")
    ap.add_argument("--model", default="gru", choices=["gru", "rl128", "rl384"],
                    help="gru: the headline consensus model; rl128 / rl384: read-level models (BASELINE config 4b)")
    ap.add_argument("--rl-depth", type=int, default=50, help="read-level models: reads per window")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="dry check of the N > 1 path on a 1-GPU box: every rank uses device 0, the timing barrier runs "
                         "over gloo (RCCL refuses two ranks on one device); the value is NOT a scaling measurement")
    ap.add_argument("--procs-per-gpu", type=int, default=1,
                    help="K > 1: re-launch as K ranks of `--shared-gpu` on ONE GPU (medaka_amd.launch --procs-per-gpu K): the line's "
                         "`value` / `host_to_host` are the aggregates of the K processes, `n_gpus` counts the ranks")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="N > 1 smoke check: initialise the process group (RCCL, or gloo if RCCL is not usable), barrier, MAX / SUM over "
                         "the ranks, print one JSON line with `barrier_backend` and `ranks_seen`, exit -- no model, no GPU work")
    ap.add_argument("--device-only", action="store_true",
                    help="profiling runs: only the device-resident timed steps (no host-to-host, CPU baseline, PCIe diet)")
    ap.add_argument("--stream-host", type=int, default=None, help="host path: 1 (default) = copies in time slabs under the recurrences / a split call's result under the second half of its last scan; 0 = one copy each side; 2 = a split call's result behind a side-stream head kernel (experiments)")
    ap.add_argument("--pinned-input", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--pageable-input", action="store_true",
                    help="host-to-host batches only from a pageable input tensor (the reference's collate) instead of the page-locked "
                         "one the engine's Batch.collate produces")
    ap.add_argument("--host-reps", type=int, default=7, help="timed host-to-host batches (median reported)")
    ap.add_argument("--extra-rl", type=float, default=15.0,
                    help="seconds of CPU baseline granted to the rl_lstm384 line that the default run appends under `extra` (0 = no extra line)")
    ap.add_argument("--margin-256", type=int, default=0,
                    help="1: also time 3 steps at a margin of 256 columns (`value_at_margin_256`: what a model with a longer memory runs at); "
                         "off by default -- the larger workspace it allocates hands the old one back to the driver")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "bench_full.json"),
                    help="where the complete record goes (stdout carries only the compact line the driver parses)")
    ap.add_argument("--loop-batches", type=int, default=32,
                    help="batches of the fed loop (threaded loader -> collate -> predict_on_batch -> writer; 0 = skip)")
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(weights_path, x_host, probs_sample, budget_s):
    """BASELINE.md section 4 on this box's host cores, bounded by `budget_s` of wall clock.

    Each (batch, threads) configuration runs in a child process (oracle/cpu_baseline.py: 1 short warm-up,
    up to 3 timed passes, median) that is killed when its share of the budget is over; what finished
    before a kill is kept, what did not fit is listed.  Plan: B in {10, 100, 200} x threads in
    {1, 2, all cores}, cheapest first."""
    import subprocess
    import tempfile
    import numpy as np
    cores = usable_cores()
    B_all, T = x_host.shape[0], x_host.shape[1]
    t_start = time.perf_counter()
    left = lambda: budget_s - (time.perf_counter() - t_start)
    thread_set = sorted({1, min(2, cores), cores})
    plan = [(10, cores), (10, 2), (10, 1), (100, cores), (200, cores), (100, 2), (200, 2), (100, 1), (200, 1)]
    plan = [(b, t) for b, t in plan if b <= B_all and t in thread_set]
    table, skipped, ref, kind = [], [], None, "port"
    est_rate = {}
    with tempfile.TemporaryDirectory() as tmp:
        xin = os.path.join(tmp, "x.npy")
        np.save(xin, x_host[:max(b for b, _ in plan)])
        for b, nthreads in plan:
            cols = b * T
            rate = est_rate.get(nthreads)
            if left() < 5 or (rate is not None and 1.2 * cols / rate > left()):
                skipped.append({"batch": b, "threads": nthreads, "reason": "wall-clock cap"})
                continue
            outp = os.path.join(tmp, f"ref_{b}_{nthreads}.npy")
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--weights", weights_path,
                   "--input", xin, "--batch", str(b), "--threads", str(nthreads), "--out", outp]
            lines = []
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=max(5.0, min(left(), 0.5 * budget_s)))
                lines = r.stdout.splitlines()
            except subprocess.TimeoutExpired as e:
                lines = (e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")).splitlines()
            passes = [json.loads(l) for l in lines if l.startswith("{")]
            if not passes:
                skipped.append({"batch": b, "threads": nthreads, "reason": "no pass finished inside its time slice"})
                est_rate[nthreads] = 1.0          # do not try larger batches with this thread count
                continue
            kind = passes[0]["kind"]
            med = statistics.median(p["seconds"] for p in passes)
            est_rate[nthreads] = cols / med
            table.append({"batch": b, "threads": nthreads, "columns_per_s": cols / med, "median_s": med, "passes": len(passes)})
            log(f"cpu baseline B={b} threads={nthreads}: {cols / med:,.0f} columns/s (median of {len(passes)}, {med:.2f}s)")
            if os.path.exists(outp) and (ref is None or b > ref.shape[0]):
                ref = np.load(outp)
    if not table:
        return ({"value": None, "unit": "pileup columns/s", "cores": 0, "kind": kind, "not_run": skipped,
                 "sample": "no configuration finished inside the wall-clock cap"}, None, None)
    best = max(table, key=lambda r: r["columns_per_s"])
    base = {"value": best["columns_per_s"], "unit": "pileup columns/s", "cores": best["threads"], "kind": kind,
            "host_cores_available": cores, "os_cpu_count": os.cpu_count(), "table": table, "not_run": skipped,
            "wall_clock_cap_s": budget_s,
            "passes": best["passes"],
            "sample": f"B={best['batch']} x {T}, {best['threads']} torch threads, fp32 PyTorch-CPU, {best['passes']} pass(es) + warm-up; best cell "
                      f"of BASELINE.md s4 grid in {budget_s:.0f} s",
            "sample_note": f"the box shows os.cpu_count() = {os.cpu_count()}, {cores} usable; kind = "
                           + ("reference: the unmodified reference GRUModel.predict_on_batch (/root/reference present)" if kind == "reference" else
                              "port: the three torch calls of reference gru.py:66-71 restated (nn.GRU + Linear + softmax; /root/reference does not "
                              "exist on the GPU box)")}
    n = ref.shape[0]
    parity = {"max_abs_dp": float(np.abs(probs_sample[:n] - ref).max()),
              "argmax_identical": bool((probs_sample[:n].argmax(-1) == ref.argmax(-1)).all()),
              "tolerance": 1e-4, "columns_checked": int(n * T)}
    return base, parity, ref


class LoopSample:
    """What the prediction loop reads of a `medaka.common.Sample` (common.py:60-80): a name and the features view."""
    __slots__ = ("name", "features", "labels")

    def __init__(self, name, features):
        self.name, self.features, self.labels = name, features, None


def reference_collate(samples):
    """The expression of the reference's `Batch.collate` for counts matrices (torch_ext.py:147-148)."""
    import torch
    from medaka_amd.torch_ext import Batch
    return Batch(counts_matrix=torch.stack([torch.from_numpy(s.features) for s in samples]).float())


def fed_loop(model, windows, batch_size, n_batches, collate, warm=2, cache=8, sample_workers=2, per_sample_submit=False, detail=False):
    """The engine inside the thread structure of the reference's inference loop (prediction.py:36-60, 225-370):
    `sample_workers` loader threads put Samples (views of a region's feature array, as `Sample.chunks` makes them)
    on a bounded queue, ONE Batcher thread groups `batch_size` of them and runs `collate`, the main thread calls
    `model.predict_on_batch` and hands every row of the result to a one-thread writer (DataStore.write_executor,
    datastore.py:196) that copies it out -- the minimum an HDF5 write does.  By default the main thread hands a
    batch's rows to the writer with ONE submit; `per_sample_submit` does it the reference's way, one executor submit
    per sample from the main thread (the reference makes one per sample FIELD, datastore.py:283-300), which costs the
    main thread ~45 us each under GIL contention.  Queues block instead of spinning (the reference polls with
    get_nowait).  Returns per-batch timings; the first `warm` batches are not counted."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    total = n_batches * batch_size
    samples_q, batches_q = queue.Queue(maxsize=cache * batch_size), queue.Queue(maxsize=cache)
    DONE = object()

    def region_worker(k):
        for i in range(k, total, sample_workers):
            samples_q.put((i, LoopSample(f"w{i}", windows[i % len(windows)])))
        samples_q.put(DONE)
    collate_ms = []

    def batch_worker():
        pending, stops, nxt = {}, 0, 0
        data = []
        while stops < sample_workers or pending:
            if stops < sample_workers:
                item = samples_q.get()
                if item is DONE:
                    stops += 1
                else:
                    pending[item[0]] = item[1]
            while nxt in pending:                      # samples in region order, as one loader per region yields them
                data.append(pending.pop(nxt))
                nxt += 1
                if len(data) == batch_size:
                    t0 = time.perf_counter()
                    batch = collate(data)
                    collate_ms.append(1e3 * (time.perf_counter() - t0))
                    batches_q.put((data, batch))
                    data = []
        batches_q.put(DONE)
    threads = [threading.Thread(target=region_worker, args=(k,), daemon=True) for k in range(sample_workers)]
    threads.append(threading.Thread(target=batch_worker, daemon=True))
    for t in threads:
        t.start()
    writer = ThreadPoolExecutor(1)
    sink = np.empty(windows[0].shape[0] * 5, dtype=np.float32)
    touched = [0]

    def write_row(prob):
        np.copyto(sink, prob.numpy().reshape(-1))
        touched[0] += 1

    def write_rows(class_probs):
        for prob in class_probs:
            write_row(prob)
    predict_ms, wait_ms, hand_ms, futures, t_start, done = [], [], [], [], None, 0
    ahead_of = getattr(model, "_engine", None) if hasattr(getattr(model, "_engine", None), "timing") and hasattr(model, "gru") else None
    started_ahead = [0]
    while True:
        t0 = time.perf_counter()
        item = batches_q.get()
        if item is DONE:
            break
        data, batch = item
        t1 = time.perf_counter()
        class_probs = model.predict_on_batch(batch)
        t2 = time.perf_counter()
        if ahead_of is not None and ahead_of.timing()["host_streamed"] & 8:      # this batch's forward had been enqueued ahead of its call
            started_ahead[0] += done >= warm
        if per_sample_submit:
            for sample, prob in zip(data, class_probs):
                futures.append(writer.submit(write_row, prob))
            del prob
        else:
            futures.append(writer.submit(write_rows, class_probs))
        del class_probs, batch, item
        done += 1
        wait_ms.append(1e3 * (t1 - t0)); predict_ms.append(1e3 * (t2 - t1)); hand_ms.append(1e3 * (time.perf_counter() - t2))
        if done == warm:
            for f in futures:
                f.result()
            futures = []
            t_start = time.perf_counter()
    for f in futures:
        f.result()
    elapsed = time.perf_counter() - t_start
    writer.shutdown()
    for t in threads:
        t.join()
    assert touched[0] == total
    timed = n_batches - warm
    cols = timed * batch_size * windows[0].shape[0]
    # the main thread's own cycle (wait for a batch + predict + hand the rows over): what a long run converges to -- `value`
    # also pays the writer's last batch after the loop, 1/12 of it here, 1/5000 of a genome's
    cycle = statistics.median(w + p + h for w, p, h in zip(wait_ms[warm:], predict_ms[warm:], hand_ms[warm:]))
    return {"value": cols / elapsed, "unit": "pileup columns/s", "ms_per_batch": 1e3 * elapsed / timed,
            "timed_batches": timed, "warmup_batches": warm,
            "main_thread_cycle_ms_median": cycle, "steady_state_value": batch_size * windows[0].shape[0] / (1e-3 * cycle),
            "collate_ms_median": statistics.median(collate_ms[warm:]),
            "predict_ms_median": statistics.median(predict_ms[warm:]),
            "main_thread_wait_for_batch_ms_median": statistics.median(wait_ms[warm:]),
            "main_thread_hand_to_writer_ms_median": statistics.median(hand_ms[warm:]),
            "writer_submits_per_batch": batch_size if per_sample_submit else 1,
            "forwards_started_ahead": started_ahead[0],       # timed batches whose forward the previous call had already enqueued (mdk_gru_forward_pipelined)
            **({"predict_ms_all": [round(v, 3) for v in predict_ms], "wait_ms_all": [round(v, 3) for v in wait_ms],
                "collate_ms_all": [round(v, 3) for v in collate_ms]} if detail else {})}


def loop_windows(T, depth, seed, n_win=64):
    """Overlapping row views of one region's feature array, as `Sample.chunks` makes them."""
    import numpy as np
    from medaka_amd import synth
    step = T - 1000                                  # chunk_len 10000, chunk_ovlp 1000 (medaka.py:266-272)
    base = synth.counts_windows(8, step, depth=depth, seed=seed).reshape(-1, 10)
    region = np.concatenate([base] * (-(-((n_win - 1) * step + T) // base.shape[0])))
    return [region[i * step:i * step + T] for i in range(n_win)]


def loop_report(model, B, T, depth, seed, n_batches, host_to_host_rate=None):
    """`fed_loop` with the reference's collate and with the engine's (medaka_amd.torch_ext.stack_counts)."""
    from medaka_amd import torch_ext
    windows = loop_windows(T, depth, seed)
    out = {"what": "sample workers -> Batcher thread (collate) -> model.predict_on_batch -> one-thread writer that copies "
                   "every row out; thread structure of reference prediction.py:36-60, 225-370",
           "batch_windows": B, "chunk_len": T}
    fast = lambda data: torch_ext.Batch.collate(data)
    for name, fn in (("reference_collate", reference_collate), ("engine_collate", fast)):
        fed_loop(model, windows, B, 14, fn, warm=1)                    # allocator warm-up: the loader runs 8 batches ahead, every one a page-locked block
        out[name] = fed_loop(model, windows, B, n_batches, fn)
        log(f"fed loop, {name}: {out[name]['value'] / 1e6:.1f} M columns/s, collate {out[name]['collate_ms_median']:.2f} ms, "
            f"predict {out[name]['predict_ms_median']:.2f} ms per batch")
    out["engine_collate_per_sample_submit"] = fed_loop(model, windows, B, n_batches, fast, per_sample_submit=True)
    log(f"fed loop, engine collate, one writer submit per sample: {out['engine_collate_per_sample_submit']['value'] / 1e6:.1f} M columns/s")
    out["value"] = out["engine_collate"]["value"]
    out["engine_collate"]["threads"] = torch_ext.COLLATE_THREADS
    if host_to_host_rate:
        out["frac_of_host_to_host"] = out["value"] / host_to_host_rate
        out["reference_collate"]["frac_of_host_to_host"] = out["reference_collate"]["value"] / host_to_host_rate
    return out



MFMA_FLOP = 2 * 16 * 16 * 32        # one v_mfma_f32_16x16x32_f16
SUSTAINED_MFMA_FRACTION = 0.66      # of the nominal issue rate over tens of seconds on this chip (profiles/probes/mfma_burn.hip: 0.64-0.67)


def engine_plan(split, B, T, half):
    """The work-group shape of the pass the timed steps ran, from the ENGINE's own planner (include/medaka_amd.h
    `mdk_pass_plan`, device-free) -- not re-derived here: round 5's line priced half precision on 16-window groups the engine
    no longer picks there."""
    from medaka_amd import engine as _engine, models as _models
    chunks = split["chunks"]
    vwin = chunks * B if chunks > 1 else B
    cols = split["columns"] if chunks > 1 else T
    return _engine.pass_plan(vwin, cols, half=half, gpu_share=_models.gpu_share(), split_chunks=chunks if chunks > 1 else 0,
                             host_checks_range=chunks > 1)


def kernel_table(timing_lists, fused_layers, split, B, T, half, traffic_families=None, plan=None):
    """One entry per hot kernel family of the consensus forward: milliseconds per step (hipEvents on the engine's streams),
    algorithmic FLOP (the network's own MACs on the REAL columns), issued FLOP (every MFMA the kernels execute: split
    products, the hi|lo row padding, the margin columns of a split scan), both as a fraction of the 2.5 PFLOP/s fp16 dense
    peak, and the HBM rate the kernel's algorithmic bytes imply.  The step-level figures use the device-resident total."""
    rec0, rec1, gi, head, total = (statistics.mean(v) if v else 0.0 for v in timing_lists)
    cols = B * T
    vcols = split["chunks"] * B * split["columns"] if split["chunks"] > 1 else cols       # columns the kernels really scan
    plan = plan or engine_plan(split, B, T, half)
    nq = plan["windows_per_group"] // 4                 # windows per lane group: 4-, 8- or (half precision) 16-window work-groups
    wg_cols = vcols / (4.0 * nq) * 2                    # (work-group, step) pairs of one layer: both directions
    prod = 1 if half else 2                             # MFMAs per (k-step, gate): W_hi and W_lo passes (the hi|lo rows ride along)
    proj_prod = 1 if half else 3
    fused1, fused_head = bool(fused_layers & 2), bool(fused_layers & 256)
    final_head = bool(fused_layers & 512)               # the scan's second half writes the probabilities itself: no head kernel
    k = []
    def entry(name, ms, algo_mac, mfma, hbm_bytes, note):
        if ms <= 0:
            return
        algo, issued = 2.0 * algo_mac * cols, mfma * MFMA_FLOP
        k.append({"kernel": name, "ms_per_step": ms, "algorithmic_gflop": algo / 1e9, "issued_gflop": issued / 1e9,
                  "algorithmic_tflops": algo / ms / 1e9, "frac_algorithmic_of_fp16_peak": algo / ms / 1e9 / PEAK_F16_DENSE_TFLOPS,
                  "frac_issued_of_fp16_peak": issued / ms / 1e9 / PEAK_F16_DENSE_TFLOPS,
                  "hbm_algorithmic_gb": hbm_bytes / 1e9, "hbm_gb_per_s": hbm_bytes / ms / 1e6, "note": note})
    entry("k_rec_mfma<XIN> (layer 0: recurrence + fused K=10 projection; k_pack_x inside its span)", rec0, 98304 + 7680,
          wg_cols * 8 * (12 * prod + 3 * prod), vcols * (1024 + 512 / (4 * nq)) + vcols * (40 + 512 / (4 * nq)),
          "8 waves x (24 + 6) MFMAs per work-group and step")
    if fused1:
        entry("k_rec_fused (layer 1: K=256 projection + recurrence" + (" + classifier Linear" if fused_head else "") +
              (" + softmax" if final_head else "") + " in one kernel)", rec1,
              98304 + 196608 + (1280 if fused_head else 0), wg_cols * 8 * (12 * prod + 12 * proj_prod + (1 if fused_head else 0)),
              vcols * (2048 + 1024 + (40 if fused_head else 0)) + (cols * 20 if final_head else 0),
              "per work-group and strip of 8 steps: 8 waves x (288 projection + 8 x 24 recurrence" + (" + 8 head" if fused_head else "") + ") MFMAs; gi never in HBM")
    else:
        entry("k_rec_mfma (layer 1 recurrence)", rec1, 98304, wg_cols * 8 * 12 * prod, vcols * 4096, "reads gi (3072 B/column), writes h")
        entry("k_gi_gemm (layer 1 projection)", gi, 196608, vcols / 64.0 * 8 * 96 * proj_prod * 2, vcols * 4096, "writes gi as fp32: 3072 B/column")
    if not final_head:
        entry("k_head_combine (bias + softmax of the partial logits)" if fused_head else "k_head_tiled (Linear + softmax)", head,
              0 if fused_head else 1280, 0, (vcols * 40 + cols * 20) if fused_head else (vcols * 1024 + cols * 20), "HBM streaming")
    issued_total = sum(e["issued_gflop"] for e in k)
    step = {"device_total_ms": total, "algorithmic_gflop": FLOP_PER_COLUMN * cols / 1e9, "issued_gflop": issued_total,
            "frac_algorithmic_of_fp16_peak": FLOP_PER_COLUMN * cols / total / 1e9 / PEAK_F16_DENSE_TFLOPS if total else None,
            "frac_issued_of_fp16_peak": issued_total / total / PEAK_F16_DENSE_TFLOPS if total else None,
            "frac_issued_of_sustained_rate": issued_total / total / (PEAK_F16_DENSE_TFLOPS * SUSTAINED_MFMA_FRACTION) if total else None,
            "sustained_rate_note": f"a register-only MFMA loop on every SIMD sustains {SUSTAINED_MFMA_FRACTION:.2f} of the nominal rate over tens of "
                                   "seconds on this chip (power management; profiles/probes/mfma_burn.hip, profiles/r3_experiments/README.md)",
            "virtual_columns_scanned": vcols, "real_columns": cols, "windows_per_work_group": 4 * nq,
            "work_groups_per_direction": plan["work_groups"]}
    if traffic_families:
        for e in k:
            for fam, rec in traffic_families.items():
                if fam.split("<")[0] in e["kernel"] and rec.get("hbm_bytes_per_step") and ("XIN=1" in fam) == ("XIN" in e["kernel"]):
                    e["hbm_measured_gb"] = rec["hbm_bytes_per_step"] / 1e9
    return k, step

def pmc_summary_r4(name="r6_pmc_step.csv"):
    """Matrix-pipe busy share and HBM rate per kernel family from the committed counter summary of THIS build at 200 x 10000
    (profiles/r5_pmc_step.csv: rocprofv3 --pmc passes of `bench.py --device-only --steps 1`, summed over the step by
    profiles/pmc_step.py).  SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs."""
    import csv
    path = os.path.join(ROOT, "profiles", name)
    try:
        acc = {}
        for row in csv.DictReader(open(path)):
            acc.setdefault(row["kernel_family"], {})[row["counter"]] = float(row["sum_over_step"])
        out = {"source": f"profiles/{name}"}
        for fam, c in acc.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
                out[fam] = {"mfma_busy_of_chip": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)}
        return out
    except Exception as exc:
        return {"error": str(exc)}


RL_CONV2_FLOP = 2 * 128 * 128 * 17      # per (window, read, position): Conv1d(128 -> 128, k = 17), the bulk of k_rl_front


def pmc_summary(n_cus_in_use, rec_avg_ms, traffic_bytes, name="r3_pmc_step.csv"):
    """What north_star asks beside the roofline fraction: matrix-pipe busy share and HBM GB/s of the dominant kernel,
    from the committed counter summary of THIS build at 200 x 10000 (profiles/r3_pmc_step.csv: rocprofv3 --pmc passes of
    `bench.py --device-only --steps 1`, summed over the step by profiles/pmc_step.py) and this run's launch time.
    SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs."""
    import csv
    path = os.path.join(ROOT, "profiles", name)
    try:
        busy = active = 0.0
        for row in csv.DictReader(open(path)):
            fam = row["kernel_family"]
            if fam.startswith("k_rec_mfma") and "fallback" not in fam:
                if row["counter"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    busy += float(row["sum_over_step"])
                elif row["counter"] == "GRBM_GUI_ACTIVE":
                    active += float(row["sum_over_step"]) / 8.0
        if not busy or not active:
            return None
        out = {"source": f"profiles/{name} (B=200, T=10000, " + ("split scan, 1000 virtual windows" if name == "r3_pmc_step.csv" else "sequential scan") + ")",
               "mfma_busy_pct_of_chip": 100.0 * busy / (active * 1024),
               "mfma_busy_pct_on_cus_in_use": 100.0 * busy / (active * 4 * n_cus_in_use), "cus_in_use": n_cus_in_use}
        if traffic_bytes:
            out["hbm_gbps"] = traffic_bytes / (rec_avg_ms * 1e-3) / 1e9
            out["hbm_frac_of_8tbps"] = out["hbm_gbps"] / 8000.0
        return out
    except Exception:
        return None


def rl_traffic(model, B, P, D):
    """HBM bytes of one k_rl_front launch from the committed PMC summary (profiles/traffic_rl.json: FETCH_SIZE x 2 +
    WRITE_SIZE, collected by profiles/collect_round3.sh at 100 x 10000 x 50), scaled by the read positions of this run."""
    path = os.path.join(ROOT, "profiles", "traffic_rl.json")
    try:
        t = json.load(open(path))[model]
        return t["k_rl_front_bytes_per_launch"] * (float(B) * P * D) / t["read_positions"]
    except Exception:
        return None


def rl_kernel_entries(wide, B, P, D, front_ms, total_ms, half, front_tflops):
    """The read-level forward per kernel family: the front end (hipEvents of its own) and what follows it -- the LSTM stack
    with its projections and the head (device total minus the front end; the per-kernel split of that remainder is in the
    committed trace, profiles/r5_rl384_kernel_stats.csv / r5_rl128_kernel_stats.csv)."""
    rest = max(total_ms - front_ms, 1e-6)
    cols = float(B) * P
    if wide:      # lstm 384, four uni-directional layers (reverse, forward, reverse, forward), layer 0 fed by the 128 CNN channels
        mac = (128 + 384) * 1536 + 3 * (384 + 384) * 1536
        layers, what = 4, ("4 x k_lstm_wide (LSTM 384 on clusters of 12 CUs, h exchanged through L2; B / 8 clusters: 13 x 12 = 156 CUs at "
                           "batch 100) with k_gemm_rows (next layer's projection) beside it on a side stream, then k_linear_softmax")
    else:         # lstm 128, two bi-directional layers
        mac = 2 * (128 + 128) * 512 + 2 * (256 + 128) * 512
        layers, what = 2, "per layer k_gi_gemm<NG = 4> (projection) + k_rec_mfma<CELL = 1> (LSTM 128 recurrence, both directions), then k_head_tiled"
    issue = 1 if half else 4
    tf = 2.0 * mac * cols / (rest * 1e-3) / 1e12
    return [
        {"kernel": "k_rl_front", "ms_per_step": front_ms, "algorithmic_tflops": front_tflops,
         "frac_algorithmic_of_fp16_peak": front_tflops / PEAK_F16_DENSE_TFLOPS,
         "frac_issued_of_fp16_peak": front_tflops * (1 if half else 3) / PEAK_F16_DENSE_TFLOPS},
        {"kernel": "LSTM stack + head: " + what, "ms_per_step": rest, "layers": layers,
         "us_per_scan_step_and_layer": 1e3 * rest / (layers * P),
         "algorithmic_mac_per_column": mac, "algorithmic_tflops": tf, "frac_algorithmic_of_fp16_peak": tf / PEAK_F16_DENSE_TFLOPS,
         "frac_issued_of_fp16_peak": tf * issue / PEAK_F16_DENSE_TFLOPS,
         "bound": "the latency of " + str(layers * P) + " dependent steps (a step cannot be shortened and a split scan does not help here: "
                  "profiles/r4_experiments/README.md), not the matrix pipe"},
    ]


def main_rl(args, print_line=True):
    """BASELINE config 4b: the read-level model (reference LatentSpaceLSTM) over uint8 read matrices, one GPU
    per rank, input resident in HBM.  rl384 = the bundled rl_lstm384 architecture (lstm 384, 4 x uni-directional,
    dwells), weights from a seed (20 MB, not committed); rl128 = class defaults, committed trained-like weights."""
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.build()
    from medaka_amd import dist, models, synth
    ranks = dist.Ranks()
    if ranks.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ranks.world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback of the engine")
    dev = torch.device("cuda", ranks.local_rank if ranks.local_rank < torch.cuda.device_count() else 0)
    torch.cuda.set_device(dev)
    wide = args.model == "rl384"
    B = args.batch if args.batch != 200 else 100            # reference CLI default batch for these models
    P, D = args.chunk_len, args.rl_depth
    if wide:
        kw = dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)
        state = synth.synth_rl_state(seed=21, **kw)
    else:
        kw = dict()
        state = dict(np.load(os.path.join(ROOT, "tests", "golden", "rl_weights_trained.npz")))
    m = models.LatentSpaceLSTM(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
    m = m.to(dev).eval()
    if args.half:
        m.half()
    x_small = synth.synth_reads(min(B, 8), P, D, use_dwells=wide, seed=1 + ranks.rank, empty_tail=False)
    x = torch.from_numpy(x_small).repeat((B + x_small.shape[0] - 1) // x_small.shape[0], 1, 1, 1)[:B].contiguous().to(dev)
    eng = m.engine()
    eng.enable_timing(True)
    front, total = [], []
    holder = {}

    def step():
        with torch.inference_mode():
            holder["y"] = m(x)
        t = eng.timing()
        front.append(t["front_ms"]); total.append(t["total_ms"])
    elapsed, _ = dist.timed_steps(ranks, step, lambda: torch.cuda.synchronize(dev), steps=args.steps, warmup=args.warmup)
    front, total = front[-args.steps:], total[-args.steps:]
    cols = B * P
    value = ranks.world * cols * args.steps / elapsed
    issue = 1 if args.half else 3           # fp32 parity: hi*hi + lo*hi + hi*lo products on the fp16 pipe
    flop = RL_CONV2_FLOP * float(B) * P * D
    achieved = flop / (statistics.mean(front) * 1e-3) / 1e12
    result = {
        "metric": "pileup columns/sec (read-level CNN + LSTM inference)", "value": value, "unit": "pileup columns/s",
        "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp16 operands, fp32 accumulate)" if args.half else
                 "f32 (fp16 hi+lo split operands on the fp16 MFMA pipe, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"LatentSpaceLSTM({'lstm 384, 4 x uni-directional, dwells: rl_lstm384 architecture' if wide else 'lstm 128, bi-directional'}), "
                               f"batch {B} windows x {P} positions x {D} reads (uint8 read matrix), input resident in HBM",
                   "batch_windows": B, "chunk_len": P, "reads_per_window": D,
                   "read_positions_per_s": value * D,
                   "weights": "synth.synth_rl_state(seed=21)" if wide else "tests/golden/rl_weights_trained.npz",
                   "parallelism": f"{ranks.world} independent replicas, window-sharded, no collective"},
        "roofline": {"kernel": "k_rl_front (embedding + conv1 + BN + conv17 as an implicit GEMM on MFMA + BN + masked mean over reads)",
                     "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_DENSE_TFLOPS / issue, "unit": "TFLOP/s",
                     "frac": achieved / (PEAK_F16_DENSE_TFLOPS / issue), "traffic": rl_traffic(args.model, B, P, D),
                     "avg_launch_ms": statistics.mean(front), "launches_timed": len(front),
                     "algorithmic_flop_per_launch": flop,
                     "note": f"algorithmic = conv2 only, {RL_CONV2_FLOP} FLOP per (window, read, position); peak = fp16 dense "
                             f"{PEAK_F16_DENSE_TFLOPS:.0f} TFLOP/s / {issue} fp16 products issued per algorithmic MAC; a register-only MFMA loop "
                             "on all SIMDs sustains 0.81 (4 s) to 0.64-0.67 (25 s) of the nominal rate on this chip (profiles/probes/mfma_burn*.hip), "
                             "the kernel's matrix pipe is 81 % busy at a power-limited 1.69 GHz (profiles/r3_experiments/front/)",
                     "kernel_ms_per_step": {"front": statistics.mean(front), "device_total": statistics.mean(total)},
                     "kernels": rl_kernel_entries(wide, B, P, D, statistics.mean(front), statistics.mean(total), args.half, achieved),
                     "wide_retries": eng.timing()["wide_retries"]},
    }
    # host tensor in -> host tensor out, as the prediction loop calls it
    from medaka_amd.torch_ext import Batch
    xb = Batch(read_level_features=x.cpu())
    h2h = []
    for _ in range(4):
        t0 = time.perf_counter()
        m.predict_on_batch(xb)
        h2h.append(time.perf_counter() - t0)
    h_med = ranks.max_over_ranks(statistics.median(h2h[1:]))
    result["host_to_host"] = {"value": ranks.world * cols / h_med, "unit": "pileup columns/s", "ms_per_batch_median": 1e3 * h_med,
                              "timed_batches": 3, "frac_of_device_resident": (cols / h_med) / (value / ranks.world),
                              "what": f"model.predict_on_batch(Batch(read_level_features=<uint8 CPU tensor, {x.numel() / 1e6:.0f} MB>)) -> CPU tensor"}
    result["metric_8d"] = {"value": result["host_to_host"]["value"], "unit": "pileup columns/s", "what": "SURVEY.md 8d: host tensor in -> host tensor out"}
    if ranks.rank == 0:
        if args.cpu_budget > 0 and ranks.world == 1:
            # CPU baseline: the functional PyTorch-CPU restatement of LatentSpaceLSTM.forward (oracle/rl_oracle.py, pinned
            # to the unmodified reference) on a bounded sample: 8 windows x 2000 positions x D reads, 1 warm-up + median of 3
            from oracle import rl_oracle
            cores = usable_cores()
            torch.set_num_threads(cores)
            xs = x_small[:8, :min(P, 2000)]
            rl_oracle.rl_forward(xs[:1, :200], state, use_dwells=wide, bidirectional=not wide)
            times, ref = [], None
            for _ in range(3):
                t0 = time.perf_counter()
                ref = rl_oracle.rl_forward(xs, state, use_dwells=wide, bidirectional=not wide)
                times.append(time.perf_counter() - t0)
                if sum(times) > args.cpu_budget:
                    break
            dt = statistics.median(times)
            log(f"cpu baseline (read-level oracle): {xs.shape[0] * xs.shape[1] / dt:,.0f} columns/s, {len(times)} passes of {dt:.2f} s on {cores} threads")
            with torch.inference_mode():
                out = m(torch.from_numpy(np.ascontiguousarray(xs)).to(dev)).cpu().numpy()
            result["cpu_baseline"] = {"value": xs.shape[0] * xs.shape[1] / dt, "unit": "pileup columns/s",
                                      "cores": cores, "kind": "port", "passes": len(times), "median_s": dt,
                                      "sample": f"{xs.shape[0]} windows x {xs.shape[1]} positions x {D} reads (a {B * P // (xs.shape[0] * xs.shape[1])}x smaller "
                                                f"sample of the {B} x {P} x {D} workload: every window and position costs the same on the CPU path, "
                                                f"so the rate carries over), oracle/rl_oracle.py (functional PyTorch-CPU fp32), median of {len(times)} after a warm-up"}
            result["parity"] = {"max_abs_dp": float(np.abs(out - ref).max()),
                                "argmax_identical": bool((out.argmax(-1) == ref.argmax(-1)).all()),
                                "columns_checked": int(xs.shape[0] * xs.shape[1])}
        if print_line:
            result["summary"] = {"unit": "M columns/s (ms)", "value": round(value / 1e6, 3), "ms_per_step": round(result["ms_per_step"], 2),
                                 "metric_8d": round(result["host_to_host"]["value"] / 1e6, 3),
                                 "parity_max_abs_dp": (result.get("parity") or {}).get("max_abs_dp"),
                                 "cpu_baseline": (result.get("cpu_baseline") or {}).get("value")}
            emit(result, args.full_out)
    ranks.close()
    return result if ranks.rank == 0 else None


def summary_of(result):
    """The figures of the line in < 1500 characters, for readers that keep only its tail (every value is also further up,
    with its definition): rates in M columns/s, times in ms."""
    def g(d, *path, scale=1.0, nd=1):
        for k in path:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return round(d * scale, nd) if isinstance(d, (int, float)) and not isinstance(d, bool) else d
    M = 1e-6
    sp = result.get("scan_split") or {}
    out = {"unit": "M columns/s (ms)",
           "device_resident": g(result, "value", scale=M), "ms_per_step": g(result, "ms_per_step", nd=3),
           "metric_8d": g(result, "metric_8d", "value", scale=M), "metric_8d_ms": g(result, "metric_8d", "ms_per_batch_median", nd=2),
           "first_calls_ms": g(result, "host_to_host", "first_calls_ms"),
           "fed_loop": g(result, "fed_loop", "value", scale=M), "pcie_diet": g(result, "pcie_diet_columns_per_s", scale=M),
           "fed_loop_predict_ms": g(result, "fed_loop", "engine_collate", "predict_ms_median", nd=2),
           "fed_loop_started_ahead": g(result, "fed_loop", "engine_collate", "forwards_started_ahead"),
           "sequential_scan": g(result, "sequential_scan", "value", scale=M),
           "sequential_fed_loop": g(result, "sequential_fed_loop", "value", scale=M),
           "scan_split": {k: (float(f"{sp[k]:.3g}") if isinstance(sp.get(k), float) else sp.get(k)) for k in ("chunks", "margin", "status", "max_delta")},
           "roofline_frac": g(result, "roofline", "frac", nd=4), "roofline_frac_issued": g(result, "roofline", "frac_issued", nd=4),
           "parity_max_abs_dp": g(result, "parity", "max_abs_dp", nd=9), "cpu_baseline": g(result, "cpu_baseline", "value", scale=M, nd=4),
           "cpu_cores": g(result, "cpu_baseline", "cores")}
    h = (result.get("extra") or {}).get("half") or {}
    if h:
        out["half"] = {"value": g(h, "value", scale=M), "ms_per_step": g(h, "ms_per_step", nd=3), "metric_8d": g(h, "metric_8d", "value", scale=M),
                       "metric_8d_ms": g(h, "metric_8d", "ms_per_batch_median", nd=2), "first_calls_ms": g(h, "metric_8d", "first_calls_ms"),
                       "fed_loop": g(h, "fed_loop", "value", scale=M), "fed_loop_predict_ms": g(h, "fed_loop", "predict_ms_median", nd=2),
                       "sequential_fed_loop": g(h, "sequential_fed_loop", "value", scale=M),
                       "margin": g(h, "scan_split", "margin"), "max_abs_dp": g(h, "parity", "max_abs_dp", nd=9),
                       "argmax_identical_columns": g(h, "parity", "argmax_identical_columns"), "columns_checked": g(h, "parity", "columns_checked"),
                       "split": g(h, "scan_split", "status"), "roofline_frac": g(h, "roofline", "frac", nd=4),
                       "roofline_frac_issued": g(h, "roofline", "frac_issued", nd=4), "error": h.get("error")}
    r = (result.get("extra") or {}).get("rl384") or {}
    if r:
        out["rl384"] = {"value": g(r, "value", scale=M, nd=3), "ms_per_step": g(r, "ms_per_step", nd=2),
                        "metric_8d": g(r, "host_to_host", "value", scale=M, nd=3), "parity": g(r, "parity", "max_abs_dp", nd=9), "error": r.get("error")}
    return out


LINE_BUDGET = 6144          # bytes of the ONE stdout line (round 5's grew to 20.8 KB and the driver's record came back unparsed)
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline", "summary")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_issued", "traffic", "avg_launch_ms", "launches_timed",
                 "algorithmic_flop_per_launch", "peak_note")
CPU_KEYS = ("value", "unit", "cores", "kind", "passes", "sample")


def _clip(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1].rstrip() + "~"


def _finite(o):
    """NaN / infinity -> None, recursively: the line must survive a strict JSON parser."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def compact_line(result):
    """The ONE line of stdout: exactly the keys the driver keeps, strings clipped, `summary` last, <= LINE_BUDGET bytes by
    construction (tests/test_bench_line_cpu.py).  Everything else the run measured -- `extra`, `fed_loop`, the per-kernel table,
    every `what` / `note` -- goes to the side file (`--full-out`, default bench_full.json beside this script) and to stderr."""
    roof = result.get("roofline") or {}
    cpu = result.get("cpu_baseline") or {}
    cfg = {k: _clip(v, 150 if k == "workload" else 90) for k, v in (result.get("config") or {}).items()}
    line = {k: result.get(k) for k in LINE_KEYS[:10]}
    line["dtype"] = _clip(result.get("dtype"), 80)
    line["data"] = _clip(result.get("data"), 40)
    line["config"] = cfg
    line["roofline"] = {k: _clip(roof.get(k), 100 if k == "kernel" else 80) for k in ROOFLINE_KEYS} if roof else None
    line["cpu_baseline"] = {k: _clip(cpu.get(k), 120) for k in CPU_KEYS} if cpu else None
    line["summary"] = result.get("summary")
    line = _finite(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_BUDGET:            # cannot happen with the caps above unless `summary` outgrew its own: shed it piece by piece
        for k in ("rl384", "half", "first_calls_ms"):
            (line.get("summary") or {}).pop(k, None)
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) <= LINE_BUDGET:
                break
    assert len(text) <= LINE_BUDGET, len(text)
    return text


def emit(result, full_out):
    """stdout: the compact line (and nothing else); the side file and stderr: everything."""
    full = json.dumps(_finite(result), allow_nan=False)
    try:
        with open(full_out, "w") as f:
            f.write(full + "\n")
    except OSError as exc:
        log(f"could not write {full_out}: {exc}")
    print(full, file=sys.stderr, flush=True)
    print(compact_line(result), flush=True)


def settle_margin(eng, step, max_calls=72):
    """Untimed calls until the split scan's margin learner (include/medaka_amd.h "scan_split_adapt") has stopped moving, so that
    the timed region measures the DEFAULT configuration in its steady state -- learner on, no trial left to run: a smaller
    margin is tried after `adapt` quiet certified calls and a rejected trial costs its call a second forward, once.  Settled =
    `adapt` + 2 calls in a row at one margin without a rejection.  Returns the margins the calls were answered at."""
    adapt = int(os.environ.get("MDK_SCAN_SPLIT_ADAPT", "8"))
    seen, same, last = [], 0, None
    for _ in range(max_calls):
        step()
        sp = eng.split()
        if sp["chunks"] <= 1:
            break
        key = (sp["margin"], sp["fallbacks"], sp["status"])
        same = same + 1 if key == last else 0
        last = key
        seen.append(sp["margin"])
        if adapt == 0 or same >= adapt + 2:
            break
    return seen


def half_section(args, ref_probs):
    """What `medaka inference` runs on a GPU BY DEFAULT (reference prediction.py:164-168: model.half() unless
    --full_precision): this same benchmark with --half in a child process of its own (a fresh engine, the same code path as
    the line above; this process is idle meanwhile), reduced to its figures; parity of ITS host-to-host result against the
    fp32 PyTorch-CPU result of this run's cpu_baseline (handed over through a temporary file)."""
    import subprocess
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.abspath(__file__), "--half", "--cpu-budget", "0", "--extra-rl", "0", "--extra-half", "0",
               "--steps", str(max(5, args.steps)), "--warmup", str(max(2, args.warmup)), "--batch", str(args.batch),
               "--chunk-len", str(args.chunk_len), "--depth", str(args.depth), "--loop-batches", str(args.loop_batches)]
        if ref_probs is not None:
            ref = os.path.join(tmp, "ref.npy")
            np.save(ref, ref_probs)
            cmd += ["--parity-ref", ref]
        full = os.path.join(tmp, "half_full.json")
        cmd += ["--full-out", full]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not os.path.exists(full):
            raise RuntimeError(f"child bench --half failed (rc {r.returncode}): {r.stderr[-400:]}")
        h = json.load(open(full))
    roof = h.get("roofline") or {}
    res = {"what": "model.half() -- the reference CLI's GPU default (prediction.py:164-168): `bench.py --half` on the same batch in a child "
                   "process, same definitions as the line above (fp16 operands, fp32 accumulate, one product: k_rec_fused<HP>)",
           "value": h["value"], "unit": h["unit"], "ms_per_step": h["ms_per_step"], "steps": h["steps"], "warmup": h["warmup"], "dtype": h["dtype"],
           "scan_split": {k: h["scan_split"].get(k) for k in ("chunks", "columns", "margin", "status", "max_delta", "fallbacks",
                                                              "first_call_audited", "first_call_audit_max_dp", "margins_seen")},
           "metric_8d": {k: h["host_to_host"].get(k) for k in ("value", "unit", "ms_per_batch_median", "timed_batches", "first_calls_ms",
                                                               "frac_of_device_resident", "one_copy_each_way_ms_per_batch")},
           "sequential_scan": h.get("sequential_scan"), "sequential_fed_loop": h.get("sequential_fed_loop"),
           "fed_loop": ({"value": h["fed_loop"]["value"], **{k: h["fed_loop"]["engine_collate"].get(k) for k in
                                                             ("ms_per_batch", "predict_ms_median", "collate_ms_median", "timed_batches")}}
                        if h.get("fed_loop") else None),
           "pcie_diet_columns_per_s": h.get("pcie_diet_columns_per_s"),
           "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_issued", "avg_launch_ms",
                                                 "launches_timed", "peak_note", "kernels", "step")},
           "parity": h.get("parity")}
    log(f"half precision: {res['value'] / 1e6:.1f} M columns/s device-resident ({res['ms_per_step']:.2f} ms), "
        f"{res['metric_8d']['value'] / 1e6:.1f} M host-to-host, split {res['scan_split']['status']}, parity {(res.get('parity') or {}).get('max_abs_dp')}")
    return res


def main():
    args = parse()
    if args.procs_per_gpu > 1 and "WORLD_SIZE" not in os.environ:
        # convenience: one command for the K-processes-per-GPU measurement of profiles/r3_procs_per_gpu.txt
        import subprocess
        rest, skip = [], False
        for a in sys.argv[1:]:
            if skip:
                skip = False
            elif a in ("--procs-per-gpu", "--gpus"):
                skip = True
            elif not (a.startswith("--procs-per-gpu=") or a.startswith("--gpus=") or a == "--shared-gpu"):
                rest.append(a)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.procs_per_gpu}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + args.procs_per_gpu), os.path.abspath(__file__),
               "--shared-gpu", "--gpus", str(args.procs_per_gpu)] + rest
        if os.environ.get("MDK_BENCH_DRY"):
            print(" ".join(cmd))
            raise SystemExit(0)
        raise SystemExit(subprocess.call(cmd))
    if args.model != "gru":
        return main_rl(args)
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.build()
    from medaka_amd import dist, models, synth

    torch.set_num_threads(usable_cores())       # host-side torch ops (the reference's collate): the cores we really have
    log('start')
    if args.shared_gpu:      # K processes on ONE GPU (medaka_amd.launch --procs-per-gpu K): every engine takes 1/K of the CUs
        os.environ.setdefault("MEDAKA_AMD_PROCS_PER_GPU", os.environ.get("WORLD_SIZE", "1"))
    ranks = dist.Ranks(backend="gloo" if args.shared_gpu else None)
    if ranks.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ranks.world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.dry_ranks:
        t0 = time.perf_counter()
        ranks.barrier()
        seen, slowest = ranks.ranks_seen(), ranks.max_over_ranks(time.perf_counter() - t0)
        if ranks.rank == 0:
            print(json.dumps({"dry_ranks": True, "n_gpus": ranks.world, "ranks_seen": seen, "barrier_backend": ranks.barrier_backend,
                              "fallback_reason": ranks.fallback_reason, "barrier_s_max_over_ranks": slowest,
                              "devices_visible_to_rank0": torch.cuda.device_count() if torch.cuda.is_available() else 0}), flush=True)
        ranks.close()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback of the engine")
    # (ranks that a launcher isolated with HIP_VISIBLE_DEVICES see one device each, index 0)
    dev = torch.device("cuda", 0 if (args.shared_gpu or ranks.local_rank >= torch.cuda.device_count()) else ranks.local_rank)
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
    if args.deferred_store is not None:
        eng.set_option("deferred_store", args.deferred_store)
    if args.scan_split is not None:
        eng.set_option("scan_split", args.scan_split)
    if args.scan_split_margin is not None:
        eng.set_option("scan_split_margin", args.scan_split_margin)
    eng.enable_timing(True)

    out_holder = {}

    def step():
        with torch.inference_mode():
            out_holder["y"] = model.forward(x_dev)

    rec_ms, total_ms, gi_ms, head_ms, rec_l0, rec_l1, fused_flags = [], [], [], [], [], [], [0]

    def step_timed():
        step()
        t = eng.timing()
        rec_ms.extend(t["rec_ms"])
        rec_l0.append(t["rec_ms"][0])
        rec_l1.append(t["rec_ms"][1] if len(t["rec_ms"]) > 1 else 0.0)
        gi_ms.append(sum(t["gi_ms"]))
        head_ms.append(t["head_ms"])
        total_ms.append(t["total_ms"])
        fused_flags[0] = t["fused_layers"]

    log('engine ready, timing')
    if args.device_only:
        # profiling form: the process's dispatches are exactly those of the requested steps -- no audit of the model's
        # first split call (a second, sequential forward), no extra call
        eng.set_option("scan_split_audit", 0)
        first_call = {"audited": False, "audit_max_dp": 0.0}
    else:
        step()                               # the model's first call: a split scan is audited against the sequential one
        torch.cuda.synchronize(dev)
        first_call = eng.split()
    # `value` is the DEFAULT configuration in its steady state (round 5 held the margin learner still for the timed
    # steps): the learner (option "scan_split_adapt": a smaller margin on trial after 8 quiet certified calls; a rejected trial
    # costs its call a second forward, once) is left ON, and run to where it stops moving in untimed calls first
    margins_seen = [first_call.get("margin")] if not args.device_only else []
    if not args.device_only and args.scan_split is None and args.scan_split_margin is None:
        margins_seen += settle_margin(eng, lambda: (step(), torch.cuda.synchronize(dev)))
    elapsed, mine = dist.timed_steps(ranks, step_timed, lambda: torch.cuda.synchronize(dev),
                                     steps=args.steps, warmup=args.warmup)
    log(f'timed region done: {elapsed:.3f}s for {args.steps} steps')
    # keep only the timed steps' kernel records
    n_layers = len(eng.timing()["rec_ms"])
    rec_ms = rec_ms[-args.steps * n_layers:]
    cols_per_step = B * T
    value = ranks.world * cols_per_step * args.steps / elapsed
    split = eng.split()                     # what the timed steps did: chunks per window, certificate
    split["ranks_certified"] = int(ranks.sum_over_ranks(1.0 if split["status"] == "certified" else 0.0))     # (every rank splits its own batch)
    split["first_call_audited"] = first_call["audited"]          # (one extra, untimed call before the warm-up steps)
    split["first_call_audit_max_dp"] = first_call["audit_max_dp"]
    if args.device_only:
        if ranks.rank == 0:
            print(json.dumps({"metric": "pileup columns/sec (consensus bi-GRU inference)", "value": value,
                              "unit": "pileup columns/s", "n_gpus": ranks.world, "steps": args.steps,
                              "ms_per_step": 1e3 * elapsed / args.steps, "device_only": True, "batch_windows": B,
                              "scan_split": split,
                              "rec_ms_per_step": sum(rec_ms) / args.steps,
                              "rec_l0_ms": statistics.mean(rec_l0[-args.steps:]), "rec_l1_ms": statistics.mean(rec_l1[-args.steps:]),
                              "gi_ms_per_step": statistics.mean(gi_ms[-args.steps:]),
                              "head_ms_per_step": statistics.mean(head_ms[-args.steps:])}), flush=True)
        ranks.close()
        return
    # host tensor in -> host tensor out (what run_prediction's loop sees), every rank at once
    eng.enable_timing(False)
    from medaka_amd.torch_ext import Batch
    x_cpu = torch.from_numpy(x_host)
    # the input tensor as the engine's Batch.collate builds it (page-locked; medaka_amd.torch_ext.stack_counts, installed over
    # the reference's collate by integration.install) -- and, for comparison, as the reference's torch.stack leaves it (pageable)
    variants = [("pageable (reference collate)", x_cpu)] if args.pageable_input else \
               [("page-locked (engine collate)", x_cpu.pin_memory()), ("pageable (reference collate)", x_cpu)]
    if args.stream_host is not None:
        eng.set_option("stream_host", args.stream_host)
    # The first calls are recorded as they come (`first_calls_ms`), untimed calls follow until `settle_s` seconds of host-path
    # traffic have passed (at least 8 calls), and the median is taken over the timed calls after those.  Why: device memory
    # handed back to the driver (hipFree) is wiped by the kernel on the DMA engines, in the background, at ~25 GB/s -- and while
    # that runs every strided copy of the host path takes 130 us longer (10.9 instead of 8.0 ms per call).  Until round 5 every
    # audit of the split scan freed 12 GB of gi workspace, so the ~40 calls behind each were slow ("the box's DMA needs 0.3 s to
    # wake up", profiles/r4_experiments/README.md); the audit allocates nothing now (profiles/r5_experiments/README.md section 9)
    # and the first calls are at the settled rate unless something else in the process has just freed device memory.
    h2h_all, first_calls = {}, {}
    settle_s, settle = 0.6, 0
    for vname, xv in variants:
        xb = Batch(counts_matrix=xv)
        h2h = []

        def host_step():
            t0 = time.perf_counter()
            out_holder["p"] = model.predict_on_batch(xb)
            h2h.append(time.perf_counter() - t0)
        log(f'host-to-host batches, input {vname}')
        n_timed = max(5, args.host_reps)
        t_settle = time.perf_counter()
        while len(h2h) < 8 or ranks.max_over_ranks(time.perf_counter() - t_settle) < settle_s:      # (every rank leaves the loop together)
            host_step()
        settle = max(settle, len(h2h))
        dist.timed_steps(ranks, host_step, lambda: None, steps=n_timed, warmup=2)
        h2h_all[vname] = (ranks.max_over_ranks(statistics.median(h2h[-n_timed:])), n_timed)
        first_calls[vname] = [round(1e3 * t, 3) for t in h2h[:6]]
        log(f'host-to-host, input {vname}: median {1e3 * h2h_all[vname][0]:.2f} ms (first calls: ' + " ".join(f"{1e3 * t:.1f}" for t in h2h[:6]) + ')')
    primary = variants[0][0]
    h_med, n_h2h = h2h_all[primary]
    xb = Batch(counts_matrix=variants[0][1])
    eng.set_option("stream_host", 0)
    plain = []
    for _ in range(4):
        t0 = time.perf_counter()
        model.predict_on_batch(xb)
        plain.append(time.perf_counter() - t0)
    eng.set_option("stream_host", 1)

    result = {
        "metric": "pileup columns/sec (consensus bi-GRU inference)",
        "value": value, "unit": "pileup columns/s", "n_gpus": ranks.world, "steps": args.steps,
        "procs_per_gpu": ranks.world if args.shared_gpu else 1,
        "barrier_backend": ranks.barrier_backend, "ranks_seen": ranks.ranks_seen(),
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp16 operands, fp32 accumulate)" if args.half else
                 "f32 (fp16 hi+lo split operands on the fp16 MFMA pipe, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"r1041_e82_400bps_sup-architecture consensus (GRUModel 10->2x biGRU128->5), synthetic {args.depth}x pileup, "
                               f"batch {B} x {T} per GPU, x and probabilities in HBM",
                   "batch_windows": B, "chunk_len": T, "columns_per_step_per_gpu": cols_per_step,
                   "weights": "tests/golden/weights_trained.npz (reference-trained, synthetic data)",
                   "weights_note": "published model archives are git-LFS stubs offline",
                   "parallelism": f"{ranks.world} independent replicas, window-sharded, no collective"
                                  + (" (DRY CHECK: ranks share device 0)" if args.shared_gpu else "")},
        "scan_split": dict(split, what="chunks per window of the timed steps (include/medaka_amd.h \"scan_split\"): the batch ran as "
                           f"{split['chunks'] * B} windows of {split['columns']} columns; every junction certified on the device "
                           "(max_delta = largest |h_warm - h_carried|, threshold 2^-18; 2^-10 in half precision); the model's first call "
                           "was also run as the sequential scan on the device and compared in full (first_call_audit_max_dp); the "
                           "margin is the one the model's learner SETTLED at in untimed calls before the timed region (`margins_seen`; "
                           "option \"scan_split_adapt\" left at its default: the timed steps are the default configuration)"
                           if split["chunks"] > 1 else "sequential scan",
                           margins_seen=margins_seen),
        "host_to_host": {
            "value": ranks.world * cols_per_step / h_med, "unit": "pileup columns/s",
            "ms_per_batch_median": 1e3 * h_med, "timed_batches": n_h2h, "warmup": 2 + settle, "settle_seconds": settle_s,
            "first_calls_ms": first_calls[primary],
            "frac_of_device_resident": (cols_per_step / h_med) / (value / ranks.world),
            "input": primary,
            "other_inputs": {k: {"value": ranks.world * cols_per_step / v[0], "ms_per_batch_median": 1e3 * v[0]}
                             for k, v in h2h_all.items() if k != primary},
            "what": "model.predict_on_batch(Batch(counts_matrix=<CPU tensor>)) -> CPU tensor, per rank, the SAME batch object every "
                    "time (so nothing of it is on the device beforehand), median over the timed batches, max over ranks; a split "
                    "call copies x in once, in front of the forward, and sends the probabilities home in column chunks (2-D DMA copies) "
                    "under the second half of the last layer's scan, which writes them itself (rec_fused.hpp HEAD = 2); "
                    "`first_calls_ms`: the calls straight after the device-resident section as they came (through round 4: 11 ms each "
                    "-- the driver was wiping the 12 GB an audit had just freed, on the DMA engines; the audit frees nothing now: "
                    "profiles/r5_experiments/README.md section 9); the fed loop below hands every "
                    "NEW batch to the device from the Batcher thread, so there the input does not wait for PCIe",
            "one_copy_each_way_ms_per_batch": 1e3 * statistics.median(plain),
        },
    }
    result["metric_8d"] = {"value": result["host_to_host"]["value"], "unit": "pileup columns/s",
                           "ms_per_batch_median": result["host_to_host"]["ms_per_batch_median"],
                           "what": "SURVEY.md section 8d's metric: columns / wall time of predict_on_batch calls, host tensor in -> host tensor out "
                                   "(= host_to_host.value).  `value` above is the device-resident rate: this build's task statement fixes `value` as "
                                   "\"whole-job throughput with inputs already resident in HBM when the timed region starts\" and rules the "
                                   "PCIe-inclusive rate out of it; neither BASELINE.json nor SURVEY.md says that, SURVEY 8d asks for THIS figure"}
    result["value_basis"] = "device-resident (inputs in HBM at the start of the timed region); SURVEY 8d's host-to-host rate is `metric_8d`"

    shared_loop = None
    ref_probs = None
    if args.shared_gpu and ranks.world > 1 and args.loop_batches > 2:
        # K processes per GPU, each running the whole fed loop (loader threads -> engine collate -> predict_on_batch ->
        # writer): the deployment `medaka_amd.launch --procs-per-gpu K` produces; aggregate = sum over the processes
        from medaka_amd import torch_ext
        windows = loop_windows(T, args.depth, 4321 + ranks.rank)
        fast = lambda data: torch_ext.Batch.collate(data)
        fed_loop(model, windows, B, 14, fast, warm=1)
        ranks.barrier()
        mine_loop = fed_loop(model, windows, B, args.loop_batches, fast)
        shared_loop = {"value": ranks.sum_over_ranks(mine_loop["value"]), "unit": "pileup columns/s", "processes": ranks.world,
                       "rank0": mine_loop, "what": "every process runs the fed loop with the engine's collate; sum of their rates"}
    if ranks.rank == 0:
        if shared_loop:
            result["fed_loop_shared"] = shared_loop
        n = args.steps
        tfile = os.path.join(ROOT, "profiles", "traffic.json" if split["chunks"] > 1 else "r3_seq_traffic.json")
        traffic_doc = None
        if os.path.exists(tfile) and B == 200 and T == 10000 and not args.half:
            try:
                traffic_doc = json.load(open(tfile))
            except Exception:
                traffic_doc = None
        kernels, step_level = kernel_table((rec_l0[-n:], rec_l1[-n:], gi_ms[-n:], head_ms[-n:], total_ms[-n:]), fused_flags[0], split, B, T,
                                           args.half, (traffic_doc or {}).get("families"))
        # the dominant kernel: the longest per step
        dom = max(kernels, key=lambda e: e["ms_per_step"])
        fused1 = bool(fused_flags[0] & 2)
        if "k_rec_fused" in dom["kernel"]:
            # MACs the split issues per algorithmic MAC: 3 products in the projection, 4 (the hi|lo row pairs) in the recurrence and head
            mac_p, mac_r = 196608.0, 98304.0 + (1280.0 if fused_flags[0] & 256 else 0.0)
            issue_factor = 1.0 if args.half else (3.0 * mac_p + 4.0 * mac_r) / (mac_p + mac_r)
        else:
            issue_factor = 1.0 if args.half else 4.0
        peak = PEAK_F16_DENSE_TFLOPS / issue_factor
        dom_traffic = None
        if traffic_doc:
            dom_traffic = traffic_doc.get("k_rec_fused_bytes_per_step") if "k_rec_fused" in dom["kernel"] else traffic_doc.get("k_rec_mfma_bytes_per_launch")
        result["roofline"] = {
            "kernel": ("k_rec_fused: layer 1 = K=256 projection + recurrence + Linear + softmax; HEAD=1 + HEAD=2 launch pair"
                       if fused_flags[0] & 512 else dom["kernel"].split(" (")[0] + ": one launch = one layer pass, both directions"),
            "kernel_note": dom["kernel"] + (" -- one layer pass over all (virtual) windows, both directions, every step = TWO launches of the kernel: "
                                            "steps [0, T/2) with HEAD = 1 (partial logits), [T/2, T) with HEAD = 2 (probabilities); "
                                            "`avg_launch_ms` / `algorithmic_flop_per_launch` are those of the pair, each launch has half of both"
                                            if fused_flags[0] & 512 else
                                            " -- one launch = one layer pass over all (virtual) windows, both directions, every step")
                           + ("" if split["chunks"] == 1 else f"; the split scan runs {split['chunks'] * B} windows of {split['columns']} columns, the "
                              "algorithmic FLOP are those of the REAL columns (margins are overhead)"),
            "bound": "mfma", "achieved": dom["algorithmic_tflops"], "peak": peak, "unit": "TFLOP/s",
            "frac": dom["algorithmic_tflops"] / peak, "traffic": dom_traffic,
            "frac_issued": dom["frac_issued_of_fp16_peak"],
            "avg_launch_ms": dom["ms_per_step"], "launches_timed": n,
            "algorithmic_flop_per_launch": dom["algorithmic_gflop"] * 1e9,
            "note": f"peak = fp16 dense MFMA {PEAK_F16_DENSE_TFLOPS:.0f} TFLOP/s / {issue_factor:.2f} fp16 MACs issued per algorithmic MAC (fp32 parity through "
                    "fp16 hi+lo operands: 3 products in the projection, 4 in the recurrence); `frac_issued` counts every MFMA the kernel "
                    f"executes (row padding, margin columns) against the undivided {PEAK_F16_DENSE_TFLOPS:.0f}; a native fp32-MFMA kernel would be capped at "
                    f"{PEAK_F32_MATRIX_TFLOPS} TFLOP/s, of which this launch reaches {dom['algorithmic_tflops'] / PEAK_F32_MATRIX_TFLOPS:.2f}",
            "peak_note": (f"{peak:.0f} TF = 2.5 PF fp16 dense MFMA (hardware peak)" if args.half else
                          f"{peak:.0f} TF = 2.5 PF fp16 dense / {issue_factor:.2f} fp16 MACs per MAC (fp32 parity); not a hw peak"),
            "frac_of_f32_matrix_peak": dom["algorithmic_tflops"] / PEAK_F32_MATRIX_TFLOPS,
            "kernels": kernels, "step": step_level,
            "pmc": pmc_summary_r4() if (B == 200 and T == 10000 and not args.half and split["chunks"] > 1 and fused1) else None,
            "whole_network_tflops": FLOP_PER_COLUMN * cols_per_step / (statistics.mean(total_ms[-n:]) * 1e-3) / 1e12,
            "kernel_launches_per_step": eng.timing()["rec_launches"],
            "kernel_ms_per_step": {"rec": sum(rec_ms) / args.steps, "gi": statistics.mean(gi_ms[-n:]),
                                   "head": statistics.mean(head_ms[-n:]), "device_total": statistics.mean(total_ms[-n:])},
        }
        if args.cpu_budget > 0 and ranks.world == 1:   # the CPU baseline is a single-GPU-run figure
            probs = out_holder["p"].numpy()
            result["cpu_baseline"], result["parity"], ref_probs = cpu_baseline(
                os.path.join(ROOT, "tests", "golden", "weights_trained.npz"), x_host, probs, args.cpu_budget)
            if result["cpu_baseline"]["value"]:
                result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
        if args.parity_ref:
            ref = np.load(args.parity_ref)
            got = out_holder["p"].numpy()[:ref.shape[0]]
            result["parity"] = {"max_abs_dp": float(np.abs(got - ref).max()),
                                "argmax_identical_columns": int((got.argmax(-1) == ref.argmax(-1)).sum()),
                                "argmax_identical": bool((got.argmax(-1) == ref.argmax(-1)).all()),
                                "columns_checked": int(ref.shape[0] * T), "tolerance": 1e-4,
                                "against": "the reference probabilities handed in with --parity-ref (the fp32 line's PyTorch-CPU result)"}
        if args.loop_batches > 2 and ranks.world == 1:
            result["fed_loop"] = loop_report(model, B, T, args.depth, 4321, args.loop_batches,
                                             result["host_to_host"]["value"])
        # the same with the PCIe diet (SURVEY 8f f2 + f3): uint16 counts + uint32 depth in (24 B/column),
        # argmax class + its probability out (5 B/column)
        cnt = np.minimum(np.rint(x_host * 60.0), 65535).astype(np.uint16)
        dep = np.full(x_host.shape[:2], 60, dtype=np.uint32)
        diet = []
        for _ in range(8):       # (the first calls allocate the aux buffers and the page-locked results)
            t0 = time.perf_counter()
            out_holder["diet"] = model.predict_on_counts(cnt, dep, decoded=True)
            diet.append(time.perf_counter() - t0)
        result["pcie_diet_columns_per_s"] = cols_per_step / statistics.median(diet[3:])
    # Sections that re-arm the margin learner (setting "scan_split" does) or make the workspace grow (a larger margin: the old
    # activations are handed back to the driver, which wipes them on the DMA engines -- behind which the host path's strided
    # copies wait) come LAST: in round 5 they sat in front of the host-path sections, and the half line's
    # first hundred host-to-host calls ran at 8.6 ms instead of 5.6.
    if split["chunks"] > 1:
        # the same steps as the plain sequential scan, for the record (not `value`): 3 steps after 1 warm-up
        eng.set_option("scan_split", 0)
        seq_elapsed, _ = dist.timed_steps(ranks, step, lambda: torch.cuda.synchronize(dev), steps=3, warmup=1)
        result["sequential_scan"] = {"value": ranks.world * cols_per_step * 3 / seq_elapsed, "unit": "pileup columns/s",
                                     "ms_per_step": 1e3 * seq_elapsed / 3, "steps": 3}
        if args.loop_batches > 2 and ranks.world == 1:
            # ... and the fed loop on sequential scans: what a model whose certificate is rejected (one that latches state) runs at
            from medaka_amd import torch_ext
            windows = loop_windows(T, args.depth, 4321)
            fast = lambda data: torch_ext.Batch.collate(data)
            fed_loop(model, windows, B, 14, fast, warm=1)
            sl = fed_loop(model, windows, B, max(8, args.loop_batches // 2), fast)
            result["sequential_fed_loop"] = {k: sl[k] for k in ("value", "unit", "ms_per_batch", "timed_batches", "predict_ms_median",
                                                                "main_thread_cycle_ms_median", "steady_state_value")}
            log(f"fed loop on sequential scans: {sl['value'] / 1e6:.1f} M columns/s, predict {sl['predict_ms_median']:.2f} ms per batch")
        eng.set_option("scan_split", args.scan_split if args.scan_split is not None else 1)
        if args.margin_256 and (args.scan_split_margin or 128) < 256:
            # what a model that needs twice the default margin would run at (the margin is the split scan's price)
            eng.set_option("scan_split_adapt", 0)
            eng.set_option("scan_split_margin", 256)
            m_elapsed, _ = dist.timed_steps(ranks, step, lambda: torch.cuda.synchronize(dev), steps=3, warmup=1)
            m_split = eng.split()
            eng.set_option("scan_split_margin", args.scan_split_margin or 128)
            eng.set_option("scan_split_adapt", int(os.environ.get("MDK_SCAN_SPLIT_ADAPT", "8")))
            result["value_at_margin_256"] = {"value": ranks.world * cols_per_step * 3 / m_elapsed, "unit": "pileup columns/s",
                                             "ms_per_step": 1e3 * m_elapsed / 3, "steps": 3, "chunks": m_split["chunks"],
                                             "columns": m_split["columns"], "status": m_split["status"]}
    if ranks.world == 1 and not args.shared_gpu and not args.half and args.extra_half:
        # the precision `medaka inference` selects on a GPU by default, on the same line (reference prediction.py:164-168)
        try:
            result.setdefault("extra", {})["half"] = half_section(args, ref_probs)
        except Exception as exc:                      # the headline line must not depend on it
            result.setdefault("extra", {})["half"] = {"error": f"{type(exc).__name__}: {exc}"}
    if ranks.world == 1 and not args.shared_gpu and args.extra_rl > 0:
        # BASELINE config 4b on the same line: the read-level rl_lstm384 architecture, timed by this same run
        import copy
        a2 = copy.copy(args)
        a2.model, a2.steps, a2.warmup, a2.cpu_budget, a2.half = "rl384", 3, 1, (min(args.cpu_budget, args.extra_rl) if args.cpu_budget > 0 else 0), False
        del model, eng
        torch.cuda.empty_cache()
        try:
            result.setdefault("extra", {})["rl384"] = main_rl(a2, print_line=False)
        except Exception as exc:                      # the headline line must not depend on it
            result.setdefault("extra", {})["rl384"] = {"error": f"{type(exc).__name__}: {exc}"}
    if ranks.rank == 0:
        result["summary"] = summary_of(result)        # LAST key: whoever keeps only the tail of this line keeps the figures
        emit(result, args.full_out)
    ranks.close()


if __name__ == "__main__":
    main()
