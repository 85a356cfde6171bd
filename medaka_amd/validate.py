"""python -m medaka_amd.validate <model.tar.gz | weights.npz> -- what does THIS model do on the engine?

Every published consensus model is a git-LFS stub where this engine was built, so the split scan's margin, the
certificate's behaviour and the rates (device-resident, host to host, fed loop) were measured on weights trained in the build container.  This is the one
command that answers the same questions for a real archive on the first box that has one (reference `options.py:11-14`,
`datastore.py:135-157`):

  * loads the archive through the reference's own `ModelStoreTGZ.load_model` with `integration.install()` active in
    strict mode (an `.npz` state dict -- the goldens of tests/golden -- goes straight into `models.GRUModel`);
  * runs i.i.d. and structured synthetic pileups (`synth.STRUCTURED_KINDS`) at the requested shape in fp32-parity
    and half mode;
  * prints the margin table (forced margins: largest junction difference, certified?), what the learning rule settles
    at, the certificate per input structure, max |dp| and argmax identity against PyTorch-CPU on a sample, and the
    device-resident / host-to-host rates.

`--plan-only` needs no GPU: it loads the model, says whether the engine covers it and how the shape would be split.
The CPU comparison is a CHECKER (the reference model itself when the archive came through medaka, else the three torch
calls of reference gru.py:66-71 on torch.nn modules); nothing here is a fallback of the product path.
"""
import argparse
import json
import sys
import time

import numpy as np

MARGINS = (64, 96, 128, 192, 256, 384, 512)


def _log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_model(path, device):
    """-> (engine-backed or reference model on `device`, reference model on the CPU or None, description)."""
    import torch
    from medaka_amd import integration, models
    if path.endswith(".npz"):
        state = {k: torch.from_numpy(v) for k, v in dict(np.load(path)).items()}
        bidir = any(k.endswith("_reverse") for k in state)
        n_layers = 1 + max(int(k.split("_l")[1].split("_")[0]) for k in state if k.startswith("gru.weight_ih_l"))
        kw = dict(num_features=state["gru.weight_ih_l0"].shape[1], gru_size=state["gru.weight_hh_l0"].shape[1],
                  n_layers=n_layers, bidirectional=bidir)
        model = models.GRUModel(**kw)
        model.load_state_dict(state)
        cpu = _TorchCpuGru(state, **kw)
        return (model.to(device).eval() if device.type == "cuda" else model.eval()), cpu, {"source": "npz state dict", "class": "GRUModel", "kwargs": kw}
    import medaka.datastore as ds          # the reference has to be importable for its own archives (pickled model_from_dict)
    with ds.ModelStoreTGZ(path) as store:
        cpu = store.load_model(device=torch.device("cpu"))
    desc = {"source": "ModelStoreTGZ.load_model", "class": type(cpu).__name__, "kwargs": cpu.to_dict().get("kwargs")}
    if device.type != "cuda":
        return cpu, cpu, desc
    import os
    os.environ["MEDAKA_AMD"] = "strict"
    integration.install()
    try:
        with ds.ModelStoreTGZ(path) as store:
            model = store.load_model(device=device)
    finally:
        integration.uninstall()
    return model, cpu, desc


class _TorchCpuGru:
    """The three torch calls of reference gru.py:66-71 on torch.nn modules, for `.npz` inputs (a checker, CPU only)."""

    def __init__(self, state, num_features, gru_size, n_layers, bidirectional):
        import torch
        self.gru = torch.nn.GRU(num_features, gru_size, num_layers=n_layers, bidirectional=bidirectional, batch_first=True)
        self.linear = torch.nn.Linear(gru_size * (2 if bidirectional else 1), 5)
        self.gru.load_state_dict({k[4:]: v for k, v in state.items() if k.startswith("gru.")})
        self.linear.load_state_dict({k[7:]: v for k, v in state.items() if k.startswith("linear.")})

    def predict_on_batch(self, batch):
        import torch
        with torch.inference_mode():
            return torch.softmax(self.linear(self.gru(batch.counts_matrix.float())[0]), dim=-1)


def describe(model, desc, B, T):
    """Device-free part: is the model inside the engine's envelope, how would (B, T) be split."""
    import ctypes
    from medaka_amd import integration, lib
    out = dict(desc)
    name = desc["class"]
    if name == "GRUModel":
        out["engine_covers_it"] = type(model).__module__.startswith("medaka_amd") or bool(integration._gru_supported(model))
    elif name == "LatentSpaceLSTM":
        out["engine_covers_it"] = type(model).__module__.startswith("medaka_amd") or bool(integration._rl_supported(model))
        out["note"] = "read-level model: no split scan; measure with `bench.py --model rl384` / `--model rl128`"
    else:
        out["engine_covers_it"] = name == "MajorityVoteModel"
    shape = lib.SplitShape()
    lib.check(lib.load().mdk_split_plan(B, T, 1, 1, 128, ctypes.byref(shape)), "mdk_split_plan")
    out["split_plan_at_margin_128"] = {"batch": B, "columns": T, "chunks": shape.chunks, "virtual_columns": shape.columns, "margin": shape.margin}
    if name == "GRUModel" and out["engine_covers_it"]:
        # how the passes of such a call are launched (mdk_pass_plan): the forward proper, and the audit's sequential scan
        from medaka_amd import engine
        kw = desc.get("kwargs") or {}
        arch = dict(num_features=int(kw.get("num_features", 10)), num_layers=int(kw.get("n_layers", 2)),
                    bidirectional=bool(kw.get("bidirectional", True)))
        S = shape.chunks
        try:
            out["pass_plan"] = {
                "forward": engine.pass_plan(S * B if S > 1 else B, shape.columns if S > 1 else T, split_chunks=S if S > 1 else 0,
                                            host_checks_range=S > 1, host_out=True, host_in=S <= 1, **arch),
                "audit_scan": engine.pass_plan(B, T, lean=True, host_checks_range=True, **arch) if S > 1 else None,
            }
            # (api.hip ensure_workspace: two activation buffers of D x 128 floats per column, the fused head's partial logits;
            # gi = 3 x 128 floats per column and direction, only for passes that need it)
            rows = -(-(S * B if S > 1 else B) // 8) * 8 * (shape.columns if S > 1 else T)
            D, L = (2 if arch["bidirectional"] else 1), arch["num_layers"]
            out["pass_plan"]["workspace_GB"] = round(rows * D * (512 * (2 if L > 1 else 1) + (20 if L > 1 else 0)) / 1e9, 2)
            out["pass_plan"]["gi_GB_if_needed"] = round(rows * D * 1536 / 1e9, 2)
        except RuntimeError as exc:            # e.g. more than 16 features: the exact variant only
            out["pass_plan"] = {"error": str(exc)}
    return out


def _rate(fn, sync, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def _fed_loop(model, eng, x, n_batches=24, ahead=4):
    """A loader thread collating `ahead` batches in front of the main thread's predict_on_batch calls."""
    import queue
    import threading
    from medaka_amd.torch_ext import Batch

    class _S:
        def __init__(self, f):
            self.features = f
    rows = [np.ascontiguousarray(r) for r in x]
    q = queue.Queue(maxsize=ahead)

    def loader():
        for _ in range(n_batches + 2):
            q.put(Batch.collate([_S(r) for r in rows]))
    t = threading.Thread(target=loader, daemon=True)
    t.start()
    started, t0 = 0, None
    for i in range(n_batches + 2):
        model.predict_on_batch(q.get())
        if i == 1:
            t0 = time.perf_counter()                 # (two warm-up batches)
        elif i > 1:
            started += bool(eng.timing()["host_streamed"] & 8)
    dt = (time.perf_counter() - t0) / n_batches
    t.join()
    return {"columns_per_s": x.shape[0] * x.shape[1] / dt, "ms_per_batch": 1e3 * dt, "batches": n_batches, "forwards_started_ahead": started}


def run(model, cpu, B, T, depth, half, sample_windows, log=_log):
    """One precision: margin table, learned margin, certificate per input structure, parity sample, rates."""
    import torch
    from medaka_amd import synth
    from medaka_amd.torch_ext import Batch
    dev = model.device()
    if half:
        model.half()
    eng = model.engine()
    res = {"precision": "half (fp16 operands, fp32 accumulate)" if half else "fp32 parity (fp16 hi+lo split)"}
    x = synth.counts_windows(B, T, depth=depth, seed=4242)
    xd = torch.from_numpy(x).to(dev)
    sync = lambda: torch.cuda.synchronize(dev)

    def fwd():
        with torch.inference_mode():
            return model.forward(xd)
    # ---- margin table on i.i.d. pileups: forced margins, forced chunk count (a rejection is answered sequentially, not escalated)
    eng.set_option("scan_split_audit", 0)
    eng.set_option("scan_split", 0)
    seq = fwd().cpu().numpy()
    table = {}
    for g in MARGINS:
        if T < 8 * g:
            continue
        eng.set_option("scan_split_margin", g)
        eng.set_option("scan_split", max(2, min(16, 1024 // B, T // (4 * g))))
        out = fwd().cpu().numpy()
        info = eng.split()
        table[g] = {"status": info["status"], "chunks": info["chunks"], "max_junction_delta": info["max_delta"],
                    "max_dp_vs_sequential": float(np.abs(out - seq).max())}
    res["margin_table_iid"] = table
    need = min((g for g, r in table.items() if r["status"] == "certified"), default=None)
    res["smallest_certified_margin"] = need
    log(f"  margins: " + ", ".join(f"{g}:{r['status'][:4]}({r['max_junction_delta']:.1e})" for g, r in table.items()))
    # ---- product default: auto mode, audits on, the margin learned over a few calls
    eng.set_option("scan_split_margin", 128)
    eng.set_option("scan_split_audit", 1)
    eng.set_option("scan_split", 1)
    eng.set_option("scan_split_adapt", 4)
    seen = []
    for _ in range(16):
        fwd()
        seen.append(eng.split()["margin"])
    learned = eng.split()
    res["learned"] = {"margins_over_16_calls": seen, "settled_at": learned["margin"], "status": learned["status"], "chunks": learned["chunks"],
                      "rejected_certificates": learned["fallbacks"], "audits": learned["audits"], "audit_worst_dp": learned["audit_worst_dp"]}
    dt = _rate(fwd, sync)
    res["device_resident"] = {"columns_per_s": B * T / dt, "ms_per_batch": 1e3 * dt, "scan": learned["status"], "margin": learned["margin"]}
    eng.set_option("scan_split", 0)
    dts = _rate(fwd, sync, steps=3, warmup=1)
    res["sequential_scan"] = {"columns_per_s": B * T / dts, "ms_per_batch": 1e3 * dts}
    eng.set_option("scan_split", 1)
    xb = Batch(counts_matrix=torch.from_numpy(x).pin_memory())
    for _ in range(8):
        model.predict_on_batch(xb)
    dth = _rate(lambda: model.predict_on_batch(xb), lambda: None, steps=7, warmup=0)
    res["host_to_host"] = {"columns_per_s": B * T / dth, "ms_per_batch": 1e3 * dth}
    # ... and the loop the reference runs (prediction.py:44-52, 225-370): a loader thread collates batches ahead (every batch is on
    # its way to the device when it is made), the main thread calls predict_on_batch -- which also starts the NEXT batch's forward
    # before it returns (include/medaka_amd.h mdk_gru_forward_pipelined)
    res["fed_loop"] = _fed_loop(model, eng, x, n_batches=24)
    if half:
        res["learned"]["fp32_parity_probes"] = eng.split().get("probes")        # half precision: margins are used only after an fp32-parity probe
    log(f"  settled at margin {learned['margin']} ({learned['status']}); {B * T / dt / 1e6:.1f} M columns/s device-resident, "
        f"{B * T / dth / 1e6:.1f} M host to host, {res['fed_loop']['columns_per_s'] / 1e6:.1f} M fed loop "
        f"({res['fed_loop']['forwards_started_ahead']} of {res['fed_loop']['batches']} forwards started ahead), sequential scan {B * T / dts / 1e6:.1f} M")
    # ---- per input structure: certificate and parity against PyTorch-CPU on a sample of windows
    kinds = {}
    for kind in ("iid",) + tuple(synth.STRUCTURED_KINDS):
        xk = x if kind == "iid" else synth.structured_windows(kind, B, T, depth=depth, seed=4243)
        out = model.predict_on_batch(Batch(counts_matrix=torch.from_numpy(xk))).numpy()
        info = eng.split()
        n = min(sample_windows, B)
        ref = cpu.predict_on_batch(Batch(counts_matrix=torch.from_numpy(xk[:n]))).float().numpy()
        kinds[kind] = {"status": info["status"], "margin": info["margin"], "max_junction_delta": info["max_delta"],
                       "max_abs_dp_vs_cpu": float(np.abs(out[:n] - ref).max()),
                       "argmax_identical": int((out[:n].argmax(-1) == ref.argmax(-1)).sum()), "columns_checked": int(n * T)}
        log(f"  {kind:12s} {info['status']:9s} margin {info['margin']:3d}  max|dp| vs CPU {kinds[kind]['max_abs_dp_vs_cpu']:.2e}  "
            f"argmax identical {kinds[kind]['argmax_identical']}/{n * T}")
    res["inputs"] = kinds
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("model", help="a medaka model archive (*.tar.gz; needs the reference importable) or an .npz state dict")
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--chunk-len", type=int, default=10000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--sample-windows", type=int, default=8, help="windows per input structure compared with PyTorch-CPU")
    ap.add_argument("--precision", choices=["both", "fp32", "half"], default="both")
    ap.add_argument("--plan-only", action="store_true", help="no GPU: load the model, report engine coverage and the split plan")
    ap.add_argument("--json", default=None, help="also write the report here")
    args = ap.parse_args(argv)
    import torch
    B, T = args.batch, args.chunk_len
    if args.plan_only:
        model, _, desc = load_model(args.model, torch.device("cpu"))
        report = {"model": describe(model, desc, B, T), "plan_only": True,
                  "checks": ["margin table " + str(list(MARGINS)), "learned margin over 16 calls", "rates: device-resident, sequential, host to host, fed loop",
                             "inputs: iid + " + ", ".join(__import__("medaka_amd.synth", fromlist=["x"]).STRUCTURED_KINDS)]}
    else:
        if not torch.cuda.is_available():
            raise SystemExit("medaka_amd.validate needs a HIP device (or --plan-only): the engine has no CPU path")
        dev = torch.device("cuda", 0)
        report = {"shape": {"batch": B, "columns": T, "depth": args.depth}}
        for half in ([False, True] if args.precision == "both" else [args.precision == "half"]):
            model, cpu, desc = load_model(args.model, dev)          # a fresh model per precision: half() rounds the weights
            report.setdefault("model", describe(model, desc, B, T))
            if desc["class"] != "GRUModel" or not type(model).__module__.startswith("medaka_amd"):
                report["note"] = "not a counts model on the engine: nothing to validate here"
                break
            _log(f"[{'half' if half else 'fp32 parity'}] {args.model}: {B} x {T}")
            report["half" if half else "fp32"] = run(model, cpu, B, T, args.depth, half, args.sample_windows)
    text = json.dumps(report, indent=1, default=float)
    print(text)
    if args.json:
        with open(args.json, "w") as fh:
            fh.write(text)
    return report


if __name__ == "__main__":
    main()
