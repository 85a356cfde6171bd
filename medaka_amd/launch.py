"""N-GPU launcher for the consensus inference step (BASELINE configs 3 and 5).

    python -m medaka_amd.launch --gpus 8 calls_to_draft.bam draft.fasta outdir \
        --model r1041_e82_400bps_sup_v5.0.0 --batch_size 200 [-- <extra medaka inference args>]

The reference's own recipe for scaling is one `medaka inference --regions ...` job per batch of regions,
then one `medaka sequence *.hdf` (reference README.md:294-330; `medaka_consensus` itself runs a single
job, scripts/medaka_consensus:185-199).  This launcher is that recipe for one node of MI355X:

  1. contig lengths from `draft.fasta.fai` (or a scan of the FASTA);
  2. `sharding.shard_regions`: the regions `medaka inference` would cut for itself (prediction.py:100-110),
     dealt longest-first to N shards -- so the union of the shards' samples is exactly a single run's;
  3. one child per shard with `HIP_VISIBLE_DEVICES=<its GPU>` (the reference takes device 0 of the visible set,
     prediction.py:136-138; a HIP_VISIBLE_DEVICES list already set for the launcher is honoured: shard GPUs are
     picked from it) and `MEDAKA_AMD=strict` (swaps the engine in at `ModelStoreTGZ.load_model` and makes any
     return to the PyTorch model an error, medaka_amd/integration.py):
     `medaka inference <bam> <outdir>/shard_i.hdf --regions shard_i.bed ...`;
     `--procs-per-gpu K` starts K such children per GPU on K times as many shards: at the reference's batch
     sizes one process keeps 50-100 of 256 CUs busy (latency-bound recurrence), so K = 2-4
     processes share a GPU; each child is told (`MEDAKA_AMD_PROCS_PER_GPU=K`) and sizes its work-groups so that
     all K fit the chip at once.  Counts-matrix models only: the LSTM(384) cluster recurrence of the read-level
     models needs the whole GPU and refuses to be shared;
  4. waits; if one child fails the others are stopped and the launcher exits non-zero;
  5. prints -- with `--sequence` also runs -- `medaka sequence shard_*.hdf draft.fasta consensus.fasta`.

No collective anywhere: windows never exchange state.  The children are ordinary processes, not
torch.distributed ranks; nothing is shared but the page cache of the BAM.
"""
import argparse
import os
import shlex
import signal
import subprocess
import sys
import time

from medaka_amd import sharding


def contig_lengths(draft):
    """[(name, length)] in file order, from the .fai index when present."""
    fai = draft + ".fai"
    if os.path.exists(fai):
        out = []
        for line in open(fai):
            f = line.rstrip("\n").split("\t")
            if len(f) >= 2:
                out.append((f[0], int(f[1])))
        return out
    out, name, n = [], None, 0
    with open(draft) as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    out.append((name, n))
                name, n = line[1:].split()[0], 0
            else:
                n += len(line.strip())
    if name is not None:
        out.append((name, n))
    return out


def parse_region(text, lengths):
    """'name', 'name:a-b', 'name:a', 'name:a-', 'name:-b' as reference `Region.from_string` reads them
    (common.py:670-710; the sequence name may itself contain ':'), clipped to the sequence like
    `get_bam_regions` does.  A string that is a sequence name as a whole is taken as that sequence."""
    name, start, end = text, 0, None
    if text not in lengths and ":" in text:
        head, bounds = text.rsplit(":", 1)
        try:
            if bounds.startswith("-"):
                a, b = 0, int(bounds.replace("-", ""))
            elif "-" not in bounds:
                a, b = int(bounds), None
            elif bounds.endswith("-"):
                a, b = int(bounds[:-1]), None
            else:
                a, b = (int(v) for v in bounds.split("-"))
        except ValueError:
            raise ValueError(f"cannot parse region '{text}'") from None
        name, start, end = head, a, b
    if name not in lengths:
        raise KeyError(f"{name} is not a sequence of the draft")
    end = lengths[name] if end is None else min(end, lengths[name])
    return sharding.Region(name, start, end)


def plan(draft, n_shards, bam_chunk, chunk_ovlp, regions=None, chunk_len=10000):
    """Per-shard region lists.  `regions`: optional subset, strings as `medaka inference --regions` takes them."""
    contigs = contig_lengths(draft)
    if regions:
        lengths = dict(contigs)
        contigs = [parse_region(r, lengths) for r in regions]
    return sharding.shard_regions(contigs, n_shards, bam_chunk=bam_chunk, chunk_ovlp=chunk_ovlp, chunk_len=chunk_len)


def visible_gpus(n_gpus, first_gpu=0):
    """The device identifiers the children get, one per GPU: positions [first_gpu, first_gpu + n_gpus) of the
    launcher's own HIP_VISIBLE_DEVICES list when one is set (a parent restricted to 4,5,6,7 must not send its
    children to 0-3), plain ordinals otherwise."""
    own = [v.strip() for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v.strip()]
    if own:
        if first_gpu + n_gpus > len(own):
            raise ValueError(f"--gpus {n_gpus} from position {first_gpu}, but HIP_VISIBLE_DEVICES lists {len(own)} devices")
        return own[first_gpu:first_gpu + n_gpus]
    return [str(first_gpu + i) for i in range(n_gpus)]


def write_bed(path, regions):
    with open(path, "w") as fh:
        for r in regions:
            fh.write(f"{r.ref_name}\t{r.start}\t{r.end}\n")


def build_commands(args, shards):
    """[(env additions, argv, hdf path)] for the non-empty shards; shard i runs on GPU i mod n_gpus."""
    jobs = []
    base = shlex.split(args.inference_cmd)
    gpus = visible_gpus(args.gpus, args.first_gpu)
    for i, regs in enumerate(shards):
        if not regs:
            continue
        bed = os.path.join(args.outdir, f"shard_{i}.bed")
        hdf = os.path.join(args.outdir, f"shard_{i}.hdf")
        write_bed(bed, regs)
        argv = base + [args.bam, hdf, "--regions", bed, "--batch_size", str(args.batch_size),
                       "--chunk_len", str(args.chunk_len), "--chunk_ovlp", str(args.chunk_ovlp),
                       "--bam_chunk", str(args.bam_chunk), "--threads", str(args.threads),
                       "--bam_workers", str(args.bam_workers)]
        if args.model:
            argv += ["--model", args.model]
        argv += args.extra
        env = {"HIP_VISIBLE_DEVICES": gpus[i % len(gpus)],
               "MEDAKA_AMD": "0" if args.reference_model else ("1" if args.lenient else "strict"),
               "MEDAKA_AMD_PROCS_PER_GPU": str(args.procs_per_gpu), "MEDAKA_AMD_SHARD": str(i),
               "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        if getattr(args, "reproducible", False):
            env["MDK_SCAN_SPLIT"] = "0"          # sequential scans: bits independent of batching, hence of the sharding
        jobs.append((env, argv, hdf))
    return jobs


def run(jobs, poll_s=0.5, log_dir=None):
    """Start every job, wait; on the first failure stop the rest.  Returns the list of return codes."""
    procs = []
    codes = [None] * len(jobs)
    try:
        for k, (env, argv, _) in enumerate(jobs):         # (inside the try: a failed start still stops the earlier children)
            full_env = dict(os.environ, **env)
            full_env.pop("CUDA_VISIBLE_DEVICES", None)       # one selector only: HIP_VISIBLE_DEVICES
            log = open(os.path.join(log_dir, f"shard_{env.get('MEDAKA_AMD_SHARD', k)}.log"), "w") if log_dir else None
            try:
                child = subprocess.Popen(argv, env=full_env, stdout=log, stderr=subprocess.STDOUT if log else None,
                                         start_new_session=True)
            except BaseException:
                if log:
                    log.close()
                raise
            procs.append((child, log))
        while any(c is None for c in codes):
            for k, (p, _) in enumerate(procs):
                if codes[k] is None:
                    codes[k] = p.poll()
            if any(c not in (None, 0) for c in codes):
                break
            time.sleep(poll_s)
    finally:
        for k, (p, log) in enumerate(procs):
            if p.poll() is None:                          # stop exactly the process groups we started
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, signal.SIGKILL)
                    p.wait()
            codes[k] = p.returncode
            if log:
                log.close()
    return codes


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m medaka_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("bam")
    ap.add_argument("draft", help="draft assembly FASTA (its .fai is used when present)")
    ap.add_argument("outdir")
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--first-gpu", type=int, default=0, dest="first_gpu")
    ap.add_argument("--model", default=None)
    ap.add_argument("--regions", nargs="+", default=None, help="restrict to these contigs / regions")
    ap.add_argument("--batch_size", type=int, default=200)
    ap.add_argument("--chunk_len", type=int, default=10000)      # reference defaults: medaka.py:266-272
    ap.add_argument("--chunk_ovlp", type=int, default=1000)
    ap.add_argument("--bam_chunk", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=2)            # README.md:332-336
    ap.add_argument("--bam_workers", type=int, default=2)
    ap.add_argument("--inference-cmd", default="medaka inference", dest="inference_cmd",
                    help="command that runs one shard (tests substitute a stub)")
    ap.add_argument("--sequence-cmd", default="medaka sequence", dest="sequence_cmd")
    ap.add_argument("--sequence", action="store_true", help="also run `medaka sequence` over the shard HDFs")
    ap.add_argument("--procs-per-gpu", type=int, default=1, dest="procs_per_gpu",
                    help="inference processes sharing each GPU (counts-matrix models; 2-4 fill an MI355X at batch 100-200)")
    ap.add_argument("--reference-model", action="store_true", dest="reference_model",
                    help="children keep the reference's own PyTorch model (MEDAKA_AMD=0)")
    ap.add_argument("--lenient", action="store_true",
                    help="MEDAKA_AMD=1 instead of strict: a model outside the engine's envelope runs on the reference "
                         "implementation with a warning instead of stopping the job")
    ap.add_argument("--reproducible", action="store_true",
                    help="children run with MDK_SCAN_SPLIT=0: the sequential scan's probabilities do not depend on how windows "
                         "are batched, so shard HDFs are bit-identical to a single-process run's (at about half the speed)")
    ap.add_argument("--dry-run", action="store_true", dest="dry_run", help="write the BED files, print the commands")
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = []
    if "--" in argv:                      # everything after `--` is passed through to every child
        cut = argv.index("--")
        argv, extra = argv[:cut], argv[cut + 1:]
    args = ap.parse_args(argv)
    args.extra = extra
    if args.gpus < 1 or args.procs_per_gpu < 1 or args.procs_per_gpu > 8:
        ap.error("--gpus >= 1 and 1 <= --procs-per-gpu <= 8")
    if args.procs_per_gpu > 1 and args.model and "rl_lstm384" in args.model and not args.reference_model:
        ap.error("--procs-per-gpu > 1 is for counts-matrix models: the LSTM(384) cluster recurrence of the "
                 f"read-level model '{args.model}' needs all CUs of its GPU (DESIGN.md 4.5)")
    return args


def main(argv=None):
    args = parse(argv)
    os.makedirs(args.outdir, exist_ok=True)
    shards = plan(args.draft, args.gpus * args.procs_per_gpu, args.bam_chunk, args.chunk_ovlp, args.regions,
                  chunk_len=args.chunk_len)
    jobs = build_commands(args, shards)
    for env, cmd, _ in jobs:
        print("HIP_VISIBLE_DEVICES=%s MEDAKA_AMD=%s %s" % (env["HIP_VISIBLE_DEVICES"], env["MEDAKA_AMD"], shlex.join(cmd)))
    seq = shlex.split(args.sequence_cmd) + [h for _, _, h in jobs] + [args.draft, os.path.join(args.outdir, "consensus.fasta")]
    print(shlex.join(seq))
    if args.dry_run:
        return 0
    t0 = time.time()
    codes = run(jobs, log_dir=args.outdir)
    print(f"{len(jobs)} inference jobs finished in {time.time() - t0:.1f}s, return codes {codes}", file=sys.stderr)
    if any(c != 0 for c in codes):
        return 1
    if args.sequence:
        return subprocess.call(seq)
    return 0


if __name__ == "__main__":
    sys.exit(main())
