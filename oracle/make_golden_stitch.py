"""End-to-end goldens: the UNMODIFIED reference from pileup to stitched FASTQ, in memory.

Run in the build container only (needs /root/reference):

    python oracle/make_golden_stitch.py

What runs, all of it reference code under oracle/ref_shim.py:

    medaka.prediction.run_prediction        prediction.py:14-81   (threaded DataLoader, batches of
                                                                   batch_size, short last batch, B=1
                                                                   remainder pass of predict(), :196-209)
      medaka.features.SampleGenerator.samples    features.py:1283-1313 (quarantine of short pileups,
                                                                        Sample.chunks, common.py:429-453)
      medaka.torch_ext.Batch.collate             torch_ext.py:110-166
      GRUModel.predict_on_batch                  models.py:303-313, gru.py:58-72  (PyTorch-CPU fp32)
    medaka.datastore.DataIndex._get_sorted_index datastore.py:452-484 (order in which `medaka sequence`
                                                                       reads the samples back)
    medaka.stitch._stitch_samples               stitch.py:33-83   (trim_samples, decode_consensus)
    medaka.stitch.collapse_neighbours           stitch.py:172-197
    medaka.stitch.write_fastx_segment           stitch.py:15-30   (the un-filled naming of stitch(): 260-275)

Only two things are substituted, because `pysam`/`h5py`/`libmedaka` are not installed here: the BAM
pileup (`feature_encoder.bam_to_sample` returns synthetic pileups, post-processed by the reference's
own `_post_process_pileup`) and the HDF5 file (`medaka.datastore.DataStore` -> an in-memory store).

Output: tests/golden/stitch_cases.npz -- inputs (raw counts, depth, positions, regions, chunking
parameters) and, per case, the FASTQ text, the order and names of the samples written, the batch
shapes the model saw and every trim slice `trim_samples` chose.
"""
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from medaka_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# case -> chunking parameters of `medaka inference` and the synthetic contigs:
#   contig name -> list of pileup pieces (start major, columns, depth, seed); more than one piece =
#   a coverage gap inside the contig (bam_to_sample returns one Sample per covered stretch)
CASES = {
    # BASELINE configs[0] shape: batch_size 10, 10 000-column windows overlapping by 1000
    "cfg1": dict(chunk_len=10000, chunk_ovlp=1000, batch_size=10, bam_chunk=1_000_000, contigs={
        "ctgA": [(0, 102345, 50, 501)],        # 11 stepped windows + 1 right-aligned: batches of 10 and 2
        "ctgB": [(0, 4321, 50, 502)],          # pileup narrower than chunk_len: remainder pass, B=1
    }),
    # small shapes: every stitch branch in a few thousand columns
    "mini": dict(chunk_len=1000, chunk_ovlp=200, batch_size=4, bam_chunk=4000, contigs={
        "m1": [(0, 5300, 40, 511)],            # one region, right-aligned last window
        "m2": [(100, 2500, 40, 512), (2600, 1700, 30, 513)],   # coverage gap -> two FASTQ records
        "m3": [(0, 600, 40, 514)],             # remainder pass
        "m4": [(0, 14000, 60, 515)],           # longer than bam_chunk: predict() cuts overlapping regions
        "m5": [(0, 9000, 60, 516)],            # the same, with the regions' pileups disagreeing on minor columns
    }, jitter={"m5"}),                         # inside the overlap: trim_samples' junction heuristic
}


def make_sources(fe, common, spec):
    """contig -> list of reference `Sample`s built by the reference's own post-processing."""
    sources, raw_store = {}, {}
    for ctg, pieces in spec["contigs"].items():
        sources[ctg] = []
        for k, (start, n_cols, depth, seed) in enumerate(pieces):
            raw = synth.counts_windows(1, n_cols, depth=depth, seed=seed, raw=True)
            pos = np.empty(n_cols, dtype=[("major", int), ("minor", int)])
            pos["major"], pos["minor"] = raw["major"][0] + start, raw["minor"][0]
            region = common.Region(ctg, int(pos["major"][0]), int(pos["major"][-1]) + 1)
            smp = fe._post_process_pileup(raw["counts"][0].astype(np.uint64), pos, region)
            assert smp.features.dtype == np.float32 and smp.features.shape == (n_cols, 10)
            sources[ctg].append(smp)
            raw_store[f"{ctg}/{k}/counts"] = raw["counts"][0]
            raw_store[f"{ctg}/{k}/depth"] = np.asarray(smp.depth).astype(np.uint32)
            raw_store[f"{ctg}/{k}/major"] = pos["major"].astype(np.int64)
            raw_store[f"{ctg}/{k}/minor"] = pos["minor"].astype(np.int64)
    return sources, raw_store


def region_pileups(common, sources, region, jitter=False):
    """What `bam_to_sample(bam, region)` hands over: the covered stretches inside the region.
    `jitter`: a region that does not start the contig loses every third minor column of its first 150
    major positions -- real pileups of neighbouring regions differ like this where the reads used
    differ (common.py:386-390) -- so that the overlap is not 1-to-1 and stitch needs its heuristic."""
    out = []
    for s in sources[region.ref_name]:
        maj = s.positions["major"]
        lo, hi = np.searchsorted(maj, region.start, side="left"), np.searchsorted(maj, region.end, side="left")
        if hi > lo:
            piece = s.slice(slice(lo, hi))
            if jitter and region.start > 0:
                pos = piece.positions
                minors = np.nonzero((pos["minor"] > 0) & (pos["major"] < region.start + 150))[0]
                keep = np.ones(len(pos), dtype=bool)
                keep[minors[::3]] = False
                piece = piece.slice(keep)
            out.append(piece)
    return out


def run_case(name, spec, state, model=None):
    """`model`: the object handed to the unmodified `run_prediction` in place of the reference's own GRUModel (tests pass
    the engine-backed class with a CPU stand-in for the device); default: the reference model on PyTorch-CPU."""
    arch, models, te = ref_shim.reference_modules()
    import medaka.common as common
    import medaka.datastore
    import medaka.features
    import medaka.labels
    import medaka.prediction as prediction
    import medaka.stitch as stitch

    fe = medaka.features.CountsFeatureEncoder()
    sources, raw_store = make_sources(fe, common, spec)

    class PileupEncoder:
        """`feature_encoder` stand-in: only bam_to_sample is called by SampleGenerator."""
        def bam_to_sample(self, bam, region):
            return region_pileups(common, sources, region, region.ref_name in spec.get("jitter", ()))

    class MemStore:
        """In-memory stand-in of DataStore(output, 'a') for run_prediction (datastore.py:263-300)."""
        samples = []          # shared by every instance of one case, in write order
        registry = set()

        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def write_sample(self, sample):
            if sample.name not in MemStore.registry:
                MemStore.registry.add(sample.name)
                # label_probs arrive as torch tensors (prediction.py:47-51) and come back from HDF5 as arrays
                MemStore.samples.append(sample.amend(label_probs=np.array(sample.label_probs.numpy())))

    if model is None:
        model = arch.GRUModel(num_features=10, num_classes=5, gru_size=128).eval()
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    seen_batches = []
    orig_predict = model.predict_on_batch

    def spy(batch):
        seen_batches.append(tuple(batch.counts_matrix.shape))
        return orig_predict(batch)
    model.predict_on_batch = spy

    saved = medaka.datastore.DataStore
    medaka.datastore.DataStore = MemStore
    try:
        # predict(): prediction.py:92-110 (short regions aside, long ones cut by bam_chunk) ...
        bam_regions = [common.Region(c, 0, int(max(s.positions["major"][-1] for s in sources[c])) + 1)
                       for c in spec["contigs"]]
        regions, remainder = [], []
        for region in bam_regions:
            if region.size < spec["chunk_len"]:
                remainder.append(region)
            elif region.size > spec["bam_chunk"]:
                regions.extend(region.split(spec["bam_chunk"], overlap=spec["chunk_ovlp"], fixed_size=False))
            else:
                regions.append(region)
        # ... :176-209 (batched pass, then the remainders one by one without chunking)
        rem = prediction.run_prediction("mem", None, regions, model, PileupEncoder(), spec["chunk_len"],
                                        spec["chunk_ovlp"], batch_size=spec["batch_size"], bam_workers=2)
        remainder.extend(r[0] for r in rem)
        n_first = len(seen_batches)
        if remainder:
            left = prediction.run_prediction("mem", None, remainder, model, PileupEncoder(), spec["chunk_len"],
                                             spec["chunk_ovlp"], batch_size=1, enable_chunking=False)
            assert not left
    finally:
        medaka.datastore.DataStore = saved

    # `medaka sequence`: stitch.py:199-275 serial path without --fillgaps
    class _Idx:
        samples = sorted((s.name, "mem") for s in MemStore.samples)
    index = medaka.datastore.DataIndex._get_sorted_index(_Idx())
    by_name = {s.name: s for s in MemStore.samples}
    ls = medaka.labels.HaploidLabelScheme()
    trims = []
    orig_trim = common.Sample.trim_samples

    def spy_trim(gen, *a, **k):
        for s, last, heur in orig_trim(gen, *a, **k):
            trims.append((s.name, bool(last), bool(heur)))
            yield s, last, heur
    common.Sample.trim_samples = staticmethod(spy_trim)
    junctions = []          # (end_1, start_2, heuristic) of every overlap junction (common.py:344-427)
    orig_ovl = common.Sample.overlap_indices

    def spy_ovl(s1, s2):
        r = orig_ovl(s1, s2)
        junctions.append((-1 if r[0] is None else int(r[0]), -1 if r[1] is None else int(r[1]), int(r[2])))
        return r
    common.Sample.overlap_indices = staticmethod(spy_ovl)
    try:
        pieces = []
        for ctg in sorted(index):
            length = int(max(s.positions["major"][-1] for s in sources[ctg])) + 1
            for region in common.Region(ctg, 0, length).split(int(1e6), overlap=0, fixed_size=False):
                samples = (by_name[d["sample_key"]] for d in index[ctg])
                pieces.extend(stitch._stitch_samples(samples, ls, region, 0))
    finally:
        common.Sample.trim_samples = staticmethod(orig_trim)
        common.Sample.overlap_indices = staticmethod(orig_ovl)
    fh = io.StringIO()
    ref_name, counter = None, 0
    for (rname, start, stop), seq_parts, quals in stitch.collapse_neighbours(iter(pieces)):
        counter = counter + 1 if ref_name == rname else 0
        stitch.write_fastx_segment(fh, ("{}_{} {}-{}".format(rname, counter, start, stop + 1), seq_parts, quals),
                                   qualities=True)
        ref_name = rname
    out = {f"{name}/raw/{k}": v for k, v in raw_store.items()}
    out[f"{name}/fastq"] = np.array(fh.getvalue())
    out[f"{name}/written"] = np.array([s.name for s in MemStore.samples])
    out[f"{name}/batches"] = np.array(seen_batches, dtype=np.int64)
    out[f"{name}/n_batches_first_pass"] = np.array(n_first)
    out[f"{name}/trimmed"] = np.array([t[0] for t in trims])          # names of the trimmed views, in order
    out[f"{name}/trim_last"] = np.array([t[1] for t in trims])
    out[f"{name}/junctions"] = np.array(junctions, dtype=np.int64).reshape(-1, 3)
    # a light pin of the probabilities themselves: per written sample, the sum over columns of p[argmax]
    out[f"{name}/pmax_sum"] = np.array([float(np.asarray(s.label_probs).max(-1).sum(dtype=np.float64))
                                        for s in MemStore.samples])
    print(name, "batches", seen_batches[:3], "...", len(seen_batches), "samples", len(MemStore.samples),
          "trims", len(trims), "heuristic junctions", sum(j[2] for j in junctions), "of", len(junctions), "fastq bytes", len(fh.getvalue()))
    print(fh.getvalue()[:200].replace("\n", " | "))
    MemStore.samples, MemStore.registry = [], set()
    return out


def main():
    torch.set_num_threads(os.cpu_count())
    state = dict(np.load(os.path.join(GOLD, "weights_trained.npz")))
    out = {}
    for name, spec in CASES.items():
        out.update(run_case(name, spec, state))
    path = os.path.join(GOLD, "stitch_cases.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
