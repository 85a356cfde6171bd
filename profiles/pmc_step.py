"""Per-STEP counter totals per kernel family from rocprofv3 --pmc CSVs (profiles/collect_round2.sh runs
`bench.py --device-only --steps 1 --warmup 0`: every dispatch of the process belongs to exactly one forward).

    python profiles/pmc_step.py <pmc dir> <out.csv> [traffic.json]

Unlike r1's pmc_summarize.py nothing is averaged over unlike dispatches: counters are SUMMED over all
dispatches of a kernel family in the step (the chunked recurrence is 10 + 7 launches of one symbol, the
empty fallback twins have their own symbol and sum to ~0).  traffic.json = HBM bytes per step per family,
FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE,
both reported in KB by rocprofv3, next to the algorithmic bytes of DESIGN.md."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    """Kernel family + the template arguments that matter, from the mangled name."""
    m = re.match(r"_ZN3mdk\d+(k_[a-z_0-9]+?)I(.*?)EEv", name)
    if not m:
        m2 = re.match(r"_ZN3mdk(?:L)?\d+(k_[a-z_0-9]+)", name) or re.search(r"mdk::(k_[a-z_0-9]+)", name)
        return m2.group(1) if m2 else None
    base, targs = m.group(1), m.group(2)
    vals = re.findall(r"L[ib](\d+)E", targs)
    if base == "k_rec_mfma" and len(vals) >= 4:
        pf, nq, xin, hp = vals[:4]
        twin = " fallback-twin" if pf == "4" else ""
        return f"k_rec_mfma<NQ={nq},XIN={xin},HP={hp}>{twin}"
    if base == "k_rec_fused" and len(vals) >= 2:
        return f"k_rec_fused<K={32 * int(vals[0])},HEAD={vals[1]}>"
    return base


def main():
    root, out_csv = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(float))
    n_disp = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            acc[fam][row["Counter_Name"]] += float(row["Counter_Value"])
            n_disp[fam][row["Counter_Name"]] += 1
    lines = ["kernel_family,counter,dispatches_in_step,sum_over_step"]
    for fam in sorted(acc):
        for c in sorted(acc[fam]):
            lines.append(f'"{fam}",{c},{n_disp[fam][c]},{acc[fam][c]:.6g}')
    open(out_csv, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if len(sys.argv) > 3:
        B, T = 200, 10000
        cols = B * T
        vcols = 1000 * 2256          # the split scan's virtual batch at this shape: what the kernels really stream
        algo = {"k_rec_mfma<NQ=2,XIN=1,HP=0>": vcols * (1024 + 256),     # h out 1024 B/col + packed x 1 KB per (8 windows, step, dir)
                "k_rec_fused<K=256,HEAD=1>": vcols * (2048 + 1024 + 40) // 2,  # first half of the scan: both input directions read by both output directions, h out, partial logits out
                "k_rec_fused<K=256,HEAD=2>": vcols * (2048 + 1024 + 40) // 2 + cols * 20,   # second half: the other direction's partial logits in, probabilities out
                "k_rec_fused<K=256,HEAD=0>": vcols * (2048 + 1024),
                "k_rec_mfma<NQ=2,XIN=0,HP=0>": vcols * 4096,             # (unfused: gi 2 x 1536 + h 1024)
                "k_gi_gemm": vcols * (1024 + 3072), "k_head_tiled": vcols * 1024 + cols * 20,
                "k_head_combine": vcols * 40 + cols * 20, "k_pack_x": vcols * (40 + 256)}       # (k_split_gather is an empty launch unless the input leaves fp16 range)
        fams, total = {}, 0.0
        for fam in sorted(acc):
            if "FETCH_SIZE" in acc[fam] or "WRITE_SIZE" in acc[fam]:
                b = (2.0 * acc[fam].get("FETCH_SIZE", 0.0) + acc[fam].get("WRITE_SIZE", 0.0)) * 1024.0
                fams[fam] = {"FETCH_SIZE_KB_sum": acc[fam].get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB_sum": acc[fam].get("WRITE_SIZE", 0.0),
                             "hbm_bytes_per_step": b, "algorithmic_bytes_per_step": algo.get(fam)}
                total += b
        # one entry per layer pass (fused layer 0, layer 1), whatever work-group size the step ran with
        rec = [v["hbm_bytes_per_step"] for k, v in fams.items() if (k.startswith("k_rec_mfma<") or k.startswith("k_rec_fused<")) and "twin" not in k]
        dom = [sum(v["hbm_bytes_per_step"] for k, v in fams.items() if k.startswith("k_rec_fused<"))]      # (one layer pass = its HEAD = 1 and HEAD = 2 launches)
        traffic = {"config": f"B={B} T={T} (BASELINE configs[1]), one device-resident forward at the engine's defaults (split scan: 5 chunks per window, margin 128 -> 1000 virtual windows of 2256 columns; layer 1's projection, the classifier's Linear and the softmax fused into its recurrence kernel: HEAD = 1 for the first half of the scan, HEAD = 2 for the second); rocprofv3 --pmc FETCH_SIZE / "
                             "WRITE_SIZE in separate passes, summed over every dispatch of the step per kernel family; "
                             "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024",
                   "families": fams, "total_hbm_bytes_per_step": total,
                   "algorithmic_bytes_per_step": cols * 4156, "ratio_to_algorithmic": total / (cols * 4156),
                   "k_rec_mfma_bytes_per_launch": (sum(rec) / len(rec)) if rec else None,
                   "k_rec_fused_bytes_per_step": dom[0] if dom else None}
        json.dump(traffic, open(sys.argv[3], "w"), indent=1)
        print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
