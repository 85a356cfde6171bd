"""HBM bytes per k_rl_front launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
`bench.py --model <m> --steps 1 --warmup 0 --cpu-budget 0` (profiles/collect_round3.sh, stage pmc_rl).

    python profiles/traffic_rl.py <pmc dir> <model> <B> <P> <D> <profiles/traffic_rl.json>

FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both in KB;
averaged over the k_rl_front dispatches of the process (the bench makes one full-size launch plus, with the CPU baseline
off, nothing else of that kernel)."""
import csv
import glob
import json
import os
import sys


def main():
    root, model, B, P, D, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    tot, n = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_rl_front" in row["Kernel_Name"] and row["Counter_Name"] in tot:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
    if not n["FETCH_SIZE"] or not n["WRITE_SIZE"]:
        raise SystemExit("no k_rl_front counters found")
    fetch = tot["FETCH_SIZE"] / n["FETCH_SIZE"] * 1024 * 2
    write = tot["WRITE_SIZE"] / n["WRITE_SIZE"] * 1024
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[model] = {"k_rl_front_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                   "dispatches": n["FETCH_SIZE"], "read_positions": float(B) * P * D,
                   "algorithmic_bytes": float(B) * P * D * (5 if model == "rl384" else 4) + float(B) * P * 512,
                   "what": "uint8 read matrix in (4|5 B per read position) + pooled features out (512 B per position)"}
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(data[model]))


if __name__ == "__main__":
    main()
