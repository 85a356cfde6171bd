"""Summarise rocprofv3 --pmc CSVs (profiles/pmc_collect.sh) per kernel and counter.
    python profiles/pmc_summarize.py gpurun_out/pmc_r1 [out.csv]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = ["kernel,counter,dispatches,mean_per_dispatch"]
    for k in sorted(acc):
        if "mdk" not in k:
            continue
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append(f'"{k}",{c},{len(v)},{sum(v)/len(v):.6g}')
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
