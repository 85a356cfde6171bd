cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_rl2 -o pmc -- python $GRAFT_REPO_ROOT/profiles/bench_rl.py 100 4000 50 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_rl3 -o pmc -- python $GRAFT_REPO_ROOT/profiles/bench_rl.py 100 4000 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob, collections
for d in ("pmc_rl2", "pmc_rl3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rl_front" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, len(v), max(v))
PY
rm -rf gpurun_out/pmc_rl2 gpurun_out/pmc_rl3
