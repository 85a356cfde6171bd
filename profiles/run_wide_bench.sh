python profiles/bench_rl.py 100 10000 50 --wide 2>&1 | tail -1
python profiles/bench_rl.py 128 10000 50 --wide 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wide_prof -o wide -- python $GRAFT_REPO_ROOT/profiles/bench_rl.py 100 10000 50 --wide > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/wide_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
