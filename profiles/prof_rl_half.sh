cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/hp384 -o rl -- python $GRAFT_REPO_ROOT/profiles/bench_rl.py 100 10000 50 --wide --half > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/hp128 -o rl -- python $GRAFT_REPO_ROOT/profiles/bench_rl.py 100 10000 50 --half > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for d in hp384 hp128; do python profiles/summarize.py $(find gpurun_out/$d -name "*_results.db" | head -1) gpurun_out/$d.csv | cut -c1-110 | head -8; find gpurun_out/$d -name "*.db" -delete; done
