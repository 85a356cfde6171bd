#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r4c4; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o host -- python $R/profiles/host_trace.py 200 > $O/host_trace.log 2>&1
cd $R
db=$(find $O/kt -name "*_results.db" | head -1)
echo "db=$db"
python - "$db" <<'PY' > $O/schema.txt 2>&1
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for (n,) in db.execute("select name from sqlite_master where type in ('table','view')"):
    print(n, [r[1] for r in db.execute(f"pragma table_info({n})")])
PY
python profiles/timeline.py "$db" $O/timeline.txt > /dev/null 2> $O/timeline.err
find $O/kt -name "*.db" -delete
grep -v "^\[" $O/host_trace.log | tail -12; head -5 $O/timeline.txt; wc -l $O/timeline.txt; tail -3 $O/timeline.err
