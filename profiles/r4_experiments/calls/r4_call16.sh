#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c16; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -n 16 $O/pytest_gpu.log; tail -3 $O/smoke.log; cat $O/rc.txt
