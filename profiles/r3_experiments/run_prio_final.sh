for i in 1 2 3; do python bench.py --device-only --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('alone B=200 ms_per_step', round(r['ms_per_step'],3))"; done
python bench.py --device-only --batch 1000 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('alone B=1000 ms_per_step', round(r['ms_per_step'],3))"
bash profiles/collect_round3.sh r3d procs 2>&1 | grep "B="
timeout 400 python -m pytest tests -x -q -m gpu -k "not read_level and not wide" 2>&1 | tail -1
