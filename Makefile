# Convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly).
PY ?= python

build:            ## compile the HIP engine (gfx950) and the C oracle in-tree
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build   ## oracle vs reference goldens, host logic, ABI symbols, 2-rank gloo harness
	$(PY) -m pytest tests -x -q -m "not gpu" --durations=10

test-gpu: build   ## parity tests through the C ABI (needs an MI355X)
	$(PY) -m pytest tests -x -q -m gpu

bench: build      ## BASELINE configs[1]: 200 x 10000-column windows on one GPU
	$(PY) bench.py

goldens:          ## regenerate tests/golden from the unmodified reference (build container only)
	$(PY) oracle/make_golden.py && $(PY) oracle/make_golden_rl.py

.PHONY: build test-cpu test-gpu bench goldens
