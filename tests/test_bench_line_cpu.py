"""The ONE line bench.py prints must stay parseable by construction (round 5's grew to 20.8 KB and the driver's record came back
with `parsed: null`): size, strict JSON, the driver's keys, and kernel-table sanity from the engine's own planner."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("record", ["r5_bench_default.json", "r4_bench_default.json", "r5_bench_rl384_B100.json"])
def test_compact_line_from_a_full_record(record):
    """Round 5's (unparsed, 20.8 KB) record and two others, squeezed through compact_line."""
    bench = _bench()
    path = os.path.join(ROOT, "profiles", record)
    if not os.path.exists(path):
        pytest.skip(f"{record} not in profiles/")
    full = json.loads(open(path).read().strip().splitlines()[-1])
    full["summary"] = full.get("summary") or bench.summary_of(full)
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_BUDGET < 8192 and "\n" not in text
    line = json.loads(text, parse_constant=lambda c: pytest.fail(f"non-finite constant {c} on the line"))
    assert json.loads(json.dumps(line, allow_nan=False)) == line
    for k in REQUIRED:
        assert k in line, k
    assert list(line)[-1] == "summary"
    assert set(line) <= set(bench.LINE_KEYS)
    assert line["value"] == full["value"] and line["steps"] == full["steps"] and line["n_gpus"] == full["n_gpus"]
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert set(roof) == set(bench.ROOFLINE_KEYS) and len(roof["kernel"]) <= 100
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and len(cpu["sample"]) <= 120
    assert "workload" in line["config"] and "model" not in line["config"]
    assert len(json.dumps(line["summary"])) < 2500


def test_compact_line_survives_non_finite_numbers_and_long_strings():
    bench = _bench()
    full = {"metric": "m", "value": float("nan"), "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": float("inf"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 " * 100, "data": "synthetic",
            "config": {"workload": "w" * 1000, "parallelism": "p" * 1000},
            "roofline": {"kernel": "k" * 5000, "bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5,
                         "kernels": [{"x": "y" * 10000}], "note": "n" * 10000, "peak_note": "p" * 500},
            "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "port", "passes": 1, "sample": "s" * 5000, "table": [1] * 5000},
            "extra": {"half": {"what": "x" * 50000}}, "summary": {"a": 1}}
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_BUDGET
    line = json.loads(text)
    assert line["value"] is None and line["ms_per_step"] is None
    assert "extra" not in line and "kernels" not in line["roofline"] and "table" not in line["cpu_baseline"]


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("B,chunks,columns", [(200, 5, 2256), (100, 10, 1264), (1000, 1, 10000), (200, 1, 10000)])
def test_kernel_table_takes_its_work_group_shape_from_the_engine(half, B, chunks, columns):
    """Round 5's half line priced 16-window work-groups the engine no longer picks at 1000 chunk-windows: issued FLOP came out
    BELOW algorithmic FLOP.  The table now asks `mdk_pass_plan`; issued >= algorithmic for every kernel entry, always."""
    import __graft_entry__ as graft
    graft.build()
    bench = _bench()
    from medaka_amd import engine
    T = 10000
    split = {"chunks": chunks, "columns": columns if chunks > 1 else T, "margin": 128}
    plan = bench.engine_plan(split, B, T, half)
    vwin = chunks * B if chunks > 1 else B
    want = engine.pass_plan(vwin, split["columns"], half=half, split_chunks=chunks if chunks > 1 else 0, host_checks_range=chunks > 1)
    assert plan == want
    fused = (2 | 256 | 512) if plan["fuse_projection"] else 0
    kernels, step = bench.kernel_table(([2.0], [4.0], [0.5], [0.1], [6.5]), fused, split, B, T, half, plan=plan)
    assert step["windows_per_work_group"] == plan["windows_per_group"]
    for e in kernels:
        if e["issued_gflop"]:
            assert e["issued_gflop"] >= e["algorithmic_gflop"], e
            assert e["frac_issued_of_fp16_peak"] >= e["frac_algorithmic_of_fp16_peak"], e
    if half and chunks == 5:
        assert plan["windows_per_group"] == 8 and plan["work_groups"] == 125     # (rocprof: grid 64000 x 2 threads = 125 x 2 work-groups)
