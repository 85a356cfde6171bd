"""The launch plan of a pass (api.hip plan_pass, exported device-free as `mdk_pass_plan`): which regime a shape runs in, what
is fused, what streams, and who needs the gi workspace -- DESIGN.md sections 2 and 4.0."""
import pytest

from medaka_amd import engine


def test_latency_regime_at_the_reference_batch_sizes():
    p = engine.pass_plan(200, 10000)
    assert p == {**p, "windows_per_group": 4, "work_groups": 50, "fuse_layer0": True, "fuse_projection": False,
                 "overlap_gemm": True, "needs_gi": True}
    p = engine.pass_plan(100, 10000)          # (13 tiles of 8 windows = 104 slots: 26 groups of 4)
    assert p["windows_per_group"] == 4 and p["work_groups"] == 26 and p["overlap_gemm"] and not p["fuse_projection"]
    # too short to cut into phases: no side-stream GEMM
    assert not engine.pass_plan(200, 1000)["overlap_gemm"]


def test_throughput_regime_fuses_the_projection_and_the_head():
    p = engine.pass_plan(1000, 10000)
    assert p == {**p, "windows_per_group": 8, "work_groups": 125, "fuse_projection": True, "fuse_head": True, "final_head": True,
                 "overlap_gemm": False}
    assert p["needs_gi"]                       # device entry: the out-of-range fallback is armed on the device, and reads gi
    # T not a multiple of the strip: the unfused pair (and gi); of two strips: the scan's second half cannot deliver by itself
    assert not engine.pass_plan(1000, 10001)["fuse_projection"]
    assert engine.pass_plan(1000, 10008)["fuse_projection"] and not engine.pass_plan(1000, 10008)["final_head"]


def test_a_split_call_runs_without_gi_and_streams_its_result():
    base = dict(windows=1000, T=2256, split_chunks=5)
    assert engine.pass_plan(**base)["needs_gi"]
    p = engine.pass_plan(**base, host_checks_range=True, host_out=True)
    assert not p["needs_gi"] and p["stream_out"] and p["final_head"] and not p["stream_in"]
    # a model that has met out-of-range input keeps the device-side fallback, hence gi
    assert engine.pass_plan(**base, host_checks_range=True, out_of_range_seen=True)["needs_gi"]
    # virtual windows too short to be worth cutting the last layer's scan: the result leaves as one copy
    assert not engine.pass_plan(1000, 400, split_chunks=5, host_checks_range=True, host_out=True)["stream_out"]


def test_an_audit_is_planned_without_gi_whatever_the_batch():
    """`lean` (run_forward's audit): the regime without gi at ANY batch size -- an audit that allocates nothing frees nothing
    (freed device memory is wiped on the DMA engines: profiles/r5_experiments/README.md section 9)."""
    for windows in (1, 10, 100, 200, 500, 1000):
        for half in (False, True):
            p = engine.pass_plan(windows, 10000, lean=True, host_checks_range=True, half=half)
            assert p["windows_per_group"] == 8 and p["fuse_projection"] and p["final_head"] and not p["needs_gi"], (windows, half, p)
            assert not p["overlap_gemm"]
    assert engine.pass_plan(200, 10001, lean=True, host_checks_range=True)["needs_gi"]           # cannot run fused: gi (and it is kept)
    assert engine.pass_plan(200, 10000, lean=True, host_checks_range=True, out_of_range_seen=True)["needs_gi"]
    assert engine.pass_plan(200, 10000, lean=True, host_checks_range=True, num_layers=1)["needs_gi"] is False


def test_processes_sharing_the_gpu_plan_for_their_share():
    assert not engine.pass_plan(500, 4096)["fuse_projection"]                 # alone: 500 windows leave CUs idle for the GEMM
    assert engine.pass_plan(500, 4096, gpu_share=2)["fuse_projection"]        # half the chip each: none idle
    assert engine.pass_plan(300, 4096, gpu_share=4)["windows_per_group"] == 8


def test_half_precision_takes_larger_groups_only_when_it_must():
    assert engine.pass_plan(1000, 2256, half=True)["windows_per_group"] == 8       # fits the chip: fused layer 1
    p = engine.pass_plan(2000, 2256, half=True)
    assert p["windows_per_group"] == 16 and not p["fuse_projection"]
    assert engine.pass_plan(2000, 2256)["windows_per_group"] == 8                  # fp32 parity: never 16


def test_host_input_streams_in_only_on_the_sequential_path():
    assert engine.pass_plan(200, 10000, host_in=True, host_out=True) == {
        **engine.pass_plan(200, 10000), "stream_in": True, "stream_out": True}
    assert not engine.pass_plan(200, 1000, host_in=True, host_out=True)["stream_in"]


def test_wide_inputs_and_bad_arguments():
    with pytest.raises(RuntimeError, match="num_features"):
        engine.pass_plan(10, 1000, num_features=32)           # only the exact variant takes more than 16 features
    for kw in (dict(windows=0, T=10), dict(windows=1, T=0), dict(windows=1, T=10, gpu_share=9), dict(windows=1, T=10, split_chunks=99)):
        with pytest.raises(RuntimeError):
            engine.pass_plan(**kw)


def test_the_fused_kernel_is_not_planned_beyond_its_32_bit_offsets():
    """rec_fused.hpp addresses a tile through 32-bit byte offsets t * D * 4096 in a 2 GB buffer window: at T * D * 4096 >= 2^31
    (T >= 262144 bidirectional) loads would return 0 and stores be dropped -- also in an audit's `lean` scan of a few very
    long windows, whose result would then be delivered as "the sequential one" (ADVICE r5).  Such scans take the unfused pair."""
    edge = (1 << 31) // (2 * 4096)                                              # 262144
    for kw in (dict(lean=True, host_checks_range=True), dict()):
        below = engine.pass_plan(8 if kw else 1000, edge - 16, **kw)
        above = engine.pass_plan(8 if kw else 1000, edge, **kw)
        assert below["fuse_projection"] and below["final_head"], (kw, below)
        assert not above["fuse_projection"] and not above["final_head"] and above["needs_gi"], (kw, above)
    # one-directional models have half the column stride
    assert engine.pass_plan(8, edge, bidirectional=False, lean=True, host_checks_range=True)["fuse_projection"]
    assert not engine.pass_plan(8, 2 * edge, bidirectional=False, lean=True, host_checks_range=True)["fuse_projection"]
