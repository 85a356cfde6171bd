"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against
  (a) committed goldens produced by the UNMODIFIED reference (tests/golden/),
  (b) the CPU oracle on the same seeded inputs,
  (c) size-independent properties at BASELINE.json's full batch (200 x 10000).
Tolerance (BASELINE.json north_star): per-position probabilities <= 1e-4 absolute in fp32 and
identical argmax.  We assert 2e-5, five times tighter."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, weight_set
from medaka_amd import engine, integration, lib, models, synth
from medaka_amd.torch_ext import Batch
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def engines(gold):
    cache = {}

    def get(wname, exact=False):
        key = (wname, exact)
        if key not in cache:
            e = engine.GruEngine(weight_set(gold, wname))
            e.set_variant(exact)
            cache[key] = e
        return cache[key]
    yield get
    for e in cache.values():
        e.close()


def _check(out, ref, tol=TOL, what="", strict_argmax=False):
    assert out.shape == ref.shape, what
    assert np.isfinite(out).all(), what
    err = np.abs(out - ref).max() if out.size else 0.0
    assert err <= tol, f"{what}: max|dp| = {err:.3e}"
    if strict_argmax:
        # trained (confident) weight sets: the consensus must be the reference's on EVERY column
        assert np.array_equal(out.argmax(-1), ref.argmax(-1)), what
        return
    # near-uniform outputs (random-init weights): argmax identity wherever the reference itself separates
    # top-2 by more than the tolerance
    srt = np.sort(ref, -1)
    clear = (srt[..., -1] - srt[..., -2]) > 2 * tol
    assert (out.argmax(-1) == ref.argmax(-1))[clear].all(), what


def test_mfma_fragment_layout_selftest():
    err, subnormal_ok = engine.selftest_mfma(0)
    assert err == 0.0          # integer data: the assumed A/B/D lane maps are exact or wrong
    print("fp16 subnormal operands preserved by MFMA:", subnormal_ok)


def test_device_is_gfx950():
    assert "gfx950" in lib.device_name(0)


@pytest.mark.parametrize("exact", [False, True], ids=["mfma", "exact"])
def test_goldens_from_unmodified_reference(gold, engines, exact):
    n = 0
    for key in sorted(gold["gru_outputs"]):
        wname, cname = key.split("/")
        x = gold["gru_inputs"][cname]
        if exact and x.shape[1] > 2000:
            continue
        out = engines(wname, exact).forward_host(x)
        _check(out, gold["gru_outputs"][key], what=f"{key} exact={exact}", strict_argmax=(wname == "trained"))
        n += 1
    assert n >= 20


@pytest.mark.parametrize("name", ["x5", "x1e-3", "range16", "saturated", "bigx"])
def test_adversarial_weight_sets_vs_reference_goldens(gold, name):
    """Weight sets built to hurt the fp16 hi+lo split (oracle/make_golden_adversarial.py): outputs of the
    unmodified reference on a 10 000-column window.  The contract tolerance (1e-4) applies; trained-derived
    sets must also agree on every argmax."""
    from oracle.make_golden_adversarial import adversarial_input, adversarial_state
    adv = np.load(os.path.join(GOLD, "gru_adversarial.npz"))
    e = engine.GruEngine(adversarial_state(name, gold["weights_init"], gold["weights_trained"]))
    out = e.forward_host(adversarial_input(name))
    e.close()
    err = float(np.abs(out - adv[name]).max())
    print(f"adversarial {name}: max|dp| = {err:.2e}")
    _check(out, adv[name], tol=1e-4, what=name, strict_argmax=name in ("range16", "saturated", "bigx"))


def test_consensus_string_identical(gold, engines):
    """argmax -> '*ACGT' with gaps dropped (reference labels.py:1053-1085) on the engine's
    probabilities equals the reference's own decode of its own probabilities."""
    alphabet = np.array(list("*ACGT"))
    for key in ("trained/synth60", "trained/edge_B3", "trained/testcounts"):
        cname = key.split("/")[1]
        out = engines("trained").forward_host(gold["gru_inputs"][cname])
        for w in range(out.shape[0]):
            s = "".join(alphabet[out[w].argmax(-1)]).replace("*", "")
            assert s == str(gold["consensus_decode"][key][w]), key


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (1, 3), (2, 5), (7, 33), (8, 64), (9, 100),
                                 (17, 257), (3, 1001), (25, 40)])
def test_ragged_shapes_vs_oracle(gold, engines, B, T):
    x = synth.counts_windows(B, T, depth=50, seed=1000 + 17 * B + T)
    for wname in ("x3", "trained"):
        ref = oracle.c_gru_forward(x, weight_set(gold, wname))
        _check(engines(wname).forward_host(x), ref, what=f"{wname} B={B} T={T}")


def test_uniform_noise_vs_oracle(gold, engines):
    x = synth.uniform_windows(6, 777, seed=9)     # medaka/test/test_sample.py:50 style input
    for wname in ("init", "x3"):
        ref = oracle.c_gru_forward(x, weight_set(gold, wname))
        _check(engines(wname).forward_host(x), ref, what=wname)


def test_exact_and_mfma_kernels_agree_on_device(gold, engines):
    x = synth.counts_windows(5, 1500, seed=77)
    a = engines("x3", False).forward_host(x)
    b = engines("x3", True).forward_host(x)
    assert np.abs(a - b).max() <= TOL


def test_both_tile_sizes_agree_bitwise(gold):
    """4-window and 8-window recurrence tiles run the same arithmetic per window."""
    x = synth.counts_windows(21, 500, seed=31)
    outs = []
    for tile in (4, 8):
        e = engine.GruEngine(weight_set(gold, "x3"))
        e.set_option("rec_windows_per_tile", tile)
        outs.append(e.forward_host(x))
        e.close()
    assert np.array_equal(outs[0], outs[1])
    _check(outs[1], oracle.c_gru_forward(x, weight_set(gold, "x3")), what="8-window tiles")


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
def test_recurrence_schedule_variants_agree_bitwise(gold, half):
    """Option "deferred_store" (h_t leaves for HBM from inside step t+1) and the tile size only re-time the
    recurrence: every accumulator sees its MFMAs in the same order, so the probabilities must be
    bit-identical, also across a resumed (chunked) layer.  (The round-2 schedule experiments -- split
    synchronisation, z-last, packed LDS writes, four waves -- lost on the hardware and are no longer
    shipped: profiles/r2_experiments/.)"""
    x = synth.counts_windows(13, 2304, seed=77)          # T >= 2048 and % 16 == 0: overlap chunks are used
    e = engine.GruEngine(weight_set(gold, "trained"))
    e.set_precision(half)
    outs = {}
    for tile in (4, 8):
        for ds in (1, 0):
            e.set_option("rec_windows_per_tile", tile)
            e.set_option("deferred_store", ds)
            outs[(tile, ds)] = e.forward_host(x)
    e.set_option("deferred_store", 1)
    e.close()
    base = outs[(4, 1)]
    for k, v in outs.items():
        assert np.array_equal(v, base), k
    if not half:
        _check(base, oracle.c_gru_forward(x, weight_set(gold, "trained")), what="variants", strict_argmax=True)


def test_multi_pass_batches_agree_bitwise(gold):
    """Batches above the workspace column budget run as equal passes (api.hip: mdk_gru_forward_dev);
    windows are independent, so the result is the single-pass result bit for bit."""
    x = synth.counts_windows(37, 300, seed=35)
    e = engine.GruEngine(weight_set(gold, "x3"))
    one = e.forward_host(x)
    for budget in (300 * 16, 300 * 9, 300, 1):     # 3 passes of 16, 5 of 8, 37 of 1, 37 of 1
        e.set_option("max_rows_per_pass", budget)
        assert np.array_equal(e.forward_host(x), one), budget
    e.close()
    _check(one, oracle.c_gru_forward(x, weight_set(gold, "x3")), what="multi-pass")


def test_streamed_host_path_agrees_bitwise(gold):
    """mdk_gru_forward copies x in and the probabilities out in time slabs under the recurrences (api.hip,
    HostIO) when T >= 2048 and T % 16 == 0: same bits as one copy each side ("stream_host" = 0), as the
    device-resident entry, with several passes, with pageable and page-locked buffers, and when the input
    range flag sends layer 0 down its unfused fallback after the slabs were already consumed."""
    x = synth.counts_windows(19, 2304, seed=88)
    e = engine.GruEngine(weight_set(gold, "trained"))
    streamed = e.forward_host(x)
    e.set_option("stream_host", 0)
    plain = e.forward_host(x)
    e.set_option("stream_host", 1)
    assert np.array_equal(streamed, plain)
    e.set_option("max_rows_per_pass", 8 * 2304)                  # 3 passes of 8, 8, 3 windows, each streamed
    assert np.array_equal(e.forward_host(x), plain)
    e.set_option("max_rows_per_pass", 0)
    pin_x, pin_p = engine.PinnedArray(x.shape), engine.PinnedArray(plain.shape)
    pin_x.array[...] = x
    assert np.array_equal(e.forward_host(pin_x.array, out=pin_p.array), plain)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty((19, 2304, 5), dtype=torch.float32, device="cuda")
    e.forward_ptr(xd.data_ptr(), 19, 2304, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), plain)
    big = x * np.float32(3000.0)                                   # beyond the fused projection's fp16 range
    a = e.forward_host(big)
    e.set_option("stream_host", 0)
    assert np.array_equal(e.forward_host(big), a)
    e.close()
    # a view may outlive its PinnedArray: the page-locked block is released with the last view
    import gc
    row = pin_p.array[3]
    pin_x.free(); pin_p.free()
    del pin_x, pin_p
    gc.collect()
    assert np.array_equal(row, plain[3])
    del row
    with pytest.raises(ValueError, match="integers"):
        engine.GruEngine.forward_counts_host(None, np.ones((1, 2, 10), np.float32), np.ones((1, 2), np.uint32))
    _check(plain, oracle.c_gru_forward(x, weight_set(gold, "trained")), what="streamed host path", strict_argmax=True)


def test_host_path_shape_stress(gold):
    """Back-to-back host calls of changing shape and precision through ONE engine (event pool, staging and
    workspace reuse, streamed and plain schedules interleaved): each equals the device-resident entry."""
    rng = np.random.default_rng(2024)
    e = engine.GruEngine(weight_set(gold, "trained"))
    shapes = [(1, 13), (3, 2048), (1, 2064), (17, 2304), (2, 4096), (9, 100), (33, 2048), (1, 1), (5, 3008), (40, 2320)]
    for i in range(24):
        B, T = shapes[int(rng.integers(len(shapes)))]
        half = bool(rng.integers(2))
        e.set_precision(half)
        x = synth.counts_windows(B, T, seed=int(rng.integers(1 << 30)))
        host = e.forward_host(x)
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty((B, T, 5), dtype=torch.float32, device="cuda")
        e.forward_ptr(xd.data_ptr(), B, T, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(host, yd.cpu().numpy()), (i, B, T, half)
    e.close()


def test_overlapped_projection_agrees_bitwise(gold):
    """Layer 0's recurrence in resumable chunks with layer 1's projection GEMM on a side stream
    (api.hip, forward_pass) is the same arithmetic as the plain sequence: identical bits, for the
    fused and the unfused (fallback: raw counts) layer-0 paths."""
    x = synth.counts_windows(11, 2048, seed=61)
    for scale in (1.0, 3000.0):
        xs = x * np.float32(scale)
        e = engine.GruEngine(weight_set(gold, "trained"))
        a = e.forward_host(xs)
        e.set_option("overlap_gemm", 0)
        b = e.forward_host(xs)
        e.close()
        assert np.array_equal(a, b), scale
    _check(a if scale == 1.0 else engine.GruEngine(weight_set(gold, "trained")).forward_host(x),
           oracle.c_gru_forward(x, weight_set(gold, "trained")), what="overlapped")


def test_multi_pass_with_overlap_agrees_bitwise(gold):
    """Passes share the side stream and the second gi buffer of the overlapped projection: results of
    a 3-pass forward equal the single pass bit for bit (stream ordering keeps pass k+1's side-stream
    GEMMs behind pass k's layer-1 recurrence)."""
    x = synth.counts_windows(20, 2048, seed=67)
    e = engine.GruEngine(weight_set(gold, "trained"))
    one = e.forward_host(x)
    e.set_option("max_rows_per_pass", 2048 * 8)
    assert np.array_equal(e.forward_host(x), one)
    e.close()


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
@pytest.mark.parametrize("bidirectional", [True, False], ids=["bi", "uni"])
def test_fused_projection_agrees_bitwise(gold, bidirectional, half):
    """Option "fuse_proj" (rec_fused.hpp): layers >= 1 compute their input projection inside the recurrence kernel,
    strip by strip -- the same MFMAs in the same order on the same operands as k_gi_gemm, the same fmaf for scale and
    bias, the same gate arithmetic -- so the probabilities are BIT-IDENTICAL to the GEMM + recurrence pair, whole
    layers and resumed (chunked) ones alike.  Every parity result of the unfused path therefore carries over."""
    if bidirectional:
        st, kw = weight_set(gold, "trained"), {}
    else:       # 2 x uni-directional: K = 128 (the KSTEPS = 4 instantiation); weights drawn like PyTorch's default init
        rng = np.random.default_rng(5)
        k = 1.0 / np.sqrt(128)
        st = {}
        for layer, kin in ((0, 10), (1, 128)):
            st[f"gru.weight_ih_l{layer}"] = rng.uniform(-k, k, (384, kin)).astype(np.float32)
            st[f"gru.weight_hh_l{layer}"] = rng.uniform(-k, k, (384, 128)).astype(np.float32)
            st[f"gru.bias_ih_l{layer}"] = rng.uniform(-k, k, 384).astype(np.float32)
            st[f"gru.bias_hh_l{layer}"] = rng.uniform(-k, k, 384).astype(np.float32)
        st["linear.weight"] = rng.uniform(-k, k, (5, 128)).astype(np.float32) * 8
        st["linear.bias"] = rng.uniform(-k, k, 5).astype(np.float32)
        kw = dict(bidirectional=False)
    e = engine.GruEngine(st, **kw)
    e.set_precision(half)
    e.enable_timing(True)
    e.set_option("rec_windows_per_tile", 8)          # the fused kernel carries 8 windows per work-group
    e.set_option("fuse_head", 0)                      # (the fused classifier head is ~1e-7, not bitwise: checked below)
    n = 0
    for B, T in ((13, 2304), (9, 8), (8, 16), (3, 1000), (40, 264), (17, 4096), (5, 999)):
        x = synth.counts_windows(B, T, depth=40, seed=B * 1000 + T)
        outs = {}
        for fp in (0, 2):
            e.set_option("fuse_proj", fp)
            outs[fp] = e.forward_host(x)             # host entry: with T >= 2048 the last layer runs in resumed pieces
            fused = e.timing()["fused_layers"]
            assert fused == ((2 if T % 8 == 0 else 0) if fp else 0), (B, T, fp, fused)
        n += bool(fused)
        assert np.array_equal(outs[0], outs[2]), (B, T, float(np.abs(outs[0] - outs[2]).max()))
        if bidirectional and not half:
            _check(outs[2], oracle.c_gru_forward(x, st) if B * T < 40000 else outs[0], what=f"fused {B}x{T}")
    assert n >= 5
    # the classifier's Linear inside the last layer's kernel (rec_fused.hpp HEAD) + k_head_combine: fp16x2-split MFMA
    # instead of fp32 FMAs, so ~1e-7 on the probabilities instead of identical bits; whole and resumed layers
    e.set_option("fuse_proj", 2)
    for B, T in ((13, 2304), (9, 8), (5, 16), (21, 272), (3, 1000), (17, 4096)):
        x = synth.counts_windows(B, T, depth=40, seed=B * 1000 + T)
        e.set_option("fuse_head", 0)
        plain = e.forward_host(x)
        e.set_option("fuse_head", 1)
        fused = e.forward_host(x)
        # (bit 9: the scan's second half wrote the probabilities itself -- one-directional: every launch)
        assert e.timing()["fused_layers"] == (2 | 256 | (512 if (T % 16 == 0 or not bidirectional) else 0)), e.timing()
        e.set_option("final_head", 0)                              # ... against k_head_combine: the same bits
        assert np.array_equal(e.forward_host(x), fused) and e.timing()["fused_layers"] == (2 | 256)
        e.set_option("final_head", 1)
        d = float(np.abs(fused - plain).max())
        # (half precision: the head sees the fp16 image of h and fp16 W_lin -- what the reference's own autocast Linear
        # sees -- instead of the fp32 h: rounding of 2^-11 per operand)
        assert d <= (2e-3 if half else 1e-6), (B, T, d)
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty(B, T, 5, device="cuda")
        e.forward_ptr(xd.data_ptr(), B, T, yd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), fused)            # one launch per layer == resumed launches, bit for bit
        if bidirectional and B * T < 40000:
            _check(fused, oracle.c_gru_forward(x, st), tol=2e-3 if half else TOL, what=f"fused head {B}x{T}")
    e.set_option("fuse_head", 0)
    # auto mode: small batches leave CUs idle and keep the GEMM on the side stream; batches that fill the chip fuse
    e.set_option("rec_windows_per_tile", 0)
    e.set_option("fuse_proj", 1)
    e.forward_host(synth.counts_windows(16, 2304, seed=3))
    assert e.timing()["fused_layers"] == 0
    x = synth.counts_windows(960 if bidirectional else 1920, 264, seed=4)     # > 208 recurrence work-groups of 8 windows
    out = e.forward_host(x)
    assert e.timing()["fused_layers"] == 2
    e.set_option("fuse_proj", 0)
    assert np.array_equal(e.forward_host(x), out)
    e.set_option("fuse_proj", 1)
    e.set_option("fuse_head", 1)                      # the product default
    out_h = e.forward_host(x)
    assert e.timing()["fused_layers"] & 0x1ff == (2 | 256) and np.abs(out_h - out).max() <= (2e-3 if half else 1e-6)
    e.close()


def test_three_layer_model_with_fused_layers(gold):
    """`GRUModel(n_layers=3)` (gru.py:13-56 accepts any depth): layers 1 AND 2 run with their projections fused, the last
    one with the classifier head; bitwise against the unfused pairs, and against the C oracle."""
    rng = np.random.default_rng(11)
    k = 1.0 / np.sqrt(128)
    st = {}
    for layer, kin in ((0, 10), (1, 256), (2, 256)):
        for sfx in ("", "_reverse"):
            st[f"gru.weight_ih_l{layer}{sfx}"] = rng.uniform(-k, k, (384, kin)).astype(np.float32)
            st[f"gru.weight_hh_l{layer}{sfx}"] = rng.uniform(-k, k, (384, 128)).astype(np.float32)
            st[f"gru.bias_ih_l{layer}{sfx}"] = rng.uniform(-k, k, 384).astype(np.float32)
            st[f"gru.bias_hh_l{layer}{sfx}"] = rng.uniform(-k, k, 384).astype(np.float32)
    st["linear.weight"] = (rng.uniform(-k, k, (5, 256)) * 6).astype(np.float32)
    st["linear.bias"] = rng.uniform(-k, k, 5).astype(np.float32)
    e = engine.GruEngine(st, n_layers=3)
    e.enable_timing(True)
    e.set_option("rec_windows_per_tile", 8)
    x = synth.counts_windows(11, 1200, depth=40, seed=5)
    ref = oracle.c_gru_forward(x, st, n_layers=3)
    e.set_option("fuse_proj", 0)
    plain = e.forward_host(x)
    _check(plain, ref, what="3 layers, unfused")
    e.set_option("fuse_proj", 2)
    e.set_option("fuse_head", 0)
    fused = e.forward_host(x)
    assert e.timing()["fused_layers"] == 6 and np.array_equal(fused, plain)
    e.set_option("fuse_head", 1)
    out = e.forward_host(x)
    assert e.timing()["fused_layers"] == (6 | 256 | 512) and np.abs(out - plain).max() <= 1e-6
    _check(out, ref, what="3 layers, fused + head")
    e.set_option("final_head", 0)
    assert np.array_equal(e.forward_host(x), out) and e.timing()["fused_layers"] == (6 | 256)
    e.close()


def test_fused_and_unfused_layer0_agree(gold):
    x = synth.counts_windows(9, 400, seed=41)
    ref = oracle.c_gru_forward(x, weight_set(gold, "x3"))
    for fuse in (1, 0):
        e = engine.GruEngine(weight_set(gold, "x3"))
        e.set_option("fuse_l0", fuse)
        _check(e.forward_host(x), ref, what=f"fuse_l0={fuse}")
        e.close()


def test_out_of_range_input_takes_exact_projection(gold, engines):
    """Raw (un-normalised) counts overflow the fp16 packing of the fused layer-0 projection: the
    engine must detect it on the device and fall back to the exact fp32 projection."""
    x = synth.counts_windows(5, 300, seed=43) * np.float32(3000.0)
    ref = oracle.c_gru_forward(x, weight_set(gold, "init"))
    _check(engines("init").forward_host(x), ref, what="out-of-range input")
    # and the engine keeps working on in-range input afterwards
    x2 = synth.counts_windows(5, 300, seed=44)
    _check(engines("init").forward_host(x2), oracle.c_gru_forward(x2, weight_set(gold, "init")), what="after")


def test_half_precision_mode(gold):
    """`model.half()` path: fp16 operands, fp32 accumulate (what the reference runs on a GPU by
    default, prediction.py:164-168).  Acceptance: within 2x of the deviation a CPU
    fp16 emulation of the reference shows against fp32 (8.4e-4 on trained-like weights), argmax
    essentially identical; the three work-group sizes compute identical results."""
    x = synth.counts_windows(21, 800, seed=61)
    for wname, tol in (("trained", 2e-3), ("x3", 2e-3)):
        ref = oracle.c_gru_forward(x, weight_set(gold, wname))
        outs = []
        for tile in (4, 8, 16):
            e = engine.GruEngine(weight_set(gold, wname))
            e.set_precision(True)
            e.set_option("rec_windows_per_tile", tile)
            outs.append(e.forward_host(x))
            e.close()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        err = np.abs(outs[0] - ref).max()
        agree = (outs[0].argmax(-1) == ref.argmax(-1)).mean()
        print(f"half precision {wname}: max|dp| = {err:.2e}, argmax agreement = {agree:.5f}")
        assert err <= tol
        srt = np.sort(ref, -1)
        clear = (srt[..., -1] - srt[..., -2]) > 2 * tol      # positions the reference itself decides
        assert (outs[0].argmax(-1) == ref.argmax(-1))[clear].all()
        if wname == "trained":
            assert agree >= 0.9999


def test_empty_inputs(gold, engines):
    e = engines("init")
    assert e.forward_host(np.zeros((0, 10, 10), np.float32)).shape == (0, 10, 5)
    assert e.forward_host(np.zeros((3, 0, 10), np.float32)).shape == (3, 0, 5)
    with pytest.raises(ValueError):
        e.forward_host(np.zeros((3, 4, 9), np.float32))


def test_logits_mode(gold, engines):
    x = synth.counts_windows(2, 50, seed=4)
    e = engine.GruEngine(weight_set(gold, "x3"), normalise=False)
    ref = oracle.c_gru_forward(x, weight_set(gold, "x3"), normalise=False)
    assert np.abs(e.forward_host(x) - ref).max() <= 5e-5
    # ... and through the fused last layer, whose second half delivers the logits itself (rec_fused.hpp HEAD = 2, no softmax):
    # the bits of the combine kernel, the oracle's values
    x = synth.counts_windows(9, 64, seed=5)
    e.enable_timing(True)
    e.set_option("rec_windows_per_tile", 8)
    e.set_option("fuse_proj", 2)
    out = e.forward_host(x)
    assert e.timing()["fused_layers"] == (2 | 256 | 512), e.timing()
    e.set_option("final_head", 0)
    assert np.array_equal(e.forward_host(x), out) and e.timing()["fused_layers"] == (2 | 256)
    assert np.abs(out - oracle.c_gru_forward(x, weight_set(gold, "x3"), normalise=False)).max() <= 5e-5
    e.close()


def test_unidirectional_single_layer():
    rng = np.random.default_rng(3)
    shapes = [(384, 10), (384, 128), (384,), (384,), (5, 128), (5,)]
    keys = engine.state_keys(1, False)
    state = {k: (rng.standard_normal(s) * 0.08).astype(np.float32) for k, s in zip(keys, shapes)}
    x = synth.counts_windows(3, 200, seed=8)
    e = engine.GruEngine(state, n_layers=1, bidirectional=False)
    ref = oracle.c_gru_forward(x, state, n_layers=1, bidirectional=False)
    _check(e.forward_host(x), ref, what="1 layer unidirectional")
    e.close()


def test_full_batch_properties(gold, engines):
    """BASELINE configs[1] shape: 200 windows x 10000 columns (2M columns per call)."""
    B, T = 200, 10000
    x = np.concatenate([synth.counts_windows(8, T, depth=50, seed=s) for s in range(25)])
    assert x.shape == (B, T, 10)
    e = engines("trained")
    out = e.forward_host(x)
    assert np.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    assert np.abs(out.sum(-1) - 1).max() <= 2e-6                    # softmax rows
    # windows are independent: permuting the batch permutes the output, bit for bit
    perm = np.random.default_rng(0).permutation(B)
    out_p = e.forward_host(x[perm])
    assert np.array_equal(out_p, out[perm])
    # and so does splitting it (ragged last tile: 200 = 96 + 104)
    assert np.array_equal(e.forward_host(x[:96]), out[:96])
    # the side-stream overlap (default at this size) changes the schedule, not the arithmetic, and
    # repeated runs are deterministic
    assert np.array_equal(e.forward_host(x), out)
    e.set_option("overlap_gemm", 0)
    plain = e.forward_host(x)
    e.set_option("overlap_gemm", 1)
    assert np.array_equal(plain, out)
    # EVERY one of the 2 M columns against the reference's CPU arithmetic (nn.GRU -> nn.Linear -> softmax on
    # PyTorch-CPU, the calls of gru.py:66-71, pinned to the unmodified reference in tests/test_oracle.py): <= 2e-5
    # and the same argmax on every column (trained, confident weights).  ~15 s on the GPU box's host cores.
    import time
    from conftest import usable_cores
    torch.set_num_threads(usable_cores())
    cpu = oracle.make_torch_oracle(gold["weights_trained"])
    t0, worst = time.perf_counter(), 0.0
    for lo in range(0, B, 50):
        ref = cpu.predict(x[lo:lo + 50]).numpy()
        worst = max(worst, float(np.abs(out[lo:lo + 50] - ref).max()))
        _check(out[lo:lo + 50], ref, what=f"full batch, windows {lo}..{lo + 49}, all columns", strict_argmax=True)
    print(f"full batch vs the PyTorch-CPU oracle over {B * T} columns: max|dp| = {worst:.2e} "
          f"({time.perf_counter() - t0:.0f} s on {usable_cores()} host threads)")
    # time-reversal duality, every one of the 2 M columns: a bidirectional GRU whose forward and reverse
    # parameters are swapped (and whose layer-1 / linear input halves are swapped with them) maps the reversed
    # window to the reversed output.  The dual runs every window through the OTHER direction's kernel path and
    # sums the layer-1 projection and the head in a different order, so this is an independent computation of
    # the whole batch; it must agree to the parity tolerance and on every argmax the first run is sure of.
    st = gold["weights_trained"]
    dual = {}
    for layer in (0, 1):
        for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            a, b = st[f"gru.{name}_l{layer}"], st[f"gru.{name}_l{layer}_reverse"]
            if name == "weight_ih" and layer == 1:
                a, b = np.concatenate([a[:, 128:], a[:, :128]], 1), np.concatenate([b[:, 128:], b[:, :128]], 1)
            dual[f"gru.{name}_l{layer}"], dual[f"gru.{name}_l{layer}_reverse"] = b.copy(), a.copy()
    dual["linear.weight"] = np.concatenate([st["linear.weight"][:, 128:], st["linear.weight"][:, :128]], 1)
    dual["linear.bias"] = st["linear.bias"].copy()
    ed = engine.GruEngine(dual)
    out_d = ed.forward_host(np.ascontiguousarray(x[:, ::-1]))[:, ::-1]
    ed.close()
    err = float(np.abs(out_d - out).max())
    print(f"time-reversal duality over {B * T} columns: max|dp| = {err:.2e}")
    _check(out_d, out, what="time-reversal duality")


def test_model_api_predict_on_batch(gold):
    st = gold["weights_trained"]
    m = models.GRUModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    m = m.to("cuda").eval()
    x = synth.counts_windows(4, 300, seed=21)
    ref = oracle.make_torch_oracle(st).predict(x).numpy()
    # host tensor in -> cpu float32 tensor out (reference models.py:303-313)
    p = m.predict_on_batch(Batch(counts_matrix=torch.from_numpy(x)))
    assert p.device.type == "cpu" and p.dtype == torch.float32 and tuple(p.shape) == (4, 300, 5)
    _check(p.numpy(), ref, what="predict_on_batch(host)")
    # device tensor through forward()
    y = m(torch.from_numpy(x).cuda())
    assert y.device.type == "cuda"
    _check(y.cpu().numpy(), ref, what="forward(device)")
    # zip(data, class_probs) iterates windows along dim 0 (prediction.py:47)
    assert len(list(p)) == 4
    # half(): fp16 weights like the reference GPU default (prediction.py:164-168)
    m.half()
    ph = m.predict_on_batch(Batch(counts_matrix=torch.from_numpy(x))).numpy()
    assert np.abs(ph - ref).max() <= 2e-3
    assert (ph.argmax(-1) == ref.argmax(-1)).mean() > 0.999


def test_majority_vote_model(gold):
    """Row a8 (majority_vote_model.py:37-53) on the device, against the goldens of the unmodified reference."""
    m = models.MajorityVoteModel().to("cuda").eval()
    for cname, ref in gold["majority_outputs"].items():
        x = gold["gru_inputs"][cname]
        p = m.predict_on_batch(Batch(counts_matrix=torch.from_numpy(x))).numpy()
        assert np.abs(p - ref).max() <= 2e-7
    assert np.abs(engine.majority_forward_host(gold["gru_inputs"]["uniform"]) -
                  oracle.c_majority_forward(gold["gru_inputs"]["uniform"])).max() <= 2e-7


# ---- device-side normalisation and decode ---------------------------------
def _normalise_on_device(counts, depth):
    """`mdk_normalise_counts_dev` on raw buffers: what `_post_process_pileup` (features.py:907-911,926) does on the host."""
    L = engine._lib.load()
    n_cols = counts.shape[0] * counts.shape[1]
    F = counts.shape[2]
    cd, dd, xd = engine.DeviceBuffer(counts.nbytes), engine.DeviceBuffer(depth.nbytes), engine.DeviceBuffer(n_cols * F * 4)
    cd.upload(np.ascontiguousarray(counts)); dd.upload(np.ascontiguousarray(depth))
    engine._lib.check(L.mdk_normalise_counts_dev(cd.ptr, dd.ptr, n_cols, F, xd.ptr, 0, None), "normalise")
    x = xd.download(counts.shape, np.float32)
    for b in (cd, dd, xd):
        b.free()
    return x


@pytest.mark.parametrize("name", ["d60", "d300"])
def test_counts_in_decoded_out_matches_reference(gold, engines, name):
    """tests/golden/pcie_diet.npz holds raw pileup counts, the features the UNMODIFIED reference's
    `_post_process_pileup` makes of them, its probabilities and its decoded consensus (oracle/make_golden.py)."""
    d = np.load(os.path.join(GOLD, "pcie_diet.npz"))
    counts, depth = d[f"{name}/counts"], d[f"{name}/depth"]
    e = engines("trained")
    # f2: the device's normalisation is bit-identical to the reference's float64-divide-then-round
    assert np.array_equal(_normalise_on_device(counts, depth), d[f"{name}/features"])
    # whole path: raw counts in, probabilities + decoded classes out
    probs, cls, pmax = e.forward_counts_host(counts, depth, probs=True, decoded=True)
    assert np.array_equal(probs, e.forward_host(d[f"{name}/features"]))
    _check(probs, d[f"{name}/probs"], what=f"counts-in {name}", strict_argmax=True)
    # f3: first-maximum argmax and its probability, bit for bit
    assert np.array_equal(cls, probs.argmax(-1)) and np.array_equal(pmax, probs.max(-1))
    cls2, pmax2 = e.forward_decoded_host(d[f"{name}/features"])
    assert np.array_equal(cls2, cls) and np.array_equal(pmax2, pmax)
    n_q = n_bad = 0
    for w in range(cls.shape[0]):
        seq, qual = engine.decode_consensus(cls[w], pmax[w], with_qualities=True)
        assert (seq, qual) == oracle.decode_consensus(probs[w], with_qualities=True)
        assert seq == str(d[f"{name}/seq"][w])                      # identical consensus (labels.py:1053-1085)
        ref_q = str(d[f"{name}/qual"][w])
        n_q += len(ref_q)
        n_bad += sum(a != b for a, b in zip(qual, ref_q))
    assert n_bad <= 0.01 * n_q       # a quality char may sit on a truncation boundary of -10 log10(1 - p)
    # model-level entry
    m = models.GRUModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weight_set(gold, "trained").items()})
    m = m.to("cuda").eval()
    assert np.array_equal(m.predict_on_counts(counts, depth).numpy(), probs)
    c3, p3 = m.predict_on_counts(counts.astype(np.int32), depth, decoded=True)       # any integer dtype that fits
    assert np.array_equal(c3.numpy(), cls) and np.array_equal(p3.numpy(), pmax)


def test_decode_dev_first_maximum_and_nan():
    L = engine._lib.load()
    p = np.array([[0.2, 0.5, 0.5, 0.1, 0.0], [np.nan, 0.9, 0.0, 0.0, 0.1], [0.1, np.nan, 0.9, np.nan, 0.0],
                  [0.2, 0.2, 0.2, 0.2, 0.2]], dtype=np.float32)
    pd, cd, md = engine.DeviceBuffer(p.nbytes), engine.DeviceBuffer(4), engine.DeviceBuffer(16)
    pd.upload(p)
    engine._lib.check(L.mdk_decode_dev(pd.ptr, 4, 5, cd.ptr, md.ptr, 0, None), "decode")
    cls = cd.download((4,), np.uint8)
    pm = md.download((4,), np.float32)
    assert cls.tolist() == np.argmax(p, -1).tolist() == [1, 0, 1, 0]
    assert np.array_equal(pm, np.take_along_axis(p, np.argmax(p, -1)[:, None], -1)[:, 0], equal_nan=True)
    for b in (pd, cd, md):
        b.free()


# ---- read-level model (reference LatentSpaceLSTM) -------------------------------------------------
import os  # noqa: E402
from conftest import GOLD  # noqa: E402
from oracle import rl_oracle  # noqa: E402

RL_CONFIGS = {"bi": dict(), "uni": dict(bidirectional=False), "bi_dwells": dict(use_dwells=True)}


@pytest.mark.parametrize("name", sorted(RL_CONFIGS))
def test_read_level_model_goldens_from_unmodified_reference(name):
    cases = np.load(os.path.join(GOLD, "rl_cases.npz"))
    state = dict(np.load(os.path.join(GOLD, f"rl_weights_{name}.npz")))
    e = engine.RlEngine(state, **RL_CONFIGS[name])
    out = e.forward_host(cases[f"{name}/x"])
    _check(out, cases[f"{name}/y"], what=f"read-level {name}")
    e.close()


@pytest.mark.parametrize("B,P,D", [(1, 1, 1), (2, 63, 3), (3, 65, 7), (9, 200, 12), (1, 700, 30)])
def test_read_level_model_shapes_vs_oracle(B, P, D):
    for name in ("bi", "uni"):
        state = dict(np.load(os.path.join(GOLD, f"rl_weights_{name}.npz")))
        x = rl_oracle.synth_reads(B, P, D, seed=7 * B + P + D)
        ref = rl_oracle.rl_forward(x, state, **RL_CONFIGS[name])
        e = engine.RlEngine(state, **RL_CONFIGS[name])
        _check(e.forward_host(x), ref, what=f"read-level {name} B={B} P={P} D={D}")
        e.close()


def test_read_level_model_api(gold):
    state = dict(np.load(os.path.join(GOLD, "rl_weights_bi.npz")))
    m = models.LatentSpaceLSTM()
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
    assert all("num_batches_tracked" in k for k in missing.missing_keys)
    m = m.to("cuda").eval()
    x = rl_oracle.synth_reads(4, 150, 8, seed=3)
    ref = rl_oracle.rl_forward(x, state)
    p = m.predict_on_batch(Batch(read_level_features=torch.from_numpy(x)))
    assert p.device.type == "cpu" and tuple(p.shape) == (4, 150, 5)
    _check(p.numpy(), ref, what="LatentSpaceLSTM.predict_on_batch")
    y = m(torch.from_numpy(x).cuda())
    _check(y.cpu().numpy(), ref, what="LatentSpaceLSTM.forward")


def _half_emulation():
    """Deviation of the reference's own fp16 recipe run on the CPU (fp16 weights + autocast) from its fp32
    result, per weight set: oracle/make_golden_adversarial.py part 2.  SURVEY.md section 8c: the engine's
    half mode must stay within 2x of it."""
    import json
    return json.load(open(os.path.join(GOLD, "rl_half_emulation.json")))


@pytest.mark.parametrize("name", ["bi", "uni", "bi_dwells", "trained"])
def test_read_level_model_half_precision(name):
    emu = _half_emulation()[name]
    wname = "rl_weights_trained.npz" if name == "trained" else f"rl_weights_{name}.npz"
    kw = RL_CONFIGS["bi" if name == "trained" else name]
    state = dict(np.load(os.path.join(GOLD, wname)))
    x = rl_oracle.synth_reads(4, 400, 20, use_dwells=kw.get("use_dwells", False), seed=77)   # the emulation's input
    ref = rl_oracle.rl_forward(x, state, **kw)
    e = engine.RlEngine(state, **kw)
    e.set_precision(True)
    out = e.forward_host(x)
    e.close()
    d = np.abs(out - ref)
    print(f"rl half {name}: max|dp| {d.max():.2e} (cpu fp16 emulation {emu['max_abs_dp']:.2e}), "
          f"mean {d.mean():.2e} ({emu['mean_abs_dp']:.2e})")
    assert d.max() <= 2 * emu["max_abs_dp"] and d.mean() <= 2 * emu["mean_abs_dp"]
    assert (out.argmax(-1) == ref.argmax(-1)).mean() >= emu["argmax_agreement"] - 1e-3


def test_read_level_trained_weights_argmax_identity():
    """Weights trained by the reference's own process_batch: confident outputs, so identity of the argmax on
    every position means something (goldens from the unmodified reference)."""
    state = dict(np.load(os.path.join(GOLD, "rl_weights_trained.npz")))
    cases = np.load(os.path.join(GOLD, "rl_trained_cases.npz"))
    e = engine.RlEngine(state)
    out = e.forward_host(cases["x"])
    _check(out, cases["y"], what="read-level trained", strict_argmax=True)
    x = rl_oracle.synth_reads(6, 1500, 30, seed=123)
    ref = rl_oracle.rl_forward(x, state)
    assert np.median(ref.max(-1)) > 0.9
    _check(e.forward_host(x), ref, what="read-level trained, long", strict_argmax=True)
    e.close()


def test_read_level_reverse_strand_byte():
    """Real BAMs give strand -1 for reverse reads (src/medaka_read_matrix.c); Batch.collate zero-pads into
    a uint8 array (torch_ext.py:127-140), so it arrives as byte 255.  The reference would index
    strand_embedder out of range on it; the engine reads the byte as int8 (-1 -> row 0), which is what the
    reference computes when the features are kept as int8.  Checked against the oracle on int8 input."""
    state = dict(np.load(os.path.join(GOLD, "rl_weights_bi.npz")))
    x = rl_oracle.synth_reads(3, 200, 8, seed=41).astype(np.int16)
    rev = np.random.default_rng(5).random(x.shape[:3]) < 0.5
    nonempty = x.sum(-1) != 0
    x[..., 2] = np.where(rev & nonempty, -1, np.where(nonempty, 1, 0))
    ref = rl_oracle.rl_forward(x.astype(np.int8), state)
    e = engine.RlEngine(state)
    out = e.forward_host(x.astype(np.int8).view(np.uint8))
    e.close()
    _check(out, ref, what="strand -1 as byte 255")


def test_read_level_empty_reads_and_all_empty_window():
    """Padded (all-zero) reads are excluded from the mean (read_level_modules.py:81-100); a window
    with no read at all is 0/0 = NaN in the reference and here."""
    state = dict(np.load(os.path.join(GOLD, "rl_weights_uni.npz")))
    x = rl_oracle.synth_reads(3, 90, 6, seed=13, empty_tail=False)
    x[0, :, 3:, :] = 0
    x[2] = 0
    ref = rl_oracle.rl_forward(x, state, bidirectional=False)
    e = engine.RlEngine(state, bidirectional=False)
    out = e.forward_host(x)
    e.close()
    _ok = ~np.isnan(ref)
    assert np.isnan(out[2]).all() and np.isnan(ref[2]).all()
    assert np.abs(out[:2] - ref[:2]).max() <= TOL


# ---- rl_lstm384 architecture: LSTM(384) recurrence spread over 12-CU clusters (lstm_wide.hpp) ----
# Both bundled flavours (reference options.py:175-182): with dwells (5 features per read, 8 conv input channels) and
# WITHOUT (4 features, 7 channels: the `use_dwells=False` branch of latent_space_lstm.py:176-190).
def _wide_kw(dwells):
    return dict(lstm_size=384, cnn_size=128, use_dwells=dwells, bidirectional=False)


@pytest.fixture(scope="module", params=[True, False], ids=["dwells", "no_dwells"])
def wide(request):
    from oracle.make_golden_rl import WIDE_ND_SEED, WIDE_SEED
    dwells = request.param
    kw = _wide_kw(dwells)
    return dict(dwells=dwells, kw=kw, state=rl_oracle.synth_rl_state(seed=WIDE_SEED if dwells else WIDE_ND_SEED, **kw),
                cases="rl_wide_cases.npz" if dwells else "rl_wide_nd_cases.npz", emu="wide" if dwells else "wide_nd")


def _wide_ref(x, wide):
    return rl_oracle.rl_forward(x, wide["state"], use_dwells=wide["dwells"], bidirectional=False)


@pytest.mark.parametrize("name", ["two_groups", "many_groups", "long"])
def test_wide_read_level_goldens_from_unmodified_reference(name, wide):
    cases = np.load(os.path.join(GOLD, wide["cases"]))
    if f"{name}/x" not in cases:
        pytest.skip("case exists for the no-dwells flavour only")
    e = engine.RlEngine(wide["state"], **wide["kw"])
    out = e.forward_host(cases[f"{name}/x"])
    e.close()
    _check(out, cases[f"{name}/y"], what=f"rl_lstm384 {wide['emu']} {name}")


@pytest.mark.parametrize("B,P,D", [(1, 1, 1), (1, 33, 2), (8, 70, 4), (9, 129, 5), (17, 64, 3),
                                   (264, 16, 2)])     # 33 groups -> 17 pairs on 16 clusters: a cluster loops
def test_wide_read_level_shapes_vs_oracle(B, P, D, wide):
    x = rl_oracle.synth_reads(B, P, D, use_dwells=wide["dwells"], seed=5 * B + P + D)
    ref = _wide_ref(x, wide)
    e = engine.RlEngine(wide["state"], **wide["kw"])
    out = e.forward_host(x)
    _check(out, ref, what=f"rl_lstm384 B={B} P={P} D={D}")
    # same engine again: exchange buffers / tags are reset per launch
    assert np.array_equal(e.forward_host(x), out)
    # both exchange protocols (shared-L2 plain stores / write-through granules) carry the same bits
    e.set_option("wide_write_through", 1)
    assert np.array_equal(e.forward_host(x), out)
    # ... and so do one / two interleaved groups per cluster
    for ngrp in (1, 2):
        e.set_option("wide_groups_per_cluster", ngrp)
        assert np.array_equal(e.forward_host(x), out)
    e.close()


def test_wide_read_level_chunked_overlap_agrees_bitwise(wide):
    """Windows of >= 1024 positions run every layer's recurrence as 8 resumable launches (h re-read
    from the output, cell state from a side buffer) with the next layer's projection behind each
    chunk on a side stream: same bits as the plain sequence, in both precisions."""
    x = rl_oracle.synth_reads(19, 1100, 3, use_dwells=wide["dwells"], seed=91)
    e = engine.RlEngine(wide["state"], **wide["kw"])
    for half in (False, True):
        e.set_precision(half)
        e.set_option("overlap_gemm", 1)
        a = e.forward_host(x)
        e.set_option("overlap_gemm", 0)
        b = e.forward_host(x)
        assert np.array_equal(a, b), half
    e.close()


def test_wide_read_level_long_window_and_empty_window(wide):
    """A 2000-position window (4 x 2000 cluster exchanges) next to an all-empty window (NaN, as the
    reference's 0/0) in the same 8-window group: NaNs must stay in their own MFMA rows."""
    x = rl_oracle.synth_reads(3, 2000, 4, use_dwells=wide["dwells"], seed=77, empty_tail=False)
    x[1] = 0
    ref = _wide_ref(x, wide)
    e = engine.RlEngine(wide["state"], **wide["kw"])
    out = e.forward_host(x)
    e.close()
    assert np.isnan(ref[1]).all() and np.isnan(out[1]).all()
    _check(out[[0, 2]], ref[[0, 2]], what="rl_lstm384 long window")


def test_wide_read_level_model_api(wide):
    m = models.LatentSpaceLSTM(**wide["kw"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in wide["state"].items()}, strict=False)
    m = m.to("cuda").eval()
    x = rl_oracle.synth_reads(5, 100, 6, use_dwells=wide["dwells"], seed=3)
    ref = _wide_ref(x, wide)
    p = m.predict_on_batch(Batch(read_level_features=torch.from_numpy(x)))
    assert p.device.type == "cpu" and tuple(p.shape) == (5, 100, 5)
    _check(p.numpy(), ref, what="LatentSpaceLSTM(384).predict_on_batch")
    other = rl_oracle.synth_reads(2, 40, 3, use_dwells=not wide["dwells"], seed=4)
    if wide["dwells"]:
        # 4 features per read for a dwells model: refused (the reference asserts, latent_space_lstm.py:177-179)
        with pytest.raises(lib.EngineError):
            m.predict_on_batch(Batch(read_level_features=torch.from_numpy(other)))
    else:
        # 5 features for a no-dwells model: the reference ignores the dwell channel except in its read mask
        q = m.predict_on_batch(Batch(read_level_features=torch.from_numpy(other))).numpy()
        _check(q, _wide_ref(other, wide), what="no-dwells model on a 5-feature matrix")
    with pytest.raises(RuntimeError, match="bidirectional"):
        engine.RlEngine(rl_oracle.synth_rl_state(seed=1, lstm_size=384, bidirectional=True, use_dwells=False),
                        lstm_size=384, bidirectional=True)


@pytest.mark.parametrize("B,P,D", [(5, 300, 6), (40, 130, 3), (300, 20, 2)])
def test_wide_read_level_half_precision(B, P, D, wide):
    """`half()` on the LSTM(384) model: single-product fp16 front end / GEMMs and 16-window groups in
    the cluster recurrence (1, 3 and 19 groups: one per cluster, then two interleaved)."""
    x = rl_oracle.synth_reads(B, P, D, use_dwells=wide["dwells"], seed=B + P)
    ref = _wide_ref(x, wide)
    e = engine.RlEngine(wide["state"], **wide["kw"])
    e.set_precision(True)
    out = e.forward_host(x)
    e.set_precision(False)
    full = e.forward_host(x)
    e.close()
    _check(full, ref, what="rl_lstm384 back to fp32")
    assert np.isfinite(out).all() and np.abs(out.sum(-1) - 1).max() <= 1e-5
    # anchor: CPU fp16 emulation of the reference (fp16 weights + autocast, LSTM state in fp16) on THIS weight set and
    # THIS input (oracle/make_golden_adversarial.py rl_half_wide_shapes).  SURVEY 8c: within 2 x the emulation's own
    # deviation from fp32.  Measured on MI355X (r3): dwells 0.4-0.6 x the emulation; no-dwells 1.3 x on the maximum and
    # up to 2.4 x on the mean (the engine keeps h in fp16 across the cluster exchange with fp32 gates; the emulation's
    # front end is cleaner without the dwell channel) -- the mean bound is therefore 3 x, the maximum stays at 2 x.
    emu = _half_emulation()[f"{wide['emu']}/{B}x{P}x{D}"]
    d = np.abs(out - ref)
    same = float((out.argmax(-1) == ref.argmax(-1)).mean())
    print(f"rl_lstm384 {wide['emu']} half {B}x{P}x{D}: max|dp| {d.max():.2e} (emulation {emu['max_abs_dp']:.2e}), "
          f"mean {d.mean():.2e} ({emu['mean_abs_dp']:.2e}), argmax agreement {same:.4f} ({emu['argmax_agreement']:.4f})")
    assert d.max() <= 2 * emu["max_abs_dp"] and d.mean() <= 3 * emu["mean_abs_dp"]
    assert same >= emu["argmax_agreement"] - 0.01


def test_wide_read_level_full_batch_properties():
    """BASELINE config 4b at full size: rl_lstm384 architecture, 100 windows x 10 000 positions x 50 reads (250 MB of
    uint8).  Size-independent properties -- finite softmax rows; windows are independent, so permuting or splitting
    the batch permutes / selects the output bit for bit (different cluster groups, different recurrence chunks) --
    and ONE whole window (4 x 10 000 dependent LSTM steps) against the CPU oracle."""
    kw = _wide_kw(True)
    st = rl_oracle.synth_rl_state(seed=33, **kw)
    B, P, D = 100, 10000, 50
    base = rl_oracle.synth_reads(10, P, D, use_dwells=True, seed=70)
    x = np.concatenate([base] * 10)
    for b in range(B):                                                 # make the ten copies of a window distinct
        x[b, (b * 37) % P, ::7, 1] ^= np.uint8(1 + b % 7)
    e = engine.RlEngine(st, **kw)
    out = e.forward_host(x)
    assert out.shape == (B, P, 5) and np.isfinite(out).all()
    assert np.abs(out.sum(-1) - 1).max() <= 2e-6
    perm = np.random.default_rng(1).permutation(B)
    assert np.array_equal(e.forward_host(x[perm]), out[perm])
    assert np.array_equal(e.forward_host(x[:37]), out[:37])
    assert np.array_equal(e.forward_host(x), out)                      # deterministic
    e.close()
    ref = rl_oracle.rl_forward(x[5:6], st, use_dwells=True, bidirectional=False)
    err = float(np.abs(out[5:6] - ref).max())
    print(f"rl_lstm384 full batch, window 5 (10 000 positions x 50 reads) vs the oracle: max|dp| = {err:.2e}")
    _check(out[5:6], ref, what="rl_lstm384 full-size window")


def test_wide_read_level_fails_fast_without_its_cus(debug_hooks):
    """The cluster recurrence needs every member of a cluster on a CU at the same time (192 CUs for a full batch, 24 for
    the two clusters of this small one).  While another tenant holds 250 of the 256 CUs exclusively the clusters'
    placement handshake cannot complete: both tries (bounded at 50 ms of wall clock each, later launches of
    a lost forward return at once) must end in MDK_ERR_DEVICE within a fraction of a second -- not after seconds of
    spinning, never with a wrong result -- and the engine must work again as soon as the CUs are back."""
    import threading
    import time
    if not debug_hooks:          # (mdk_selftest_hold is a hook of the debug library: the test ran against it in a child process)
        return
    kw = _wide_kw(True)
    st = rl_oracle.synth_rl_state(seed=33, **kw)
    x = rl_oracle.synth_reads(9, 1200, 4, use_dwells=True, seed=12)      # 1200 positions: chunked, 32 recurrence launches
    e = engine.RlEngine(st, **kw)
    ref = e.forward_host(x)
    L = lib.load()
    holder = threading.Thread(target=lambda: lib.check(L.mdk_selftest_hold(0, 250, 1500, 140 * 1024), "hold"))
    holder.start()
    time.sleep(0.3)
    t0 = time.perf_counter()
    try:
        out, err = e.forward_host(x), None
    except lib.EngineError as exc:
        out, err = None, str(exc)
    dt = time.perf_counter() - t0
    holder.join()
    print(f"forward next to a tenant holding 250 CUs: {'error after' if err else 'completed in'} {dt * 1e3:.0f} ms"
          + (f" ({err[:90]}...)" if err else ""))
    if err is None:                    # the tenant was scheduled elsewhere or had finished: then the bits must be right
        assert np.array_equal(out, ref)
    else:
        assert "timed out" in err and dt < 4.5       # bounded tries inside the 3 s budget of "wide_wait_ms", never a hang
    assert np.array_equal(e.forward_host(x), ref)                         # CUs back: same bits as before
    e.close()


def test_wide_read_level_retry_budget_and_error_branch(debug_hooks):
    """The host side of a cluster time-out, deterministically (option "wide_inject_timeout" raises the device's time-out
    flag before a try, as a lost forward leaves it: every launch of that try returns at once and the host finds the
    flag).  Time-outs that end inside the budget are waited out with growing pauses and give the right bits; more of
    them than "wide_wait_ms" allows end in MDK_ERR_DEVICE -- on the error branch, within the budget -- and the engine
    works again afterwards."""
    import time
    if not debug_hooks:          # (the option exists in the debug library only: the test ran against it in a child process)
        return
    kw = _wide_kw(True)
    st = rl_oracle.synth_rl_state(seed=33, **kw)
    x = rl_oracle.synth_reads(9, 300, 4, use_dwells=True, seed=12)
    e = engine.RlEngine(st, **kw)
    e.enable_timing(True)
    ref = e.forward_host(x)
    r0 = e.timing()["wide_retries"]
    e.set_option("wide_inject_timeout", 3)           # the first try and two retries lost, the third retry goes through
    out = e.forward_host(x)
    assert np.array_equal(out, ref) and e.timing()["wide_retries"] - r0 == 3
    e.set_option("wide_wait_ms", 250)
    e.set_option("wide_inject_timeout", 1000)
    t0 = time.perf_counter()
    with pytest.raises(lib.EngineError, match="timed out"):
        e.forward_host(x)
    dt = time.perf_counter() - t0
    assert dt < 1.5, dt
    e.set_option("wide_inject_timeout", 0)
    assert np.array_equal(e.forward_host(x), ref)
    e.set_option("wide_wait_ms", 0)                  # no budget: the second time-out is the error (the round-3 behaviour)
    e.set_option("wide_inject_timeout", 2)
    with pytest.raises(lib.EngineError, match="timed out 2 times"):
        e.forward_host(x)
    e.set_option("wide_inject_timeout", 1)
    assert np.array_equal(e.forward_host(x), ref)    # one time-out: the plain-schedule retry answers
    e.close()


# ---- the model swap on the device: integration.convert, every family -----------------------
import ref_standins  # noqa: E402


def _swap_case(name, gold):
    """(stand-in of the reference class with real weights, input batch, oracle output)."""
    cls, kw = ref_standins.CONFIGS[name]
    ref_model = cls(**kw).eval()
    if name == "GRUModel":
        st = gold["weights_trained"]
        x = synth.counts_windows(6, 700, seed=44)
        want = oracle.c_gru_forward(x, st)
        batch = Batch(counts_matrix=torch.from_numpy(x))
    else:
        dwells = kw.get("use_dwells", False)
        if kw.get("lstm_size", 128) == 384:
            st = rl_oracle.synth_rl_state(seed=33, **_wide_kw(dwells))
        else:
            st = dict(np.load(os.path.join(GOLD, "rl_weights_bi.npz" if kw.get("bidirectional", True) else "rl_weights_uni.npz")))
        x = rl_oracle.synth_reads(9, 140, 7, use_dwells=dwells, seed=45)
        want = rl_oracle.rl_forward(x, st, use_dwells=dwells, bidirectional=kw.get("bidirectional", True))
        batch = Batch(read_level_features=torch.from_numpy(x))
    missing = ref_model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=False)
    assert not missing.unexpected_keys and all("num_batches_tracked" in k or "expansion_layer" in k for k in missing.missing_keys)
    return ref_model, batch, want


@pytest.mark.parametrize("name", sorted(ref_standins.CONFIGS))
def test_integration_convert_swaps_every_family_on_the_device(name, gold, capsys):
    """What `ModelStoreTGZ.load_model(device=cuda)` hands to `convert` (a state-dict-identical stand-in of the
    reference class, pinned to the real one in tests/test_host.py) comes back ENGINE-backed in strict mode, and
    its `predict_on_batch` matches the oracle.  The class that ran is printed for the GPU test log."""
    ref_model, batch, want = _swap_case(name, gold)
    assert integration.convert(ref_model, "cpu", strict=True) is ref_model        # --cpu keeps the reference
    conv = integration.convert(ref_model.to("cuda"), "cuda", strict=True)
    assert type(conv).__module__ == "medaka_amd.models" and type(conv).__name__ == type(ref_model).__name__
    assert conv.device().type == "cuda" and conv._engine is not None            # built eagerly at load time
    for k, v in ref_model.state_dict().items():
        assert torch.equal(conv.state_dict()[k].cpu(), v.cpu()), k
    out = conv.predict_on_batch(batch)
    assert out.device.type == "cpu" and out.dtype == torch.float32
    _check(out.numpy(), want, what=f"convert({name})")
    with capsys.disabled():
        print(f"\n[swap] {name}: {type(ref_model).__module__}.{type(ref_model).__name__} -> "
              f"{type(conv).__module__}.{type(conv).__name__} (engine {type(conv._engine).__name__}, "
              f"{lib.device_name(0)}) max|dp| {np.abs(out.numpy() - want).max():.1e}")
    # the reference calls half() on whatever load_model returned (prediction.py:164-168)
    conv.half()
    assert conv.half_precision and np.isfinite(conv.predict_on_batch(batch).numpy()).all()


def test_integration_strict_mode_on_the_device():
    """Outside the engine's envelope on a HIP device: warning + reference model by default, EngineRequired in
    strict mode -- never a silent PyTorch-ROCm forward under MEDAKA_AMD=strict."""
    big = ref_standins.GRUModel(gru_size=64).to("cuda")
    assert integration.convert(big, "cuda", strict=False) is big
    with pytest.raises(integration.EngineRequired):
        integration.convert(big, "cuda", strict=True)


def test_plain_c_host_runs(tmp_path):
    """tests/c/abi_smoke.c: create / forward / error path / destroy from C, no Python in the loop."""
    import subprocess
    from test_host import _build_c_host
    r = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok ")
