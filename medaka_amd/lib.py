"""ctypes binding of the C ABI in include/medaka_amd.h (libmedaka_amd.so, built in-tree).

There is no Python or CPU fallback: if the shared library is missing or does not export the
full ABI, importing the engine raises.  ctypes releases the GIL around every call, which the
reference's threaded DataLoader relies on (reference medaka/prediction.py:268-279 runs loader,
batcher and HDF-writer threads next to `model.predict_on_batch`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDK_LIB") or os.path.join(_HERE, "libmedaka_amd.so")

MDK_OK, MDK_ERR_ARG, MDK_ERR_DEVICE, MDK_ERR_OOM = 0, 1, 2, 3
MDK_PREC_FP32, MDK_PREC_FP16 = 0, 1
MDK_VARIANT_MFMA, MDK_VARIANT_EXACT = 0, 1


class GruDesc(ctypes.Structure):
    _fields_ = [("num_features", ctypes.c_int), ("hidden", ctypes.c_int),
                ("num_layers", ctypes.c_int), ("bidirectional", ctypes.c_int),
                ("num_classes", ctypes.c_int), ("normalise", ctypes.c_int)]


class RlDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "lstm_size", "cnn_size", "kernel_size0", "kernel_size1", "use_dwells", "alphabet_size",
        "embedding_size", "bidirectional", "num_classes", "normalise")]


class RlTiming(ctypes.Structure):
    _fields_ = [("front_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("wide_retries", ctypes.c_int)]


class GruTiming(ctypes.Structure):
    _fields_ = [("h2d_ms", ctypes.c_float), ("gi_ms", ctypes.c_float * 4),
                ("rec_ms", ctypes.c_float * 4), ("head_ms", ctypes.c_float),
                ("d2h_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("rec_launches", ctypes.c_int), ("n_layers", ctypes.c_int), ("host_streamed", ctypes.c_int),
                ("fused_layers", ctypes.c_int)]


class GruSplit(ctypes.Structure):
    _fields_ = [("chunks", ctypes.c_int), ("margin", ctypes.c_int), ("columns", ctypes.c_int),
                ("status", ctypes.c_int), ("max_delta", ctypes.c_float), ("fallbacks", ctypes.c_int),
                ("audited", ctypes.c_int), ("audit_max_dp", ctypes.c_float), ("audits", ctypes.c_int),
                ("audit_failures", ctypes.c_int), ("audit_worst_dp", ctypes.c_float), ("probes", ctypes.c_int),
                ("probe_max_delta", ctypes.c_float)]


class SplitShape(ctypes.Structure):
    _fields_ = [("chunks", ctypes.c_int), ("columns", ctypes.c_int), ("margin", ctypes.c_int),
                ("start", ctypes.c_int * 16), ("first", ctypes.c_int * 16), ("last", ctypes.c_int * 16)]


class PassShape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "windows_per_group", "work_groups", "fuse_layer0", "fuse_projection", "fuse_head", "final_head",
        "overlap_gemm", "stream_in", "stream_out", "needs_gi")]


SPLIT_STATUS = {0: "not used", 1: "certified", 2: "rejected", 3: "disabled"}


# every symbol include/medaka_amd.h declares: (restype, argtypes)
_vp, _i, _l, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
ABI = {
    "mdk_gru_create": (_i, [ctypes.POINTER(GruDesc), ctypes.POINTER(_vp), _i, _i, ctypes.POINTER(_vp)]),
    "mdk_gru_forward": (_i, [_vp, _vp, _i, _i, _vp]),
    "mdk_gru_forward_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mdk_gru_set_precision": (_i, [_vp, _i]),
    "mdk_gru_set_variant": (_i, [_vp, _i]),
    "mdk_gru_set_normalise": (_i, [_vp, _i]),
    "mdk_gru_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "mdk_gru_enable_timing": (_i, [_vp, _i]),
    "mdk_gru_get_timing": (_i, [_vp, ctypes.POINTER(GruTiming)]),
    "mdk_gru_get_split": (_i, [_vp, ctypes.POINTER(GruSplit)]),
    "mdk_gru_stage_input": (_i, [_vp, _vp, _i, _i, ctypes.POINTER(ctypes.c_ulonglong)]),
    "mdk_gru_forward_staged": (_i, [_vp, ctypes.c_ulonglong, _i, _i, _vp]),
    "mdk_gru_forward_pipelined": (_i, [_vp, ctypes.c_ulonglong, _i, _i, _vp, _vp]),
    "mdk_gru_drop_pending": (_i, [_vp]),
    "mdk_split_plan": (_i, [_i, _i, _i, _i, _i, ctypes.POINTER(SplitShape)]),
    "mdk_margin_sim": (_i, [_i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "mdk_pass_plan": (_i, [ctypes.POINTER(GruDesc), _i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(PassShape)]),
    "mdk_gru_device": (_i, [_vp]),
    "mdk_gru_destroy": (None, [_vp]),
    "mdk_rl_create": (_i, [ctypes.POINTER(RlDesc), ctypes.POINTER(_vp), _i, _i, ctypes.POINTER(_vp)]),
    "mdk_rl_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mdk_rl_forward_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mdk_rl_check": (_i, [_vp, _vp]),
    "mdk_rl_enable_timing": (_i, [_vp, _i]),
    "mdk_rl_get_timing": (_i, [_vp, ctypes.POINTER(RlTiming)]),
    "mdk_rl_set_precision": (_i, [_vp, _i]),
    "mdk_rl_set_normalise": (_i, [_vp, _i]),
    "mdk_rl_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "mdk_rl_device": (_i, [_vp]),
    "mdk_rl_destroy": (None, [_vp]),
    "mdk_normalise_counts_dev": (_i, [_vp, _vp, _l, _i, _vp, _i, _vp]),
    "mdk_decode_dev": (_i, [_vp, _l, _i, _vp, _vp, _i, _vp]),
    "mdk_gru_forward_counts": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "mdk_gru_forward_decoded": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mdk_majority_forward_dev": (_i, [_vp, _l, _vp, _i, _vp]),
    "mdk_majority_forward": (_i, [_vp, _l, _vp, _i]),
    "mdk_device_count": (_i, [ctypes.POINTER(_i)]),
    "mdk_device_name": (_i, [_i, ctypes.c_char_p, _sz]),
    "mdk_dev_alloc": (_i, [_i, _sz, ctypes.POINTER(_vp)]),
    "mdk_dev_free": (_i, [_i, _vp]),
    "mdk_host_alloc": (_i, [_sz, ctypes.POINTER(_vp)]),
    "mdk_host_free": (_i, [_vp]),
    "mdk_gather_rows": (_i, [_vp, ctypes.POINTER(_vp), _i, _sz, _i]),
    "mdk_memcpy_h2d": (_i, [_i, _vp, _vp, _sz]),
    "mdk_memcpy_d2h": (_i, [_i, _vp, _vp, _sz]),
    "mdk_device_synchronize": (_i, [_i]),
    "mdk_selftest_mfma": (_i, [_i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i)]),
    "mdk_last_error": (ctypes.c_char_p, []),
    "mdk_version": (ctypes.c_char_p, []),
}

# present only in the debug library (libmedaka_amd_debug.so, -DMDK_DEBUG_HOOKS): typed when found
DEBUG_ABI = {
    "mdk_selftest_burn": (_i, [_i, _i, _i]),
    "mdk_selftest_hold": (_i, [_i, _i, _i, _i]),
    "mdk_gru_debug_read": (_i, [_vp, ctypes.POINTER(ctypes.c_ulonglong), _i]),
}

_lib = None


class EngineError(RuntimeError):
    """A C-ABI call failed (reference convention: exceptions in Python)."""


def load():
    """dlopen libmedaka_amd.so and type every ABI entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X engine is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7; import it first so that one HIP runtime is shared
    # (device pointers and streams of torch tensors are then valid inside the engine).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in ABI.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    for name, (restype, argtypes) in DEBUG_ABI.items():       # the debug library's hooks, if this is one (MDK_LIB=.../libmedaka_amd_debug.so)
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    _lib = lib
    return lib


def is_debug_library():
    """True if the loaded library was built with -DMDK_DEBUG_HOOKS (test / profiling hooks present)."""
    return hasattr(load(), "mdk_selftest_hold")


def last_error():
    msg = load().mdk_last_error()
    return msg.decode() if msg else ""


def check(rc, what):
    if rc != MDK_OK:
        kind = {MDK_ERR_ARG: "bad argument", MDK_ERR_DEVICE: "device error",
                MDK_ERR_OOM: "out of device memory"}.get(rc, f"error {rc}")
        raise EngineError(f"{what}: {kind}: {last_error()}")


def device_count():
    n = ctypes.c_int(0)
    rc = load().mdk_device_count(ctypes.byref(n))
    return n.value if rc == MDK_OK else 0


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    check(load().mdk_device_name(device, buf, 256), "mdk_device_name")
    return buf.value.decode()
