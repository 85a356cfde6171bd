"""Object wrapper over the C ABI (include/medaka_amd.h): one `GruEngine` = one `mdk_gru*`.

This is the layer `models.GRUModel` (the drop-in for reference
`medaka.architectures.GRUModel`) sits on; tests and bench.py also drive it directly.
"""
import ctypes

import numpy as np

from medaka_amd import lib as _lib


def state_keys(n_layers=2, bidirectional=True):
    """torch state_dict key order of reference GRUModel (medaka/architectures/gru.py:46-55)."""
    keys = []
    for layer in range(n_layers):
        for suffix in ([""] + (["_reverse"] if bidirectional else [])):
            for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                keys.append(f"gru.{name}_l{layer}{suffix}")
    keys += ["linear.weight", "linear.bias"]
    return keys


def split_plan(B, T, gpu_share=1, scan_split=1, margin=128):
    """How the engine would split a batch of B windows of T columns (include/medaka_amd.h `mdk_split_plan`; no device
    needed): {"chunks", "columns", "margin", "start": [...], "first": [...], "last": [...]}."""
    t = _lib.SplitShape()
    _lib.check(_lib.load().mdk_split_plan(int(B), int(T), int(gpu_share), int(scan_split), int(margin), ctypes.byref(t)),
               "mdk_split_plan")
    n = t.chunks
    return {"chunks": n, "columns": t.columns, "margin": t.margin, "start": list(t.start)[:n], "first": list(t.first)[:n],
            "last": list(t.last)[:n]}


def pass_plan(windows, T, num_features=10, num_layers=2, bidirectional=True, half=False, gpu_share=1, host_in=False,
              host_out=False, split_chunks=0, host_checks_range=False, lean=False, out_of_range_seen=False):
    """How a pass of `windows` windows of T columns would be launched (include/medaka_amd.h `mdk_pass_plan`; no device needed):
    work-group granularity, what is fused / streamed, and whether the gi workspace is needed."""
    desc = _lib.GruDesc(int(num_features), 128, int(num_layers), int(bool(bidirectional)), 5, 1)
    t = _lib.PassShape()
    mode = (1 if host_checks_range else 0) | (2 if lean else 0) | (4 if out_of_range_seen else 0)
    _lib.check(_lib.load().mdk_pass_plan(ctypes.byref(desc), 1 if half else 0, int(gpu_share), int(windows), int(T),
                                         (1 if host_in else 0) | (2 if host_out else 0), int(split_chunks), mode, ctypes.byref(t)),
               "mdk_pass_plan")
    return {n: (getattr(t, n) if n in ("windows_per_group", "work_groups") else bool(getattr(t, n))) for n, _ in t._fields_}


class DeviceBuffer:
    """Raw device allocation through the C ABI (for hosts without a HIP binding of their own)."""

    def __init__(self, nbytes, device=0):
        self.device = device
        self.nbytes = int(nbytes)
        self.ptr = ctypes.c_void_p()
        _lib.check(_lib.load().mdk_dev_alloc(device, max(self.nbytes, 1), ctypes.byref(self.ptr)),
                   "mdk_dev_alloc")

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _lib.check(_lib.load().mdk_memcpy_h2d(self.device, self.ptr, arr.ctypes.data, arr.nbytes),
                   "mdk_memcpy_h2d")

    def download(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _lib.check(_lib.load().mdk_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes),
                   "mdk_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            _lib.load().mdk_dev_free(self.device, self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """Page-locked host buffer from the C ABI (`mdk_host_alloc`) with a numpy view: reusable input /
    output of `GruEngine.forward_host(x, out=...)` for hosts without torch's pinned allocator.

    The memory belongs to the ctypes block `.array` is built on and is released by a finalizer of that
    block, i.e. only after the last numpy view of it is gone: views may outlive the PinnedArray object."""

    def __init__(self, shape, dtype=np.float32):
        import weakref
        ptr = ctypes.c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _lib.check(_lib.load().mdk_host_alloc(max(nbytes, 1), ctypes.byref(ptr)), "mdk_host_alloc")
        self.ptr = ptr
        block = (ctypes.c_char * max(nbytes, 1)).from_address(ptr.value)
        # np.frombuffer keeps `block` alive through the array's base chain; when the last view dies so does the block
        weakref.finalize(block, _lib.load().mdk_host_free, ctypes.c_void_p(ptr.value))
        self.array = np.frombuffer(block, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        """Drop this object's own view (the memory follows once no other view is left)."""
        self.array = None


class GruEngine:
    """MI355X bi-GRU + Linear + softmax forward (reference gru.py:58-72) behind the C ABI."""

    def __init__(self, state, num_features=10, gru_size=128, n_layers=2, bidirectional=True,
                 num_classes=5, normalise=True, device=0):
        import threading
        self._stage_lock = threading.Lock()       # stage_input (Batcher thread) against close()
        self._h = ctypes.c_void_p()
        L = _lib.load()
        keys = state_keys(n_layers, bidirectional)
        missing = [k for k in keys if k not in state]
        if missing:
            raise KeyError(f"state is missing {missing}")
        arrs = [np.ascontiguousarray(np.asarray(state[k], dtype=np.float32)) for k in keys]
        D = 2 if bidirectional else 1
        for li in range(n_layers):
            kin = num_features if li == 0 else D * gru_size
            for di in range(D):
                w_ih, w_hh, b_ih, b_hh = arrs[4 * (li * D + di):4 * (li * D + di) + 4]
                for a, shp in ((w_ih, (3 * gru_size, kin)), (w_hh, (3 * gru_size, gru_size)),
                               (b_ih, (3 * gru_size,)), (b_hh, (3 * gru_size,))):
                    if a.shape != shp:
                        raise ValueError(f"weight shape {a.shape} != expected {shp}")
        if arrs[-2].shape != (num_classes, D * gru_size) or arrs[-1].shape != (num_classes,):
            raise ValueError("linear weight/bias shape mismatch")
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        desc = _lib.GruDesc(num_features, gru_size, n_layers, int(bidirectional), num_classes,
                            int(normalise))
        _lib.check(L.mdk_gru_create(ctypes.byref(desc), ptrs, len(arrs), device,
                                    ctypes.byref(self._h)), "mdk_gru_create")
        self.num_features, self.num_classes, self.device = num_features, num_classes, device

    # -- configuration
    def set_precision(self, half):
        _lib.check(_lib.load().mdk_gru_set_precision(self._h, 1 if half else 0), "mdk_gru_set_precision")

    def set_variant(self, variant):
        """0 / False: MFMA kernels, 1 / True: exact fp32 kernels, 2: first-generation MFMA kernel."""
        _lib.check(_lib.load().mdk_gru_set_variant(self._h, int(variant)), "mdk_gru_set_variant")

    def set_normalise(self, normalise):
        _lib.check(_lib.load().mdk_gru_set_normalise(self._h, int(bool(normalise))), "mdk_gru_set_normalise")

    def set_option(self, key, value):
        _lib.check(_lib.load().mdk_gru_set_option(self._h, key.encode(), int(value)), "mdk_gru_set_option")

    def enable_timing(self, on=True):
        _lib.check(_lib.load().mdk_gru_enable_timing(self._h, int(on)), "mdk_gru_enable_timing")

    def timing(self):
        t = _lib.GruTiming()
        _lib.check(_lib.load().mdk_gru_get_timing(self._h, ctypes.byref(t)), "mdk_gru_get_timing")
        n = t.n_layers
        return {"h2d_ms": t.h2d_ms, "gi_ms": list(t.gi_ms)[:n], "rec_ms": list(t.rec_ms)[:n],
                "head_ms": t.head_ms, "d2h_ms": t.d2h_ms, "total_ms": t.total_ms,
                "rec_launches": t.rec_launches, "fused_layers": t.fused_layers, "host_streamed": t.host_streamed}

    def split(self):
        """What the last forward did about splitting the scan (include/medaka_amd.h `mdk_gru_split`)."""
        t = _lib.GruSplit()
        _lib.check(_lib.load().mdk_gru_get_split(self._h, ctypes.byref(t)), "mdk_gru_get_split")
        return {"chunks": t.chunks, "margin": t.margin, "columns": t.columns, "status": _lib.SPLIT_STATUS.get(t.status, t.status),
                "max_delta": t.max_delta, "fallbacks": t.fallbacks, "audited": bool(t.audited), "audit_max_dp": t.audit_max_dp,
                "audits": t.audits, "audit_failures": t.audit_failures, "audit_worst_dp": t.audit_worst_dp,
                "probes": t.probes, "probe_max_delta": t.probe_max_delta}

    # -- compute
    def forward_host(self, x, out=None):
        """x: (B,T,F) float32 host array -> (B,T,C) float32 host array (synchronous)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or x.shape[2] != self.num_features:
            raise ValueError(f"expected (B, T, {self.num_features}) input, got {x.shape}")
        B, T, _ = x.shape
        if out is None:
            out = np.empty((B, T, self.num_classes), dtype=np.float32)
        elif out.shape != (B, T, self.num_classes) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 (B, T, num_classes) array")
        _lib.check(_lib.load().mdk_gru_forward(self._h, x.ctypes.data, B, T, out.ctypes.data),
                   "mdk_gru_forward")
        return out

    def forward_counts_host(self, counts, depth, probs=True, decoded=False, out=None):
        """Raw pileup counts (B,T,F) uint16 + per-column depth (B,T) uint32 -> probabilities and/or
        (argmax class uint8, its probability float32): normalisation (features.py:907-911) and
        argmax decode (labels.py:1061-1065) run on the device.
        `out`: the result arrays to fill, in the order they are returned (a caller with page-locked, recycled
        buffers saves the first-touch faults of fresh ones: 10 MB of them are 0.9 ms of a 7 ms call)."""
        counts = np.asarray(counts)
        if counts.dtype.kind not in "ui":
            raise ValueError(f"pileup counts must be integers (got {counts.dtype}): normalised features go through "
                             "forward_host / predict_on_batch")
        if counts.dtype != np.uint16:
            # the reference's counts are size_t (src/medaka_counts.c); the device path carries uint16
            if counts.size and (counts.max() > 65535 or counts.min() < 0):
                raise ValueError("pileup counts beyond 65535 do not fit the uint16 device path: normalise on the "
                                 "host (predict_on_batch) for pileups this deep")
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        depth = np.ascontiguousarray(depth, dtype=np.uint32)
        if counts.ndim != 3 or counts.shape[2] != self.num_features or depth.shape != counts.shape[:2]:
            raise ValueError(f"expected counts (B, T, {self.num_features}) and depth (B, T), got {counts.shape}, {depth.shape}")
        if not (probs or decoded):
            raise ValueError("nothing requested")
        B, T, _ = counts.shape
        want = ([((B, T, self.num_classes), np.float32)] if probs else []) + \
               ([((B, T), np.uint8), ((B, T), np.float32)] if decoded else [])
        if out is None:
            out = [np.empty(shp, dtype=dt) for shp, dt in want]
        else:
            out = list(out)
            if len(out) != len(want) or any(a.shape != shp or a.dtype != dt or not a.flags.c_contiguous or not a.flags.writeable
                                             for a, (shp, dt) in zip(out, want)):
                raise ValueError(f"out must be C-contiguous writable arrays of {want}")
        p = out[0] if probs else None
        cls, pmax = (out[-2], out[-1]) if decoded else (None, None)
        _lib.check(_lib.load().mdk_gru_forward_counts(
            self._h, counts.ctypes.data, depth.ctypes.data, B, T, p.ctypes.data if probs else None,
            cls.ctypes.data if decoded else None, pmax.ctypes.data if decoded else None), "mdk_gru_forward_counts")
        return tuple(out) if decoded else p

    def forward_decoded_host(self, x):
        """x: (B,T,F) float32 -> (argmax class (B,T) uint8, its probability (B,T) float32)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or x.shape[2] != self.num_features:
            raise ValueError(f"expected (B, T, {self.num_features}) input, got {x.shape}")
        B, T, _ = x.shape
        cls, pmax = np.empty((B, T), dtype=np.uint8), np.empty((B, T), dtype=np.float32)
        _lib.check(_lib.load().mdk_gru_forward_decoded(self._h, x.ctypes.data, B, T, cls.ctypes.data,
                                                       pmax.ctypes.data), "mdk_gru_forward_decoded")
        return cls, pmax

    def stage_input(self, x_ptr, B, T):
        """Start the host -> device copy of a batch now (`mdk_gru_stage_input`); returns the token `forward_staged` takes
        (0: not staged).  Called from the loader's Batcher thread: serialised against `close()`."""
        tok = ctypes.c_ulonglong(0)
        with self._stage_lock:
            if not self._h:
                return 0
            _lib.check(_lib.load().mdk_gru_stage_input(self._h, x_ptr, int(B), int(T), ctypes.byref(tok)), "mdk_gru_stage_input")
        return tok.value

    def forward_staged(self, token, B, T, out_ptr, next_out_ptr=None):
        """The forward of a staged batch; False if the token is no longer valid (then use the ordinary host forward).
        `next_out_ptr`: the page-locked buffer the NEXT call's result will be asked into (`mdk_gru_forward_pipelined`): if the
        next batch is on the device already, its forward is started before this call waits.  The buffer has to outlive that
        (see `promise`)."""
        if next_out_ptr is None:
            rc = _lib.load().mdk_gru_forward_staged(self._h, ctypes.c_ulonglong(token), int(B), int(T), out_ptr)
        else:
            rc = _lib.load().mdk_gru_forward_pipelined(self._h, ctypes.c_ulonglong(token), int(B), int(T), out_ptr, next_out_ptr)
        if rc == _lib.MDK_ERR_ARG:
            return False
        _lib.check(rc, "mdk_gru_forward_staged")
        return True

    # The result buffer promised to the engine for the next call (it may be written to from now on): kept HERE so that it cannot
    # be released before the engine has let go of it (close / drop_pending).
    def promise(self, tensor):
        self._promised = tensor

    def take_promised(self, shape):
        """The buffer promised last time if it has this shape (the engine may have filled it already); otherwise the promise is
        withdrawn -- the engine waits for whatever it started and forgets it -- and None is returned."""
        t = getattr(self, "_promised", None)
        self._promised = None
        if t is None:
            return None
        if tuple(t.shape) == tuple(shape):
            return t
        self.drop_pending()
        return None

    def drop_pending(self):
        if self._h:
            _lib.check(_lib.load().mdk_gru_drop_pending(self._h), "mdk_gru_drop_pending")

    def forward_ptr(self, x_ptr, B, T, out_ptr, stream=None, host=False):
        """Raw-pointer forward: host pointers (`host=True`) or device pointers + hipStream_t."""
        L = _lib.load()
        if host:
            _lib.check(L.mdk_gru_forward(self._h, x_ptr, B, T, out_ptr), "mdk_gru_forward")
        else:
            _lib.check(L.mdk_gru_forward_dev(self._h, x_ptr, B, T, out_ptr, stream),
                       "mdk_gru_forward_dev")

    def close(self):
        # a Batcher thread may be inside stage_input: no hand-over targets this engine any more, and the handle is only
        # destroyed once that call has returned
        import sys
        te = sys.modules.get("medaka_amd.torch_ext")       # (imports torch: only there if somebody registered a hand-over target)
        if te is not None:
            te.forget_stage_target(self)
        with self._stage_lock:
            h, self._h = self._h, ctypes.c_void_p()
        if h:
            _lib.load().mdk_gru_destroy(h)       # (waits for a batch started ahead)
        self._promised = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rl_state_keys(bidirectional=True):
    """state_dict keys of reference LatentSpaceLSTM in the order `mdk_rl_create` expects."""
    keys = ["base_embedder.weight", "strand_embedder.weight"]
    for conv, bn in ((0, 2), (3, 5)):
        keys += [f"read_level_conv.convs.{conv}.weight", f"read_level_conv.convs.{conv}.bias"]
        keys += [f"read_level_conv.convs.{bn}.{n}" for n in ("weight", "bias", "running_mean", "running_var")]
    keys += ["pre_pool_expansion_layer.weight", "pre_pool_expansion_layer.bias"]
    if bidirectional:
        for layer in range(2):
            for sfx in ("", "_reverse"):
                keys += [f"lstm.{n}_l{layer}{sfx}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    else:
        for i in range(4):
            keys += [f"lstm.{i}.lstm.{n}_l0" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    keys += ["linear.weight", "linear.bias"]
    return keys


class RlEngine:
    """Read-level model (reference LatentSpaceLSTM.forward, latent_space_lstm.py:154-207)."""

    def __init__(self, state, use_dwells=False, bidirectional=True, lstm_size=128, cnn_size=128,
                 kernel_sizes=(1, 17), alphabet_size=6, embedding_size=6, normalise=True, device=0,
                 num_classes=5):
        self._h = ctypes.c_void_p()
        keys = rl_state_keys(bidirectional)
        missing = [k for k in keys if k not in state]
        if missing:
            raise KeyError(f"state is missing {missing}")
        arrs = [np.ascontiguousarray(np.asarray(state[k], dtype=np.float32)) for k in keys]
        if len(kernel_sizes) != 2:
            raise ValueError("the engine supports exactly two read-level conv layers")
        # every tensor against the descriptor: mdk_rl_create reads these sizes from raw pointers
        nf = embedding_size + (2 if use_dwells else 1)
        D = 2 if bidirectional else 1
        want = {"base_embedder.weight": (alphabet_size, embedding_size), "strand_embedder.weight": (3, embedding_size),
                "read_level_conv.convs.0.weight": (cnn_size, nf, int(kernel_sizes[0])),
                "read_level_conv.convs.3.weight": (cnn_size, cnn_size, int(kernel_sizes[1])),
                "pre_pool_expansion_layer.weight": (lstm_size, cnn_size), "pre_pool_expansion_layer.bias": (lstm_size,),
                "linear.weight": (num_classes, D * lstm_size), "linear.bias": (num_classes,)}
        for conv, bn in ((0, 2), (3, 5)):
            want[f"read_level_conv.convs.{conv}.bias"] = (cnn_size,)
            for n in ("weight", "bias", "running_mean", "running_var"):
                want[f"read_level_conv.convs.{bn}.{n}"] = (cnn_size,)
        for k in keys:
            if ("lstm." in k) and "weight_ih" in k:
                first = k.endswith("_l0") or k.endswith("_l0_reverse")
                want[k] = (4 * lstm_size, lstm_size if (first or not bidirectional) else D * lstm_size)
            elif "lstm." in k and "weight_hh" in k:
                want[k] = (4 * lstm_size, lstm_size)
            elif "lstm." in k:
                want[k] = (4 * lstm_size,)
        for k, a in zip(keys, arrs):
            if a.shape != want[k]:
                raise ValueError(f"{k}: shape {a.shape} != expected {want[k]}")
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        desc = _lib.RlDesc(lstm_size, cnn_size, int(kernel_sizes[0]), int(kernel_sizes[1]), int(use_dwells),
                           alphabet_size, embedding_size, int(bidirectional), int(num_classes), int(normalise))
        _lib.check(_lib.load().mdk_rl_create(ctypes.byref(desc), ptrs, len(arrs), device,
                                             ctypes.byref(self._h)), "mdk_rl_create")
        self.use_dwells, self.device = use_dwells, device

    def set_precision(self, half):
        _lib.check(_lib.load().mdk_rl_set_precision(self._h, 1 if half else 0), "mdk_rl_set_precision")

    def set_normalise(self, normalise):
        _lib.check(_lib.load().mdk_rl_set_normalise(self._h, int(bool(normalise))), "mdk_rl_set_normalise")

    def set_option(self, key, value):
        _lib.check(_lib.load().mdk_rl_set_option(self._h, key.encode(), int(value)), "mdk_rl_set_option")

    def enable_timing(self, on=True):
        _lib.check(_lib.load().mdk_rl_enable_timing(self._h, int(on)), "mdk_rl_enable_timing")

    def timing(self):
        t = _lib.RlTiming()
        _lib.check(_lib.load().mdk_rl_get_timing(self._h, ctypes.byref(t)), "mdk_rl_get_timing")
        return {"front_ms": t.front_ms, "total_ms": t.total_ms, "wide_retries": t.wide_retries}

    def check(self, stream=None):
        """Asynchronous mode ("wide_async"): raise if an earlier forward's cluster exchange timed out."""
        _lib.check(_lib.load().mdk_rl_check(self._h, stream), "mdk_rl_check")

    def forward_host(self, x):
        """x: (B, P, D, F) uint8 host array -> (B, P, 5) float32 host array."""
        x = np.ascontiguousarray(x, dtype=np.uint8)
        if x.ndim != 4:
            raise ValueError(f"expected (B, P, D, F) input, got {x.shape}")
        B, P, D, F = x.shape
        out = np.empty((B, P, 5), dtype=np.float32)
        _lib.check(_lib.load().mdk_rl_forward(self._h, x.ctypes.data, B, P, D, F, out.ctypes.data),
                   "mdk_rl_forward")
        return out

    def forward_ptr(self, x_ptr, B, P, D, F, out_ptr, stream=None, host=False):
        L = _lib.load()
        if host:
            _lib.check(L.mdk_rl_forward(self._h, x_ptr, B, P, D, F, out_ptr), "mdk_rl_forward")
        else:
            _lib.check(L.mdk_rl_forward_dev(self._h, x_ptr, B, P, D, F, out_ptr, stream), "mdk_rl_forward_dev")

    def close(self):
        if self._h:
            _lib.load().mdk_rl_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def majority_forward_host(x, device=0):
    """MajorityVoteModel.forward on the device (reference majority_vote_model.py:37-53)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.shape[-1] != 10:
        raise ValueError("majority vote expects 10 count channels")
    out = np.empty(x.shape[:-1] + (5,), dtype=np.float32)
    _lib.check(_lib.load().mdk_majority_forward(x.ctypes.data, x.size // 10, out.ctypes.data, device),
               "mdk_majority_forward")
    return out


def selftest_mfma(device=0):
    err = ctypes.c_float(-1.0)
    sub = ctypes.c_int(-1)
    _lib.check(_lib.load().mdk_selftest_mfma(device, ctypes.byref(err), ctypes.byref(sub)),
               "mdk_selftest_mfma")
    return err.value, bool(sub.value)


def decode_consensus(cls, pmax=None, symbols="*ACGT", with_gaps=False, with_qualities=False, cap=70.0):
    """Host half of the on-device decode: the reference's `HaploidLabelScheme.decode_consensus`
    (medaka/labels.py:1053-1085) continued from the device's (argmax, max-probability) pair of ONE
    sample -- gap removal, symbol lookup, and the phred string of `_phred` (labels.py:388-402)."""
    mp = np.asarray(cls).astype(np.int64)
    gap = symbols.index("*")
    mask = np.ones(mp.shape, dtype=bool) if with_gaps else (mp != gap)
    seq = np.array([ord(ch) for ch in symbols], dtype="u1")[mp[mask]].tobytes().decode()
    if not with_qualities:
        return seq
    err = np.clip(1 - np.asarray(pmax)[mask], 10 ** (-cap / 10.0), 1)
    q = np.minimum(-10 * np.log10(err), cap)
    return seq, (q.astype("u1") + 33).tobytes().decode()
