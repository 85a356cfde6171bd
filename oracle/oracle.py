"""Python face of the CPU oracle (test infrastructure, never imported by medaka_amd/).

Two restatements of the reference path `TorchModel.predict_on_batch` ->
`GRUModel.forward` (reference medaka/models.py:303-313, medaka/architectures/gru.py:46-72):

* `c_gru_forward`      -- plain C fp32 (oracle/gru_oracle.c) through ctypes;
* `TorchOracleGRU`     -- the same three PyTorch calls the reference makes
                          (nn.GRU -> nn.Linear -> softmax) on PyTorch-CPU.  Because the
                          reference's arithmetic for this path lives in PyTorch itself, this is
                          bit-for-bit the reference's CPU result and is what `bench.py` times as
                          the host baseline (`cpu_baseline.kind = "port"`).

Both are pinned against outputs of the UNMODIFIED reference classes
(oracle/make_golden.py -> tests/golden/*.npz, checked in tests/test_oracle.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# torch state_dict key order of reference GRUModel (gru.py:46-55)
def state_keys(n_layers=2, bidirectional=True):
    keys = []
    for layer in range(n_layers):
        for suffix in ([""] + (["_reverse"] if bidirectional else [])):
            for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                keys.append(f"gru.{name}_l{layer}{suffix}")
    keys += ["linear.weight", "linear.bias"]
    return keys


def build(force=False):
    """Compile oracle/gru_oracle.c -> oracle/libmdk_oracle.so (gcc, seconds)."""
    so = os.path.join(_HERE, "libmdk_oracle.so")
    src = os.path.join(_HERE, "gru_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmdk_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.mdk_oracle_gru_forward.restype = ctypes.c_int
        _LIB.mdk_oracle_majority_forward.restype = ctypes.c_int
        _LIB.mdk_oracle_num_threads.restype = ctypes.c_int
    return _LIB


def c_num_threads():
    return int(_lib().mdk_oracle_num_threads())


def c_gru_forward(x, state, gru_size=128, n_layers=2, bidirectional=True, num_classes=5,
                  normalise=True):
    """x: (B,T,I) float32 array; state: mapping name -> array (torch state_dict names)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T, I = x.shape
    keys = state_keys(n_layers, bidirectional)
    arrs = [np.ascontiguousarray(np.asarray(state[k]), dtype=np.float32) for k in keys]
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    out = np.empty((B, T, num_classes), dtype=np.float32)
    rc = _lib().mdk_oracle_gru_forward(
        ctypes.c_void_p(x.ctypes.data), B, T, I, gru_size, n_layers, int(bidirectional),
        num_classes, ptrs, int(normalise), ctypes.c_void_p(out.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"mdk_oracle_gru_forward failed rc={rc}")
    return out


def c_majority_forward(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape[:-1] + (5,), dtype=np.float32)
    _lib().mdk_oracle_majority_forward(
        ctypes.c_void_p(x.ctypes.data), ctypes.c_long(x.size // 10),
        ctypes.c_void_p(out.ctypes.data))
    return out


def make_torch_oracle(state=None, num_features=10, gru_size=128, n_layers=2,
                      bidirectional=True, seed=0):
    """Build the PyTorch-CPU restatement of reference GRUModel (gru.py:46-72)."""
    import torch

    class TorchOracleGRU(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gru = torch.nn.GRU(num_features, gru_size, num_layers=n_layers,
                                    bidirectional=bidirectional, batch_first=True)
            self.linear = torch.nn.Linear(2 * gru_size if bidirectional else gru_size, 5)
            self.normalise = True

        def forward(self, x):
            x = self.gru(x)[0]
            x = self.linear(x)
            if self.normalise:
                x = torch.softmax(x, dim=-1)
            return x

        def predict(self, x):
            """models.py:303-313 on CPU: inference_mode, fp32, returns cpu tensor."""
            with torch.inference_mode():
                return self.forward(torch.as_tensor(x, dtype=torch.float32)).detach().cpu()

    torch.manual_seed(seed)
    m = TorchOracleGRU().eval()
    if state is not None:
        m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in state.items()})
    return m


def state_to_numpy(state_dict):
    return {k: v.detach().cpu().numpy().copy() for k, v in state_dict.items()}


def normalise_counts(counts, depth):
    """CountsFeatureEncoder(normalise='total') -- reference medaka/features.py:907-911 and :926:
    integer counts / max(1, depth of the parent major column), numpy true division (float64),
    then `.astype(float32)`.  counts (..., F) integer, depth (...) integer."""
    counts = np.asarray(counts).astype(np.uint64)
    depth = np.asarray(depth).astype(np.uint64)
    return (counts / np.maximum(1, depth)[..., None]).astype(np.float32)


def decode_consensus(label_probs, symbols="*ACGT", with_gaps=False, with_qualities=False, cap=70.0):
    """HaploidLabelScheme.decode_consensus for one sample -- reference medaka/labels.py:1053-1085
    with `_phred` (labels.py:388-402): argmax, probability of the argmax class, gaps dropped,
    qualities = chr(uint8(min(-10 log10(clip(1 - p, 10^-7, 1)), 70)) + 33)."""
    mp = np.argmax(label_probs, -1)
    probs = np.take_along_axis(label_probs, np.expand_dims(mp, -1), -1).squeeze(-1)
    if not with_gaps:
        mask = mp != symbols.index("*")
        mp, probs = mp[mask], probs[mask]
    seq = np.array([ord(x) for x in symbols], dtype="u1")[mp].tobytes().decode()
    if not with_qualities:
        return seq
    err = np.clip(1 - probs, 10 ** (-cap / 10.0), 1)
    q = np.minimum(-10 * np.log10(err), cap)
    return seq, (q.astype("u1") + 33).tobytes().decode()
