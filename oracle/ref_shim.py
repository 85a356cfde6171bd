"""Import shim for the UNMODIFIED reference (test infrastructure only).

The reference package `medaka` (at /root/reference) cannot be imported as-is in
this container: its CFFI extension `libmedaka` and several third-party modules
(pysam, h5py, toml, intervaltree, ...) are not installed.  None of them are on
the consensus-inference hot path, so this module pre-seeds
`sys.modules` with inert stubs and then imports the reference modules from
where they lie.  Nothing is copied.

ONLY `tests/`, `oracle/make_golden.py` and ad-hoc validation scripts may import
this file; it needs /root/reference, which does not exist on the GPU box.
The product (`medaka_amd/`) never imports it.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MEDAKA_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "pysam", "h5py", "toml", "intervaltree", "ont_fast5_api",
    "ont_fast5_api.fast5_interface", "edlib", "parasail", "mappy",
    "wurlitzer", "spoa", "pyabpoa", "requests", "tqdm",
]


class _Anything(types.ModuleType):
    """Module stub: any attribute resolves to a dummy class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (), {})
        setattr(self, name, obj)
        return obj


def available():
    """True when the reference tree is present (this container, not the GPU box)."""
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "medaka", "architectures"))


def install():
    """Make `import medaka.architectures` work against the reference tree."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if "medaka" in sys.modules and getattr(sys.modules["medaka"], "__file__", "").startswith(REFERENCE_ROOT):
        return
    for name in _STUBS:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    # libmedaka: only plp_bases/featlen (+ a few ints) are read at import time
    # (medaka/common.py:29-35 in the reference).
    if "libmedaka" not in sys.modules:
        lm = types.ModuleType("libmedaka")
        plp = b"acgtACGTdD"

        class _Lib:
            plp_bases = plp
            featlen = 10
            fwd_del = 9
            rev_del = 8
            base_featlen = 4

        class _FFI:
            NULL = None

            @staticmethod
            def buffer(obj, n=None):
                return bytes(obj[:n]) if n is not None else bytes(obj)

            @staticmethod
            def string(obj):      # features.py:671 reads the channel codes through ffi.string
                return bytes(obj)

            def __getattr__(self, name):
                return lambda *a, **k: None

        lm.lib = _Lib()
        lm.ffi = _FFI()
        sys.modules["libmedaka"] = lm
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_modules():
    """Return (architectures, models, torch_ext) of the unmodified reference."""
    install()
    import medaka.architectures as arch
    import medaka.models as models
    import medaka.torch_ext as torch_ext
    return arch, models, torch_ext
