"""Build the HIP engine in-tree: medaka_amd/csrc/*.hip -> medaka_amd/libmedaka_amd.so (gfx950).

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("MDK_LIB_OUT") or os.path.join(HERE, "libmedaka_amd.so")
# the same sources with -DMDK_DEBUG_HOOKS: test / profiling hooks (include/medaka_amd.h, last block) that the release library does not carry
LIB_DEBUG = os.path.join(HERE, "libmedaka_amd_debug.so")
SOURCES = ["api.hip", "rl_api.hip"]
HEADERS = ["common.hpp", "layout.hpp", "host_common.hpp", "rec_mfma.hpp", "rec_fused.hpp", "gi_proj.hpp", "head.hpp", "exact.hpp",
           "rl_front.hpp", "lstm_wide.hpp", "scan_split.hpp", "gru_model.hpp", "gru_pass.hpp", "gru_split.hpp", "gru_entries.hpp",
           os.path.join("..", "..", "include", "medaka_amd.h")]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), debug=False):
    """Compile for gfx950.  Returns the path of the shared library (`debug`: the library with the test hooks)."""
    LIB = LIB_DEBUG if debug else globals()["LIB"]
    if debug:
        extra_flags = tuple(extra_flags) + ("-DMDK_DEBUG_HOOKS",)
    if not force and not needs_build(LIB):
        return LIB
    # several ranks may get here at once (bench.py under torch.distributed.run): each compiles into
    # its own temporary and renames it into place atomically, so nobody ever loads a partial file
    tmp = f"{LIB}.tmp.{os.getpid()}"
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-ffp-contract=off", "-pthread",
           "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES] + list(extra_flags)
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def build_all(force=False):
    """Release and debug library side by side (two hipcc runs in parallel)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        jobs = [ex.submit(build, force, False, (), dbg) for dbg in (False, True)]
        return [j.result() for j in jobs]


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-") and a != "--debug"]
    print(build(force=True, verbose=True, extra_flags=flags, debug="--debug" in sys.argv[1:]))
