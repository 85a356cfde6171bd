"""Every kernel of the library compiles for gfx950 WITHOUT scratch: a spilled register is reloaded through vector memory,
and in kernels whose loops keep counted `vmcnt` waits in flight (the recurrences, the projections) a reload is a
`vmcnt(0)` -- an L2 / HBM round trip in the inner loop.  Round 4 shipped three spilling instantiations of k_gi_gemm
unnoticed.  hipcc cross-compiles without a GPU; -Rpass-analysis=kernel-resource-usage reports every kernel."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
from medaka_amd import build


@pytest.mark.parametrize("source", build.SOURCES)
def test_no_kernel_uses_scratch(source, tmp_path):
    cmd = [build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c",
           os.path.join(build.CSRC, source), "-o", str(tmp_path / "o.o"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    name, rows, seen = None, [], 0
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            seen += 1
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) > 0:
            rows.append((name, int(m.group(1))))
    assert seen >= 10, "no resource report: did the remark flag change?"
    assert not rows, rows
