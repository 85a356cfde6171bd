"""Where the GPU idles in the fed loop: from a rocprofv3 --kernel-trace --memory-copy-trace database of profiles/host_trace_half.py,
the busy fraction of the recurrence kernels over the last N calls and every gap > 15 us between two of them, with what ran
in the gap.   python profiles/r6_experiments/gaps.py RESULTS.db [N_CALLS]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^mdk::", "", name)
    m = re.search(r"k_[a-z_0-9]+", name)          # (mangled names: _ZN3mdk11k_rec_fusedILi8E...)
    return (m.group(0) if m else name)[:28]


db = sqlite3.connect(sys.argv[1])
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "0"
K = [(r[1], r[2], short(r[0]), r[3]) for r in db.execute(f"select name, start, end, {qcol} from kernels")]
mt = next((t for t in tables if t in ("memory_copies", "memory_copy")), None)
C = []
if mt:
    mc = [r[1] for r in db.execute(f"pragma table_info({mt})")]
    size_c = "size" if "size" in mc else ("bytes" if "bytes" in mc else "0")
    C = [(r[1], r[2], str(r[0]).replace("MEMORY_COPY_", "")[:18], r[3]) for r in db.execute(f"select name, start, end, {size_c} from {mt}")]
K.sort()
ver = [k for k in K if k[2].startswith("k_split_verify")]
rec = [k for k in K if k[2].startswith("k_rec_")]
if len(ver) > n_calls + 1:
    t_lo, t_hi = ver[-n_calls - 1][1], ver[-1][1]
else:
    t_lo, t_hi = rec[len(rec) // 2][0], rec[-1][1]
rec = [k for k in rec if k[0] >= t_lo and k[1] <= t_hi]
busy, cur_end, gaps = 0, t_lo, []
for s, e, name, q in rec:
    if s > cur_end:
        if s - cur_end > 15000:
            gaps.append((cur_end, s))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
span = t_hi - t_lo
print(f"window: {span / 1e6:.3f} ms, {n_calls} calls = {span / 1e6 / n_calls:.3f} ms per call; recurrence kernels busy {busy / span:.3f} "
      f"({busy / 1e6 / n_calls:.3f} ms per call); {len(gaps)} gaps > 15 us, {sum(b - a for a, b in gaps) / 1e6 / n_calls:.3f} ms per call")
for a, b in gaps:
    inside = [f"{n}:{(min(e, b) - max(s, a)) / 1e3:.0f}us" for s, e, n, q in K if s < b and e > a and not n.startswith("k_rec_")]
    copies = [f"{n}[{sz // 1000}K]:{(min(e, b) - max(s, a)) / 1e3:.0f}us" for s, e, n, sz in C if s < b and e > a]
    print(f"  gap {(a - t_lo) / 1e6:9.3f} -> {(b - t_lo) / 1e6:9.3f} ms  {(b - a) / 1e3:7.1f} us   kernels: {' '.join(inside[:6])}   copies: {len(copies)} {' '.join(copies[:4])}")
