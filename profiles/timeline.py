"""Kernel / copy timeline of the LAST call of each configuration in a rocprofv3 trace of profiles/host_trace.py.
   python profiles/timeline.py RESULTS.db [OUT.txt] [CALLS]
A call ends with its k_split_verify dispatch; everything between two of those is one call.  CALLS: comma-separated call
indices to print instead (negative: from the end), e.g. "-6,-2" for a trace of bench.py."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^mdk::", "", name)
    return name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    for r in db.execute(f"select name, start, end, {qcol or '0'}, grid_x from kernels"):
        ev.append((r[1], r[2], "K", short(r[0]), r[3], r[4]))
    mt = next((t for t in tables if t in ("memory_copies", "memory_copy")), None)
    if mt:
        mc = [r[1] for r in db.execute(f"pragma table_info({mt})")]
        name_c = "name" if "name" in mc else mc[0]
        size_c = "size" if "size" in mc else ("bytes" if "bytes" in mc else None)
        for r in db.execute(f"select {name_c}, start, end, {size_c or '0'} from {mt}"):
            ev.append((r[1], r[2], "C", str(r[0])[:40], 0, r[3]))
    ev.sort()
    ends = [i for i, e in enumerate(ev) if e[2] == "K" and e[3].startswith("k_split_verify")]
    lines = []
    # the last call of the first half of the verify dispatches (configuration 1) and the very last call (configuration 2)
    picks = (("configuration 1, last call", len(ends) // 2 - 1), ("configuration 2, last call", len(ends) - 1))
    if len(sys.argv) > 3:
        picks = tuple((f"call {int(c)} of {len(ends)}", int(c) % len(ends)) for c in sys.argv[3].split(","))
    for label, idx in picks:
        if idx < 1:
            continue
        lo, hi = ends[idx - 1] + 1, ends[idx]
        t0 = ev[lo][0]
        lines.append(f"== {label}: {(ev[hi][1] - t0) / 1e6:.3f} ms from the first event to the end of the certificate")
        for s, e, kind, name, q, extra in ev[lo:hi + 1]:
            if (e - s) < 3000 and kind == "K" and not name.startswith("k_split"):
                pass
            lines.append(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  {kind} q{q} {name} {extra}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
