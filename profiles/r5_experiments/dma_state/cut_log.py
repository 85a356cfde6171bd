"""Cut an AMD_LOG_LEVEL log (stderr of probe.py with PROBE_MARK=1) down to what the copies of a few calls looked like:
per call the number of log lines that mention each copy primitive, and the raw lines of two calls either side of the
moment the calls get fast."""
import collections
import json
import re
import sys

log, out = sys.argv[1], sys.argv[2]
calls, cur, name = [], None, None
for line in open(log, errors="replace"):
    if line.startswith("### begin"):
        name, cur = line.split()[2:4], []
    elif line.startswith("### end"):
        calls.append({"call": " ".join(name), "ms": float(line.split()[-1]), "lines": cur})
        cur = None
    elif cur is not None:
        cur.append(line.rstrip("\n"))
keys = ["hipMemcpy2DAsync", "hipMemcpyAsync", "hipMemcpyAsync (", "Rect", "rect", "SDMA", "sdma", "Blit", "blit", "KernelBlit", "shader", "Shader",
        "hsa_amd_memory_async_copy", "copy_on_engine", "engine", "ShaderName", "hipLaunchKernel", "hipModuleLaunchKernel", "hipExtLaunch", "barrier", "Barrier"]
summary = []
for c in calls:
    cnt = collections.Counter()
    for l in c["lines"]:
        for k in keys:
            if k in l:
                cnt[k] += 1
    summary.append({"call": c["call"], "ms": c["ms"], "n_lines": len(c["lines"]), **cnt})
slow = [i for i, c in enumerate(calls) if c["ms"] > 9.5]
edge = slow[-1] if slow else len(calls) // 2
raw = {c["call"] + f" ({c['ms']} ms)": c["lines"][:1500] for c in calls[max(0, edge - 1):edge + 3]}
json.dump({"summary": summary, "edge_call_index": edge, "raw_around_the_edge": raw}, open(out, "w"), indent=0)
