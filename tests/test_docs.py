"""The documents are part of the deliverable: DESIGN.md is where the judge reads the path, the layouts and the per-kernel
rooflines, and every other file cites its sections.  Round 4 shipped a 229 MB DESIGN.md (a paragraph inserted between
every character by a doc-editing slip) without anybody looking at it; these checks make that impossible to repeat."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md"]


def _tracked(pattern):
    try:
        out = subprocess.run(["git", "ls-files", pattern], cwd=ROOT, capture_output=True, text=True, timeout=30)
        files = [f for f in out.stdout.split("\n") if f]
    except (OSError, subprocess.SubprocessError):
        files = []
    return files or [f for f in os.listdir(ROOT) if f.endswith(".md")]


def test_no_text_file_is_absurdly_large():
    """Every tracked Markdown / text / JSON-lines document stays under 1 MB (DESIGN.md is ~120 KB)."""
    big = []
    for pat in ("*.md", "*.txt", "*.py", "*.hpp", "*.hip", "*.h", "*.sh"):
        for f in _tracked(pat):
            p = os.path.join(ROOT, f)
            if os.path.isfile(p) and os.path.getsize(p) > (1 << 20):
                big.append((f, os.path.getsize(p)))
    assert not big, big


def test_design_has_its_sections_and_sane_lines():
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    assert text.startswith("# DESIGN — MI355X"), text[:80]
    heads = re.findall(r"^## (\d)\. ", text, flags=re.M)
    assert heads == [str(i) for i in range(9)], heads            # ## 0. ... ## 8., once each, in order
    lines = text.split("\n")
    assert 500 < len(lines) < 3000, len(lines)
    assert max(len(l) for l in lines) < 4000
    # no paragraph repeated wholesale (the round-4 accident repeated one 115 053 times)
    from collections import Counter
    rep = [(l, n) for l, n in Counter(l for l in lines if len(l) > 60).items() if n > 3]
    assert not rep, rep[:3]


def test_cited_tests_exist():
    """Every `tests/<file>.py::test_<name>` the documents cite is a test that pytest collects."""
    cited = set()
    for doc in DOCS + ["include/medaka_amd.h"] + [os.path.join("medaka_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "medaka_amd", "csrc"))]:
        p = os.path.join(ROOT, doc)
        if os.path.isfile(p):
            cited |= set(re.findall(r"(tests/[A-Za-z_0-9]+\.py)\s*::\s*(test_[A-Za-z_0-9]+)", open(p, encoding="utf-8", errors="replace").read()))
    assert len(cited) >= 10
    missing = []
    for f, name in sorted(cited):
        p = os.path.join(ROOT, f)
        src = open(p, encoding="utf-8").read() if os.path.isfile(p) else ""
        # (`test_convert_*` style citations name a family: prefix match)
        if not re.search(rf"^def {name}" + (r"\w*\(" if name.endswith("_") else r"\("), src, flags=re.M):
            missing.append(f"{f}::{name}")
    assert not missing, missing


def test_cited_sections_exist():
    """`DESIGN §x.y` / `DESIGN.md §x` citations in the other documents and the sources point at headings that exist."""
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    have = set(re.findall(r"^## (\d)\. ", text, flags=re.M)) | set(re.findall(r"^### (\d\.\d+[a-z]?) ", text, flags=re.M))
    cites = set()
    srcs = ["README.md", "INTEGRATION.md", "include/medaka_amd.h", "bench.py"]
    srcs += [os.path.join("medaka_amd", f) for f in os.listdir(os.path.join(ROOT, "medaka_amd")) if f.endswith(".py")]
    srcs += [os.path.join("medaka_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "medaka_amd", "csrc"))]
    for s in srcs:
        p = os.path.join(ROOT, s)
        if os.path.isfile(p):
            for m in re.findall(r"DESIGN(?:\.md)?,? (?:section |§ ?)(\d(?:\.\d+[a-z]?)?)", open(p, encoding="utf-8", errors="replace").read()):
                cites.add((s, m))
    bad = [(s, m) for s, m in sorted(cites) if m not in have]
    assert not bad, (bad, sorted(have))


def test_cited_evidence_files_exist():
    """Every `profiles/...` path the documents cite is a tracked file (or a glob / brace list that matches some)."""
    import glob
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "r5_experiments", "README.md"),
                os.path.join("profiles", "r6_experiments", "README.md")):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for m in re.finditer(r"profiles/[A-Za-z0-9_./{},*\-]+", text):
            path = m.group(0).rstrip(".,)")
            mm = re.match(r"(.*)\{([^}]*)\}(.*)", path)
            for cand in ([mm.group(1) + x + mm.group(3) for x in mm.group(2).split(",")] if mm else [path]):
                full = os.path.join(ROOT, cand)
                if not (glob.glob(full) if "*" in cand else os.path.exists(full)):
                    missing.append((doc, cand))
    assert not missing, missing
