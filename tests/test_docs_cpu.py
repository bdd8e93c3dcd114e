"""The documents cite scripts, profiles, tests and sources by path; a path that no longer exists is a stale claim."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "scripts/README.md", "profiles/r04_experiments.md", "profiles/r06_experiments.md"]


def test_every_cited_path_exists():
    missing = []
    for d in DOCS:
        txt = open(os.path.join(ROOT, d)).read()
        for m in re.finditer(r"`([^`\s]+)`", txt):
            p = m.group(1)
            if not re.match(r"^(scripts|profiles|tests|scanner_amd|oracle|include)/", p):
                continue
            if any(c in p for c in "<>…*{}$"):  # patterns and placeholders
                continue
            q = p.split("::")[0].rstrip(".,;:)")
            if q.startswith("oracle/_ref") or q.endswith(".so") or os.path.basename(q) in ("abi_bench", "scan_synth"):
                continue                        # built artefacts (git-ignored)
            if not os.path.exists(os.path.join(ROOT, q)):
                missing.append((d, q))
    assert not missing, missing


def test_design_document_stays_reviewable():
    """DESIGN.md is the current design and the current numbers only -- at most 40 KB, lines of at most 120
    characters; the rounds before live in HISTORY.md."""
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(txt.encode()) <= 40 * 1024
    assert max(len(l) for l in txt.split("\n")) <= 120
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))
