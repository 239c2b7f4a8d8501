"""The documents cite files as evidence (profiles/, scratch/, tools/, tests/ ...): every cited path must exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md", "profiles/README.md"]
PREFIXES = ("profiles/", "scratch/", "tools/", "tests/", "oracle/", "include/", "stochqn_amd/")


def cited_paths(text):
    for tok in re.findall(r"`([^`\n]+)`", text):
        tok = tok.split("::")[0].split(" ")[0].split(":")[0].strip()          # drop ::test names and :line ranges
        if tok.startswith(PREFIXES) and "*" not in tok and "{" not in tok and "<" not in tok and "…" not in tok:
            yield tok.rstrip(".,;:)")


def test_every_path_cited_in_the_documents_exists():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for path in cited_paths(text):
            full = os.path.join(ROOT, path)
            # built artefacts are not in the tree until build() has run; binaries of scratch tools likewise
            if path.endswith(".so") or path in ("tools/latency", "scratch/tune", "scratch/hostcost", "oracle/_ref"):
                continue
            if not os.path.exists(full):
                missing.append((doc, path))
    assert not missing, missing
