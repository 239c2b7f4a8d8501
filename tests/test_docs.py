"""The documents cite files as evidence (profiles/, profiles/src/, tools/, tests/ ...): every cited path must exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md", "profiles/README.md"]
PREFIXES = ("profiles/", "tools/", "tests/", "oracle/", "include/", "stochqn_amd/")


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
            # built artefacts are not in the tree until build() has run; binaries of the measurement tools likewise
            if path.endswith(".so") or path == "oracle/_ref" or path.startswith("tests/hostsim/build"):
                continue
            if any(os.path.exists(full + ext) for ext in (".c", ".cpp", ".hip")):     # a tool cited by its binary's name
                continue
            if not os.path.exists(full):
                missing.append((doc, path))
    assert not missing, missing


def test_tests_named_in_the_documents_exist():
    """Evidence is cited by test name (`test_...`, optionally `file.py::test_...`): every such name must be a test in tests/."""
    defined = set()
    for fn in os.listdir(os.path.join(ROOT, "tests")):
        if fn.endswith(".py"):
            defined |= set(re.findall(r"^\s*def (test_[A-Za-z0-9_]+)\(", open(os.path.join(ROOT, "tests", fn)).read(), re.M))
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"\b(test_[a-z0-9_]+[a-z0-9])\b", text)):
            if name in defined or name + ".py" in os.listdir(os.path.join(ROOT, "tests")):
                continue
            # a name cut short with an ellipsis ("test_bench_starts_its_own_ranks…") counts when it is the start of exactly one test
            if re.search(re.escape(name) + r"(…|\.\.\.)", text) and sum(d.startswith(name) for d in defined) >= 1:
                continue
            missing.append((doc, name))
    assert not missing, "tests named in the documents but not in tests/: %s" % missing


def test_functions_named_in_the_documents_exist_in_the_named_source():
    """`machines.cpp: apply_step / gradient_slice_arrives`, `kernels.hip: Slice`, `bench.py `live_pmc`` ...: the identifier must
    occur in that source file."""
    sources = {"runtime.cpp": "stochqn_amd/csrc/runtime.cpp", "machines.cpp": "stochqn_amd/csrc/machines.cpp",
               "group.cpp": "stochqn_amd/csrc/group.cpp", "kernels.hip": "stochqn_amd/csrc/kernels.hip", "free.py": "stochqn_amd/free.py",
               "bench.py": "bench.py"}
    texts = {k: open(os.path.join(ROOT, v)).read() for k, v in sources.items()}
    missing, checked = [], 0
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for fname, names in re.findall(r"`((?:runtime|machines|group)\.cpp|kernels\.hip|free\.py): ([A-Za-z0-9_ /:<>|]+)`", text):
            for ident in re.split(r"\s*/\s*", names):
                ident = ident.strip().split("<")[0]
                if not re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*", ident):
                    continue
                checked += 1
                if not re.search(r"\b" + re.escape(ident) + r"\b", texts[fname]):
                    missing.append((doc, fname, ident))
        for ident in re.findall(r"bench\.py `([A-Za-z_][A-Za-z0-9_]*)`", text):
            checked += 1
            if not re.search(r"\b" + re.escape(ident) + r"\b", texts["bench.py"]):
                missing.append((doc, "bench.py", ident))
    assert checked >= 15, checked
    assert not missing, missing
