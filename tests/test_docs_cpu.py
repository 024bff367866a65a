"""The documents cite evidence by path: every profiles/, tools/, tests/, csrc/ ... file named in DESIGN.md, README.md,
INTEGRATION.md, tools/README.md and oracle/README.md must exist in the tree."""
import itertools
import os
import re

from conftest import ROOT

DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("tools", "README.md"), os.path.join("oracle", "README.md"),
        os.path.join("docs", "LOG_r04.md"), os.path.join("docs", "LOG_r05.md"), os.path.join("docs", "LOG_r06.md"), os.path.join("docs", "PERF_MODEL.md"),
        os.path.join("tests", "README.md")]
PREFIXES = ("profiles/", "tools/", "tests/", "include/", "examples/", "oracle/", "dsabeamformer_amd/", "csrc/")


def expand(path):
    """a{b,c}d -> abd, acd (one or several brace groups)."""
    parts = re.split(r"\{([^{}]*)\}", path)
    choices = [p.split(",") if i % 2 else [p] for i, p in enumerate(parts)]
    return ["".join(c) for c in itertools.product(*choices)]


def candidates(doc_dir, path):
    if path.startswith("csrc/"):
        path = "dsabeamformer_amd/" + path
    yield os.path.join(ROOT, path)
    yield os.path.join(ROOT, doc_dir, path)


def test_every_cited_file_exists():
    missing = []
    names = set(os.listdir(os.path.join(ROOT, "profiles")))
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for quoted in re.findall(r"`([^`\n]+)`", text):
            token = quoted.split("::")[0].split(" ")[0].strip("().,;:")
            if not token.startswith(PREFIXES) and not re.match(r"r0[12]_[\w{},.*-]+\.(txt|json|csv)$", token):
                continue
            if "*" in token or "<" in token or "…" in token or token.endswith("/"):
                continue
            if re.match(r"r0[12]_", token):                       # bare profile name: lives under profiles/
                token = "profiles/" + token
            for path in expand(token):
                path = re.sub(r":[\w-]+$", "", path)                 # file:line and file:key citations
                if path.rstrip("/") == "oracle/_ref":               # named only to say that it does not exist
                    continue
                if path.startswith("profiles/"):
                    ok = os.path.basename(path) in names
                else:
                    ok = any(os.path.exists(c) for c in candidates(os.path.dirname(doc), path))
                if not ok:
                    missing.append((doc, quoted))
    assert not missing, missing


def test_design_md_stays_readable():
    """DESIGN.md is the design (<= 300 lines, no table cell of kilobytes); the per-round narratives live
    in docs/LOG_r0N.md."""
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().splitlines()
    assert len(lines) <= 300, len(lines)
    for l in lines:
        if l.startswith("|"):
            for cell in l.strip("|").split("|"):
                assert len(cell) <= 420, cell[:80]
    for n in (1, 2, 3, 4, 5, 6):
        assert os.path.exists(os.path.join(ROOT, "docs", "LOG_r0%d.md" % n))
