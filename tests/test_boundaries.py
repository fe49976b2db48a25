"""Repository boundaries the scope contract states: the oracle is test infrastructure, the
product never routes through it, and nothing that runs on the GPU box reads the reference checkout."""
from __future__ import annotations

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ORACLE_IMPORT = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)


def _py_files(*dirs):
    for d in dirs:
        yield from sorted((ROOT / d).rglob("*.py"))


def test_product_and_tools_never_import_the_oracle():
    offenders = [str(f.relative_to(ROOT)) for f in _py_files("chessvision-3lc_amd", "tools")
                 if ORACLE_IMPORT.search(f.read_text())]
    assert offenders == [], offenders


def test_bench_touches_the_oracle_only_in_the_cpu_baseline_leg():
    src = (ROOT / "bench.py").read_text()
    hits = [m.start() for m in ORACLE_IMPORT.finditer(src)]
    body_start = src.index("def cpu_baseline(")
    body_end = src.index("\ndef ", body_start + 1)
    assert hits and all(body_start < h < body_end for h in hits), hits


def test_nothing_that_runs_on_the_gpu_box_reads_the_reference_checkout():
    runtime = list(_py_files("chessvision-3lc_amd", "tools", "oracle")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]
    runtime += [f for f in _py_files("tests") if f.name not in ("make_golden.py", "make_photos.py", Path(__file__).name)]
    offenders = []
    for f in runtime:
        text = f.read_text()
        for m in re.finditer(r"/root/reference", text):
            line = text[text.rfind("\n", 0, m.start()) + 1: text.find("\n", m.end())]
            if "open(" in line or "Path(" in line or "sys.path" in line or "import" in line:
                offenders.append(f"{f.relative_to(ROOT)}: {line.strip()}")
    assert offenders == [], offenders
