"""The oracle's plain-C restatements (oracle/c_ref: every primitive op, the two composed networks in float64, the OpenCV contour
chain) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer and put under the tests that pin them -- op by op against torch, the
composed networks against the torch module trees (the contour oracle runs instrumented in tests/test_host_sanitizers.py).  The
oracle is what every parity claim rests on; this checks that it is not right by accident of what lies next to its buffers.
(Runs the nested pytest in a child process with the sanitizer runtimes preloaded: they must be the first libraries of the process.)"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CREF = ROOT / "oracle" / "c_ref"


def _runtime(name: str) -> str | None:
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True)
    path = out.stdout.strip()
    return path if out.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_c_code_is_clean_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    for unit in ("ops", "nets", "contours"):                             # the committed sources, the Makefile's floating-point flags
        out = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-fopenmp",
                              "-fPIC", "-shared", "-o", str(tmp_path / f"lib{unit}_ref.so"), str(CREF / f"{unit}_ref.c"), "-lm"],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ, CV_ORACLE_CREF_DIR=str(tmp_path), LD_PRELOAD=f"{asan}:{ubsan}",
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")
    # the bindings really load the instrumented builds
    probe = subprocess.run([sys.executable, "-c",
                            "import sys; sys.path.insert(0, '.');\n"
                            "from oracle import nets_c, contours_c\n"
                            "nets_c.library(); contours_c.library()\n"
                            "print(open('/proc/self/maps').read())"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0, probe.stderr[-2000:]
    assert str(tmp_path / "libnets_ref.so") in probe.stdout and str(tmp_path / "libcontours_ref.so") in probe.stdout
    assert str(CREF / "libnets_ref.so") not in probe.stdout
    # (the contour oracle runs instrumented, against the product on 3680 masks, in tests/test_host_sanitizers.py; the two slowest network
    # tests -- the second UNet variant and the mutation test -- are left to the uninstrumented run to keep the CPU suite short)
    run = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_ops.py", "tests/test_oracle_nets_c.py", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          "-k", "not swapped_concatenation and not composition[True]"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert " passed" in run.stdout and "failed" not in run.stdout, tail
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
