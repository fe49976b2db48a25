"""AddressSanitizer + UndefinedBehaviorSanitizer over the library's host-side C++ (csrc/contour.cpp, homography.cpp, position.cpp: mask ->
contours -> quadrangle, quadrangle -> homographies, probabilities -> labels / FEN / pawn rule), driven by tests/c_abi/host_sanitize.cpp over
a few thousand generated edge cases -- and, in the same binary, a differential test of every generated mask against the oracle's
independent C restatement (oracle/c_ref/contours_ref.c): contours by both approximation methods and the quadrangle, all equal.  GPU sanitizers are not available on the pool (the task's environment notes); the host code of the
path is what can be checked this way, and the reference has no counterpart ("Race detection / sanitizers: None")."""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "chessvision-3lc_amd" / "csrc"
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")


def test_host_side_cpp_is_clean_under_asan_and_ubsan(tmp_path):
    if not CLANG.exists() or not list(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.a")):
        pytest.skip("ROCm clang or its AddressSanitizer runtime is not installed")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
             "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}"]
    objs = []
    for unit in ("contour", "position", "homography"):               # the shipped sources, not copies
        obj = tmp_path / f"{unit}.o"
        out = subprocess.run([str(CLANG), *flags, "-c", str(CSRC / f"{unit}.cpp"), "-o", str(obj)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        objs.append(str(obj))
    # the oracle's independent C restatement of the contour chain, same sanitizers: every mask is a differential test as well
    ref_obj = tmp_path / "contours_ref.o"
    out = subprocess.run([str(CLANG.with_name("clang")), "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                          "-fno-omit-frame-pointer", "-ffp-contract=off", "-c", str(ROOT / "oracle" / "c_ref" / "contours_ref.c"), "-o", str(ref_obj)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    objs.append(str(ref_obj))
    exe = tmp_path / "host_sanitize"
    out = subprocess.run([str(CLANG), *flags, "-DWITH_ORACLE", str(ROOT / "tests" / "c_abi" / "host_sanitize.cpp"), *objs, "-o", str(exe)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-4000:])
    assert "host sanitizers: ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-4000:]
    counts = {ln.split(":")[0]: ln for ln in run.stdout.splitlines() if ":" in ln}
    assert int(counts["masks"].split()[1]) >= 3000 and int(counts["masks"].split()[3]) >= 500        # masks driven, quadrangles found
    assert int(counts["oracle"].split()[1]) >= 3 * 3000 and counts["oracle"].endswith("all equal")   # quadrangle + two contour sets per mask
