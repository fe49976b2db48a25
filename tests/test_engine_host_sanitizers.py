"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the whole library: csrc/ compiled with `--cuda-host-only` (every
.hip and .cpp unit: launch wrappers, planners, packers, the C ABI) and linked against a stand-in HIP runtime whose "device" memory is
host memory and whose launches are checked no-ops (tests/c_abi/hip_host_stub.cpp), then driven through the C ABI by
tests/c_abi/engine_host_driver.py: whole-model loads in four precisions (both UNet variants in two of them), forwards around the chunk sizes, the
single-layer entry points, `cv_process_image_v2`, profiling and calibration tables, error paths.  Kernels do not run here -- their parity
is the GPU suite's business; this is the part of the product a GPU test cannot see into: whether the code AROUND the kernels stays inside
its buffers.  GPU sanitizers are not available on the pool."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "chessvision-3lc_amd" / "csrc"
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")
UNITS = ["conv_igemm.hip", "conv_halo.hip", "pointwise.hip", "pipeline.hip", "engine.cpp", "unet.cpp", "resnet.cpp", "contour.cpp",
         "position.cpp", "homography.cpp", "cv_api.cpp"]            # the Makefile's SRCS
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]


def _run(cmd, **kw):
    out = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=1500, **kw)
    assert out.returncode == 0, (" ".join(str(c) for c in cmd[:6]), out.stderr[-3000:])
    return out


def _host_objects(tmp_path, san):
    """Every unit of the Makefile compiled host-only with the sanitizer flags `san`, the stand-in runtime, and the empty device code
    objects the host halves refer to; returns the object files to link."""
    makefile = (CSRC / "Makefile").read_text()
    assert all(u in makefile for u in UNITS) and makefile.count(".hip ") + makefile.count(".cpp ") + 1 >= len(UNITS)
    host = ["-std=c++17", "-O1", "-g", "-x", "hip", "--cuda-host-only", "--offload-arch=gfx950", "--rocm-path=/opt/rocm", *san, "-fPIC",
            "-I/opt/rocm/include", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    with ThreadPoolExecutor(max_workers=6) as pool:                    # the shipped sources, host code only
        list(pool.map(lambda u: _run([CLANG, *host, "-c", CSRC / u, "-o", tmp_path / f"{u}.o"]), UNITS))
    _run([CLANG, "-std=c++17", "-O1", "-g", *san, "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c",
          ROOT / "tests" / "c_abi" / "hip_host_stub.cpp", "-o", tmp_path / "stub.o"])
    # a host-only object still refers to the device code object of its unit by a per-unit symbol: give each an empty one
    objs = [tmp_path / f"{u}.o" for u in UNITS]
    syms = sorted({ln.split()[-1] for o in objs for ln in _run(["nm", "-u", o]).stdout.splitlines() if "__hip_fatbin" in ln})
    (tmp_path / "fatbins.c").write_text("".join(f"const char {s}[16] = {{0}};\n" for s in syms))
    _run([CLANG.with_name("clang"), "-fPIC", "-c", tmp_path / "fatbins.c", "-o", tmp_path / "fatbins.o"])
    return [*objs, tmp_path / "stub.o", tmp_path / "fatbins.o"]


def test_host_side_of_the_engine_is_clean_under_asan_and_ubsan(tmp_path):
    runtimes = sorted(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not CLANG.exists() or not runtimes:
        pytest.skip("ROCm clang or its shared AddressSanitizer runtime is not installed")
    lib = tmp_path / "libchessvision_hip_hostsan.so"
    _run([CLANG, "-shared", *SAN, "-shared-libsan", "--rtlib=compiler-rt", "-o", lib, *_host_objects(tmp_path, SAN)])
    env = dict(os.environ, CHESSVISION_HIP_LIB=str(lib), LD_PRELOAD=str(runtimes[-1]),
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")
    # negative control: the instrumentation is live in THIS library, in THIS process set-up
    bad = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({str(lib)!r}).cv_stub_selftest_overflow(); print('survived')"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "AddressSanitizer" in bad.stderr and "survived" not in bad.stdout, (bad.stdout, bad.stderr[-1500:])
    run = subprocess.run([sys.executable, str(ROOT / "tests" / "c_abi" / "engine_host_driver.py")], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=1500)
    tail = (run.stdout[-2500:], run.stderr[-4000:])
    assert run.returncode == 0, tail
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, tail
    last = run.stdout.strip().splitlines()[-1]
    assert last.startswith("engine host sanitizers: ok") and "(0 refused)" in last, tail
    assert run.stdout.count("bilinear=") == 6 and run.stdout.count("single-layer entry points: ok") == 4, tail   # 4 precisions, 2 of them with both UNet variants
    launches = int(last.split("calls, ")[1].split()[0])
    assert launches > 3000, last


def _write_blob(path, sd):                                             # the format tests/c_abi/consumer.c and engine_threads.cpp read
    import struct

    import numpy as np
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(sd)))
        for name, arr in sd.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", a.ndim))
            f.write(struct.pack("<4q", *(list(a.shape) + [0] * (4 - a.ndim))))
            f.write(a.tobytes())


def test_host_side_of_the_engine_is_race_free_under_tsan(tmp_path):
    """The round-6 soak (request threads on private and shared engine pairs, batch forwards beside them, engines that are created,
    used and destroyed meanwhile, cache trims) as a C++ program on the stand-in runtime under ThreadSanitizer: every lock, cache and
    table the threads share (engine.h: capture / legacy / load / graph mutexes, the graph graveyard, the block cache, per-engine
    mutexes, thread-local error strings) without a report.  The suite runs it small (two request threads and the batch thread on ONE engine pair, one engine
    coming and going: every lock is contended, two model loads instead of nine); the full-size run: profiles/r06_tsan_engine_threads.txt."""
    if not CLANG.exists() or not list(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.tsan-x86_64.a")):
        pytest.skip("ROCm clang or its ThreadSanitizer runtime is not installed")
    sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
    from chessvision import synthetic
    tsan = ["-fsanitize=thread", "-fno-omit-frame-pointer"]
    exe = tmp_path / "engine_threads"
    _run([CLANG, "-std=c++17", "-O1", "-g", *tsan, "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}", ROOT / "tests" / "c_abi" / "engine_threads.cpp",
          *_host_objects(tmp_path, tsan), "--rtlib=compiler-rt", "-lpthread", "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    control = subprocess.run([str(exe), "--race-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert control.returncode == 66 and "ThreadSanitizer: data race" in control.stderr       # the instrumentation is live
    _write_blob(tmp_path / "unet.blob", synthetic.unet_state_dict(1, segmenting=True))
    _write_blob(tmp_path / "resnet.blob", synthetic.resnet18_state_dict(2))
    run = subprocess.run([str(exe), str(tmp_path / "unet.blob"), str(tmp_path / "resnet.blob"), "40", "0", "1", "2"], env=env, capture_output=True,
                         text=True, timeout=1500)
    tail = (run.stdout[-1500:], run.stderr[-5000:])
    assert run.returncode == 0, tail
    assert "ThreadSanitizer" not in run.stderr, tail
    last = run.stdout.strip().splitlines()[-1]
    assert last.startswith("engine threads:") and last.endswith(" 0 failures") and "2 engines loaded" in last and "2 batch rounds" in last, tail
    assert int(last.split()[2]) >= 2 * 40, last                        # both request threads of the shared pair served their quota beside the side threads
