"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/chessvision_hip.h declares,
and fails loudly (status + message, no abort, no CPU fallback) when there is no GPU."""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from chessvision import hip_backend

    return hip_backend.load_library()


def _declared_symbols():
    text = (ROOT / "include" / "chessvision_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from chessvision import hip_backend

    declared = _declared_symbols()
    assert len(declared) >= 20
    bound = {name for name, _, _ in hip_backend.SYMBOLS}
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/chessvision_hip.h but not exported"
        assert sym in bound, f"{sym} has no ctypes prototype in hip_backend.SYMBOLS"
    assert lib.cv_abi_version() == 1


def test_code_object_targets_gfx950_only():
    so = ROOT / "chessvision-3lc_amd" / "lib" / "libchessvision_hip.so"
    blob = so.read_bytes()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_is_a_loud_error_not_a_fallback(lib):
    from chessvision.hip_backend import HipBackendError, HipEngine

    h = ctypes.c_void_p()
    assert lib.cv_engine_create(0, 1, ctypes.byref(h)) != 0
    assert b"HIP device" in lib.cv_last_error() or b"device" in lib.cv_last_error()
    with pytest.raises(HipBackendError):
        HipEngine(precision="f16")


def test_null_handles_are_errors(lib):
    assert lib.cv_unet_forward(None, None, 1, None, None) != 0
    assert b"null engine" in lib.cv_last_error()
    assert lib.cv_engine_destroy(None) == 0
