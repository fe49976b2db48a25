"""ctypes binding of ``oracle/c_ref/libnets_ref.so`` -- the two composed networks in plain C, float64 throughout.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  ``nets_ref.c`` shares no code with ``unet_ref.py`` / ``resnet_ref.py`` (torch
modules) nor with the HIP product; it pins the ORDER OF COMPOSITION (concatenation order, padding, residual placement, stem / pool
order) that per-op cross-checks cannot see.  Driven from a flat state dict (reference key names)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

# CV_ORACLE_CREF_DIR: load the libraries from another directory (tests/test_oracle_sanitizers.py: builds with AddressSanitizer + UBSan)
_DIR = Path(os.environ.get("CV_ORACLE_CREF_DIR") or Path(__file__).resolve().parent / "c_ref")
_lib = None


class _Param(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_float)), ("ndim", ctypes.c_int),
                ("shape", ctypes.c_longlong * 4)]


def library():
    global _lib
    if _lib is None:
        so = _DIR / "libnets_ref.so"
        src = _DIR / "nets_ref.c"                              # (absent in a directory of prebuilt sanitizer libraries)
        if src.exists() and (not so.exists() or so.stat().st_mtime < src.stat().st_mtime):
            subprocess.run(["make", "libnets_ref.so"], cwd=_DIR, check=True, stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(str(so))
    return _lib


def _table(state_dict):
    keep, entries = [], []
    for key, value in state_dict.items():
        if key.endswith("num_batches_tracked"):
            continue
        arr = np.ascontiguousarray(value.detach().cpu().numpy() if hasattr(value, "detach") else value, dtype=np.float32)
        keep.append(arr)
        shape = (ctypes.c_longlong * 4)(*(list(arr.shape) + [0] * (4 - arr.ndim)))
        entries.append(_Param(key.encode(), arr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), arr.ndim, shape))
    return (_Param * len(entries))(*entries), len(entries), keep


def unet_forward(state_dict, x: np.ndarray, bilinear: bool | None = None) -> np.ndarray:
    """x (N,3,H,W) float32 -> logits (N,1,H,W) float64.  ``bilinear`` defaults to what the checkpoint says (no ``up1.up.weight``)."""
    if bilinear is None:
        bilinear = "up1.up.weight" not in state_dict
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, c, h, w = x.shape
    assert c == 3
    table, count, keep = _table(state_dict)
    out = np.zeros((n, 1, h, w), dtype=np.float64)
    err = ctypes.create_string_buffer(512)
    rc = library().ref_unet_forward(table, count, int(bool(bilinear)), x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, h, w,
                                    out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), err, 512)
    del keep
    if rc:
        raise ValueError(f"nets_ref.c: {err.value.decode()}")
    return out


def resnet18_forward(state_dict, x: np.ndarray) -> np.ndarray:
    """x (N,1,H,W) float32 -> logits (N,13) float64."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, c, h, w = x.shape
    assert c == 1
    table, count, keep = _table(state_dict)
    out = np.zeros((n, 13), dtype=np.float64)
    err = ctypes.create_string_buffer(512)
    rc = library().ref_resnet18_forward(table, count, x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, h, w,
                                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), err, 512)
    del keep
    if rc:
        raise ValueError(f"nets_ref.c: {err.value.decode()}")
    return out
