"""ctypes binding of ``oracle/c_ref/libcontours_ref.so`` -- the reference's mask -> quadrangle chain (``chessvision/core.py:357-411``:
findContours RETR_CCOMP + CHAIN_APPROX_TC89_KCOS, contourArea / boundingRect filter, arcLength, approxPolyDP, rotation) restated
literally in plain C.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  ``contours_ref.c`` scans and relabels the padded image as Suzuki-Abe's
Algorithm 1 publishes it and shares no code with the product's ``csrc/contour.cpp`` (run-based components) or
``chessvision/classical.py`` (scipy labels)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

# CV_ORACLE_CREF_DIR: load the libraries from another directory (tests/test_oracle_sanitizers.py: builds with AddressSanitizer + UBSan)
_DIR = Path(os.environ.get("CV_ORACLE_CREF_DIR") or Path(__file__).resolve().parent / "c_ref")
_lib = None

NONE, TC89_KCOS = 0, 1


def library():
    global _lib
    if _lib is None:
        so = _DIR / "libcontours_ref.so"
        src = _DIR / "contours_ref.c"                              # (absent in a directory of prebuilt sanitizer libraries)
        if src.exists() and (not so.exists() or so.stat().st_mtime < src.stat().st_mtime):
            subprocess.run(["make", "libcontours_ref.so"], cwd=_DIR, check=True, stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(str(so))
        lib.ref_arc_length_closed.restype = ctypes.c_double
        lib.ref_contour_area.restype = ctypes.c_double
        _lib = lib
    return _lib


def _u8(mask):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    assert m.ndim == 2
    return m, m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def find_contours(mask: np.ndarray, method: int = TC89_KCOS):
    """cv2.findContours(mask, RETR_CCOMP, method)[0] plus the hole flag of every contour: list of (n,1,2) int32, list of bool."""
    m, mp = _u8(mask)
    h, w = m.shape
    cap_pts, cap_c = 4 * h * w + 16, h * w + 16
    xy = np.zeros((cap_pts, 2), dtype=np.int32)
    counts = np.zeros(cap_c, dtype=np.int32)
    holes = np.zeros(cap_c, dtype=np.int32)
    n = library().ref_find_contours(mp, h, w, int(method), _ip(xy), cap_pts, _ip(counts), _ip(holes), cap_c)
    if n < 0:
        raise RuntimeError(f"contours_ref.c: ref_find_contours failed ({n})")
    out, at = [], 0
    for k in range(n):
        out.append(xy[at:at + counts[k]].reshape(-1, 1, 2).copy())
        at += int(counts[k])
    return out, [bool(v) for v in holes[:n]]


def find_quadrangle(mask: np.ndarray):
    """``ChessVision._find_quadrangle`` of the reference: (4,1,2) int32 or None."""
    m, mp = _u8(mask)
    quad = np.zeros(8, dtype=np.int32)
    rc = library().ref_find_quadrangle(mp, m.shape[0], m.shape[1], _ip(quad))
    if rc < 0:
        raise RuntimeError(f"contours_ref.c: ref_find_quadrangle failed ({rc})")
    return quad.reshape(4, 1, 2) if rc == 1 else None


def arc_length(contour: np.ndarray) -> float:
    p = np.ascontiguousarray(np.asarray(contour).reshape(-1, 2), dtype=np.int32)
    return float(library().ref_arc_length_closed(_ip(p), len(p)))


def contour_area(contour: np.ndarray) -> float:
    p = np.ascontiguousarray(np.asarray(contour).reshape(-1, 2), dtype=np.int32)
    return float(library().ref_contour_area(_ip(p), len(p)))


def approx_poly_dp(contour: np.ndarray, epsilon: float) -> np.ndarray:
    p = np.ascontiguousarray(np.asarray(contour).reshape(-1, 2), dtype=np.int32)
    out = np.zeros_like(p)
    n = library().ref_approx_poly_dp_closed(_ip(p), len(p), ctypes.c_double(float(epsilon)), _ip(out))
    return out[:n].reshape(-1, 1, 2).copy()
