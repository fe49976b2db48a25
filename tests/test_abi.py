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
    assert lib.cv_abi_version() == hip_backend.ABI_VERSION == 6


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


def test_no_cxx_exception_crosses_the_boundary(lib):
    """Every extern "C" body runs inside an exception guard: an allocation failure in host C++ code (here the contour stage asked
    for a 2^62-byte label image; operator new refuses before anything is read) comes back as CV_ERR_NOMEM with a message, and
    the process lives on."""
    import numpy as np

    mask = np.zeros((4, 4), np.uint8)
    quad = (ctypes.c_int32 * 8)()
    found = ctypes.c_int(0)
    big = 2 ** 31 - 1
    status = lib.cv_find_quadrangle(mask.ctypes.data_as(ctypes.c_void_p), big, big, quad, ctypes.byref(found))
    assert status == 4                                            # CV_ERR_NOMEM
    assert b"memory" in lib.cv_last_error()
    assert lib.cv_find_quadrangle(mask.ctypes.data_as(ctypes.c_void_p), 4, 4, quad, ctypes.byref(found)) == 0   # still alive


def test_decode_positions_equals_the_python_restatement(lib):
    """cv_decode_positions (csrc/position.cpp) against ChessVision.process_position_probabilities (core.py, the readable
    restatement of reference core.py:309-355,441-469): labels, both FENs and the pawn-rule fixes, both board orientations,
    including an exact tie between the alternatives of a back-rank pawn."""
    import numpy as np

    from chessvision import ChessVision, constants, hip_backend

    rng = np.random.default_rng(0)
    probs = rng.random((40, 64, 13)).astype(np.float32)
    probs /= probs.sum(-1, keepdims=True)
    probs[0, 3] = 0
    probs[0, 3, 3], probs[0, 3, 0], probs[0, 3, 5] = 0.5, 0.25, 0.25          # white pawn on d8, bishop and rook tie behind it
    probs[1, :, :] = 0
    probs[1, :, 12] = 1.0                                                       # an empty board
    for flip in (False, True):
        names = constants.SQUARE_NAMES_FLIPPED if flip else constants.SQUARE_NAMES_NORMAL
        fens, origs, labels, fixes = hip_backend.decode_positions(probs, flip)
        assert fens[1] == "8/8/8/8/8/8/8/8"
        for b in range(probs.shape[0]):
            ref = ChessVision.process_position_probabilities(probs[b], names, None)
            assert (ref.fen, ref.original_fen) == (fens[b], origs[b])
            want = [(b, names.index(f.square_name), constants.LABEL_INDICES[f.original_piece], constants.LABEL_INDICES[f.corrected_piece])
                    for f in ref.validation_fixes]
            assert want == [f for f in fixes if f[0] == b]
            assert [constants.LABEL_NAMES[i] for i in labels[b]] == list(_labels_after_rule(ref, names))
    empty = hip_backend.decode_positions(np.zeros((0, 64, 13), np.float32))
    assert empty[0] == [] and empty[1] == [] and empty[2].shape == (0, 64) and empty[3] == []


def _labels_after_rule(ref, names):
    """Per-square symbols of a FEN, in classifier order."""
    grid = {}
    for r, row in enumerate(ref.fen.split("/")):
        f = 0
        for ch in row:
            if ch.isdigit():
                for _ in range(int(ch)):
                    grid["abcdefgh"[f] + str(8 - r)] = "f"
                    f += 1
            else:
                grid["abcdefgh"[f] + str(8 - r)] = ch
                f += 1
    return [grid[n] for n in names]
