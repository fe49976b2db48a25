"""ctypes binding of ``libchessvision_hip.so`` + the model objects ChessVision plugs in.

The reference keeps two opaque callables in ``ChessVision._board_extractor`` / ``._classifier`` and only
ever does ``obj(tensor)``, ``obj.eval()``, ``obj.to(device)`` and ``hasattr(obj, "metadata")``
(reference ``chessvision/core.py:53-54,105-106,149-150,220,241``); its YOLO wrappers
(``chessvision/utils.py:205-227,257-278``) show the accepted duck type.  ``HipBoardExtractor`` and
``HipPieceClassifier`` follow that protocol and route the forward pass through the C ABI declared in
``include/chessvision_hip.h``.  PyTorch is used for device memory and streams only.

There is NO CPU fallback: if the shared library or a gfx950 device is missing, construction raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path
from typing import Mapping

import numpy as np
import torch

_LIB_NAME = "libchessvision_hip.so"
PREC_F32, PREC_F16, PREC_F16X3, PREC_F16R = 0, 1, 2, 3
ABI_VERSION = 6
_PRECISIONS = {"f32": PREC_F32, "fp32": PREC_F32, "float32": PREC_F32, "f16": PREC_F16, "fp16": PREC_F16,
               "float16": PREC_F16, "f16x3": PREC_F16X3, "split": PREC_F16X3, "f16r": PREC_F16R}
_PREC_NAMES = {PREC_F32: "f32", PREC_F16: "f16", PREC_F16X3: "f16x3", PREC_F16R: "f16r"}


CV_ERR_NUMERIC = 5                               # include/chessvision_hip.h


class HipBackendError(RuntimeError):
    """Raised for every non-zero status of the C ABI (message = ``cv_last_error()``)."""


class NumericRangeError(HipBackendError):
    """CV_ERR_NUMERIC: a layer of an f16-based engine stored a non-finite value (an activation left the range the engine can hold, or
    a NaN reached it); the results of the call are invalid.  ``layer`` names the first such layer.  ``ChessVision.process_image`` /
    ``process_images`` / ``extract_board`` / ``classify_position`` catch it and repeat the request on an exact-f32 engine."""

    def __init__(self, message: str):
        super().__init__(message)
        self.layer = message.split("produced by '", 1)[1].split("'", 1)[0] if "produced by '" in message else "input tensor"


class _ImageResult(ctypes.Structure):             # mirrors cv_image_result_t (include/chessvision_hip.h); the pointer fields as plain
    _fields_ = [("logits", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("quadrangle", ctypes.c_float * 8),     # addresses: assigning
                ("found", ctypes.c_int32), ("board", ctypes.c_void_p), ("probabilities", ctypes.c_void_p),      # ndarray.ctypes.data costs
                ("labels", ctypes.c_void_p), ("fen", ctypes.c_char * 72), ("original_fen", ctypes.c_char * 72),   # a fifth of data_as()
                ("fixes", ctypes.c_int32 * 64), ("n_fixes", ctypes.c_int32), ("squares", ctypes.c_void_p)]


class _Param(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_float)), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 4)]


def library_path() -> Path:
    env = os.environ.get("CHESSVISION_HIP_LIB")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent.parent / "lib" / _LIB_NAME


_lib = None
_lib_lock = threading.Lock()

# (name, restype, argtypes) for every symbol include/chessvision_hip.h declares
_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_fp = ctypes.POINTER(ctypes.c_float)
SYMBOLS = [
    ("cv_abi_version", _i, []),
    ("cv_last_error", ctypes.c_char_p, []),
    ("cv_device_count", _i, [ctypes.POINTER(_i)]),
    ("cv_engine_create", _i, [_i, _i, ctypes.POINTER(_vp)]),
    ("cv_engine_destroy", _i, [_vp]),
    ("cv_trim_memory", _i, [ctypes.POINTER(ctypes.c_size_t)]),
    ("cv_load_unet", _i, [_vp, ctypes.POINTER(_Param), _i]),
    ("cv_load_resnet18", _i, [_vp, ctypes.POINTER(_Param), _i]),
    ("cv_engine_set_chunk", _i, [_vp, _i, _i]),
    ("cv_unet_forward", _i, [_vp, _vp, _i, _vp, _vp]),
    ("cv_resnet18_forward", _i, [_vp, _vp, _i, _vp, _vp]),
    ("cv_unet_forward_u8", _i, [_vp, _vp, _i, _vp, _vp, _f, _vp]),
    ("cv_resnet18_forward_u8", _i, [_vp, _vp, _i, _vp, _vp]),
    ("cv_softmax13", _i, [_vp, _vp, _i, _vp, _vp]),
    ("cv_get_activation", _i, [_vp, ctypes.c_char_p, ctypes.c_char_p, _fp, ctypes.c_size_t,
                               ctypes.POINTER(ctypes.c_int64)]),
    ("cv_model_macs", _i, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    ("cv_profile_convs", _i, [_vp, ctypes.c_char_p, _vp, _i, _vp, _i, _vp, ctypes.POINTER(_f),
                              ctypes.POINTER(_i), ctypes.POINTER(_f)]),
    ("cv_profile_entry", _i, [_vp, _i, ctypes.c_char_p, _i, ctypes.POINTER(_f), ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(_i)]),
    ("cv_op_conv2d", _i, [_vp, _vp, _i, _i, _i, _i, _fp, _i, _i, _i, _fp, _fp, _vp, _i, _vp, _vp]),
    ("cv_op_conv_transpose2x2", _i, [_vp, _vp, _i, _i, _i, _i, _fp, _i, _fp, _vp, _vp]),
    ("cv_op_outc_1x1", _i, [_vp, _vp, _i, _i, _i, _i, _fp, _fp, _f, _vp, _vp, _vp]),
    ("cv_op_maxpool2x2", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cv_op_maxpool3x3s2", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cv_op_upsample_bilinear2x", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cv_selftest_mfma", _i, [_vp, ctypes.POINTER(_f), ctypes.POINTER(_f)]),
    ("cv_find_quadrangle", _i, [_vp, _i, _i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_i)]),
    ("cv_find_quadrangles", _i, [_vp, _i, _i, _i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), _i]),
    ("cv_find_contours", _i, [_vp, _i, _i, _i, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64, ctypes.POINTER(ctypes.c_int32),
                              ctypes.POINTER(ctypes.c_int32), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    ("cv_resize_area_u8", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    ("cv_extract_squares_u8", _i, [_vp, _vp, _i, _i, _i, ctypes.POINTER(ctypes.c_double), _vp, _vp, _vp]),
    ("cv_extract_squares_u8_dev", _i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("cv_engine_workspace_bytes", _i, [_vp, ctypes.POINTER(ctypes.c_size_t)]),
    ("cv_engine_numeric_status", _i, [_vp, _vp]),
    ("cv_get_activation_exponent", _i, [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_i)]),
    ("cv_profile_entry_bytes", _i, [_vp, _i, ctypes.POINTER(ctypes.c_double)]),
    ("cv_profile_entry_kernel", _i, [_vp, _i, ctypes.c_char_p, _i]),
    ("cv_engine_export_calibration", _i, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), _i, ctypes.POINTER(_i)]),
    ("cv_engine_import_calibration", _i, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), _i, ctypes.POINTER(_i)]),
    ("cv_process_image", _i, [_vp, _vp, _vp, _i, _i, _f, _i, _i, ctypes.c_void_p, _vp]),
    ("cv_process_image_v2", _i, [_vp, _vp, _vp, _i, _i, _f, _i, _i, ctypes.c_void_p, ctypes.c_size_t, _vp]),
    ("cv_board_homographies", _i, [_fp, _i, _i, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    ("cv_decode_positions", _i, [_fp, _i, _i, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int8),
                                 ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
]


def trim_memory() -> int:
    """Hand the blocks of closed engines, which the library keeps for the next load, back to the driver (``cv_trim_memory``);
    returns the bytes freed.  The counterpart of ``torch.cuda.empty_cache()`` for this library's own allocations."""
    n = ctypes.c_size_t(0)
    _check(load_library().cv_trim_memory(ctypes.byref(n)))
    return int(n.value)


def load_library():
    """dlopen the C-ABI library (built by ``__graft_entry__.build()``); raises if it is missing."""
    global _lib
    with _lib_lock:
        if _lib is None:
            path = library_path()
            if not path.exists():
                raise HipBackendError(
                    f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
            lib = ctypes.CDLL(str(path))
            lib.cv_abi_version.restype = ctypes.c_int
            lib.cv_abi_version.argtypes = []
            found = lib.cv_abi_version()
            if found != ABI_VERSION:               # before binding the rest: a stale build fails with this, not an AttributeError
                raise HipBackendError(f"{path}: ABI version {found}, this package needs {ABI_VERSION} -- rebuild the library "
                                      "(`python -c 'import __graft_entry__ as g; g.build()'`)")
            for name, restype, argtypes in SYMBOLS:
                fn = getattr(lib, name)            # AttributeError if the export is missing
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def _check(status: int) -> None:
    if status != 0:
        msg = load_library().cv_last_error()
        text = f"[cv status {status}] {msg.decode(errors='replace') if msg else 'unknown error'}"
        raise NumericRangeError(text) if status == CV_ERR_NUMERIC else HipBackendError(text)


def _stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: torch.Tensor) -> int:
    return int(t.data_ptr())


def _np_f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().to("cpu", torch.float32).numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _as_param_table(state_dict: Mapping[str, object]):
    keep = []                                   # keeps numpy buffers alive during the call
    entries = []
    for key, value in state_dict.items():
        if key.endswith("num_batches_tracked"):
            continue
        arr = _np_f32(value)
        if arr.ndim > 4:
            raise HipBackendError(f"state-dict entry {key} has {arr.ndim} dims")
        keep.append(arr)
        shape = (ctypes.c_int64 * 4)(*(list(arr.shape) + [0] * (4 - arr.ndim)))
        entries.append(_Param(key.encode(), arr.ctypes.data_as(_fp), arr.ndim, shape))
    table = (_Param * len(entries))(*entries)
    return table, len(entries), keep


def find_quadrangle(mask: np.ndarray):
    """Binary mask (H, W) uint8 -> (4,1,2) int32 quadrangle in the reference's vertex order, or None.
    Host-side C++ (csrc/contour.cpp); needs neither a GPU nor an engine.  Same result as
    ``ChessVision._find_quadrangle`` (which stays the readable numpy restatement and the test checker)."""
    lib = load_library()
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if m.ndim != 2:
        raise HipBackendError("find_quadrangle expects a 2-D uint8 mask")
    quad = (ctypes.c_int32 * 8)()
    found = _i(0)
    _check(lib.cv_find_quadrangle(m.ctypes.data_as(_vp), m.shape[0], m.shape[1], quad, ctypes.byref(found)))
    if not found.value:
        return None
    return np.array(list(quad), dtype=np.int32).reshape(4, 1, 2)


def find_contours(mask: np.ndarray, tc89: bool = True):
    """``cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_TC89_KCOS if tc89 else CHAIN_APPROX_NONE)[0]`` (reference core.py:360)
    from the native contour stage: list of (n,1,2) int32 contours in OpenCV's order, and the list of their hole flags."""
    lib = load_library()
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if m.ndim != 2:
        raise HipBackendError("find_contours expects a 2-D uint8 mask")
    h, w = m.shape
    cap_pts, cap_c = 4 * h * w + 16, h * w + 16
    xy = np.zeros((cap_pts, 2), dtype=np.int32)
    counts = np.zeros(cap_c, dtype=np.int32)
    holes = np.zeros(cap_c, dtype=np.int32)
    n = ctypes.c_int64(0)
    i32p = ctypes.POINTER(ctypes.c_int32)
    _check(lib.cv_find_contours(m.ctypes.data_as(_vp), h, w, 1 if tc89 else 0, xy.ctypes.data_as(i32p), cap_pts,
                                counts.ctypes.data_as(i32p), holes.ctypes.data_as(i32p), cap_c, ctypes.byref(n)))
    out, at = [], 0
    for k in range(n.value):
        out.append(xy[at:at + counts[k]].reshape(-1, 1, 2).copy())
        at += int(counts[k])
    return out, [bool(v) for v in holes[:n.value]]


def find_quadrangles(masks: np.ndarray, n_threads: int = 0) -> list:
    """(N,H,W) uint8 masks -> list of (4,1,2) int32 quadrangles / None, on native host threads (GIL released)."""
    lib = load_library()
    m = np.ascontiguousarray(masks, dtype=np.uint8)
    if m.ndim != 3:
        raise HipBackendError("find_quadrangles expects (N,H,W) uint8 masks")
    n = m.shape[0]
    quads = np.zeros((n, 8), dtype=np.int32)
    found = np.zeros(n, dtype=np.int32)
    if n:
        _check(lib.cv_find_quadrangles(m.ctypes.data_as(_vp), n, m.shape[1], m.shape[2],
                                       quads.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                       found.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(n_threads)))
    return [quads[i].reshape(4, 1, 2).copy() if found[i] else None for i in range(n)]


def board_homographies(quads: np.ndarray, out_size=(512, 512), want_forward: bool = False):
    """(N,4,2) float32 quadrangles (source-image pixels, the reference's TR, TL, BL, BR order) -> (N,3,3) float64 INVERSE
    matrices (board pixel -> source pixel) of the warp ``utils.extract_perspective`` performs (reference utils.py:115-132), computed
    in OpenCV's order of operations by csrc/homography.cpp; with ``want_forward`` also the getPerspectiveTransform matrices.
    Bit-identical to ``classical.invert3(classical.get_perspective_transform(q, dest))``.  Needs neither a GPU nor an engine."""
    lib = load_library()
    q = np.ascontiguousarray(quads, dtype=np.float32).reshape(-1, 8)
    n = q.shape[0]
    inv = np.zeros((n, 3, 3), dtype=np.float64)
    fwd = np.zeros((n, 3, 3), dtype=np.float64) if want_forward else None
    dp = ctypes.POINTER(ctypes.c_double)
    if n:
        _check(lib.cv_board_homographies(q.ctypes.data_as(_fp), n, int(out_size[0]), int(out_size[1]),
                                         fwd.ctypes.data_as(dp) if want_forward else None, inv.ctypes.data_as(dp)))
    return (inv, fwd) if want_forward else inv


def process_image_native(unet_engine: "HipEngine", classifier_engine: "HipEngine", image: np.ndarray, threshold: float = 0.5,
                         flip: bool = False, fallback_quad: bool = False, stream: int | None = None) -> dict:
    """One host image through ``cv_process_image`` (the native form of ``ChessVision.process_image``, reference core.py:152-195):
    returns {"logits" (256,256) f32, "mask" (256,256) u8, "found", and when found "quadrangle" (4,1,2) f32, "board" (512,512) u8,
    "probabilities" (64,13) f32, "squares" (64,64,64,1) u8, "fen", "original_fen", "fixes" [(square index, original class, corrected class)]}."""
    lib = load_library()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise HipBackendError("process_image_native expects an (H,W,3) uint8 image")
    logits = np.empty((256, 256), np.float32)
    mask = np.empty((256, 256), np.uint8)
    board = np.empty((512, 512), np.uint8)
    probs = np.empty((64, 13), np.float32)
    squares = np.empty((64, 64, 64, 1), np.uint8)           # PositionResult.squares, cut on the native side (-10 us against the numpy reshape)
    res = _ImageResult()
    res.logits, res.mask, res.board = logits.ctypes.data, mask.ctypes.data, board.ctypes.data
    res.probabilities, res.squares, res.labels = probs.ctypes.data, squares.ctypes.data, None
    _check(lib.cv_process_image_v2(unet_engine._h, classifier_engine._h, img.ctypes.data, img.shape[0], img.shape[1],
                                   float(threshold), int(bool(flip)), int(bool(fallback_quad)), ctypes.byref(res), ctypes.sizeof(res),
                                   _stream_ptr(unet_engine.device) if stream is None else stream))    # stream: a request slot's own HIP stream
    out = {"logits": logits, "mask": mask, "found": bool(res.found)}
    if res.found:
        out.update(quadrangle=np.frombuffer(res.quadrangle, dtype=np.float32).reshape(4, 1, 2).copy(), board=board, probabilities=probs, squares=squares,
                   fen=res.fen.decode(), original_fen=res.original_fen.decode(),
                   fixes=[(int(res.fixes[4 * i + 1]), int(res.fixes[4 * i + 2]), int(res.fixes[4 * i + 3])) for i in range(res.n_fixes)])
    return out


def decode_positions(probabilities: np.ndarray, flip: bool = False):
    """(N,64,13) float32 class probabilities -> (fens, original_fens, labels (N,64) int8, fixes) for the whole job in one
    native call (csrc/position.cpp): per-square argmax, the pawn rule and both FEN strings, exactly as
    ``ChessVision.process_position_probabilities`` computes them per board.  ``fixes`` is a list of
    (board, square index, original class index, corrected class index).  Needs neither a GPU nor an engine."""
    lib = load_library()
    p = np.ascontiguousarray(probabilities, dtype=np.float32)
    if p.ndim != 3 or p.shape[1:] != (64, 13):
        raise HipBackendError("decode_positions expects (N,64,13) float32 probabilities")
    n = p.shape[0]
    fen = ctypes.create_string_buffer(max(1, n * 72))
    orig = ctypes.create_string_buffer(max(1, n * 72))
    labels = np.zeros((n, 64), dtype=np.int8)
    fixes = np.zeros((max(1, n * 16), 4), dtype=np.int32)
    n_fixes = ctypes.c_int32(0)
    if n:
        _check(lib.cv_decode_positions(p.ctypes.data_as(_fp), n, int(bool(flip)), fen, orig,
                                       labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)),
                                       fixes.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(n_fixes)))
    raw_f, raw_o = fen.raw, orig.raw
    fens = [raw_f[i * 72:(i + 1) * 72].split(b"\0", 1)[0].decode() for i in range(n)]
    origs = [raw_o[i * 72:(i + 1) * 72].split(b"\0", 1)[0].decode() for i in range(n)]
    return fens, origs, labels, [tuple(int(v) for v in fixes[i]) for i in range(n_fixes.value)]


class HipEngine:
    """One engine = one device + one arithmetic precision + packed weights + workspace."""

    def __init__(self, device: torch.device | str | int | None = None, precision: str = "f16",
                 unet_chunk: int = 0, resnet_chunk: int = 0):
        self._lib = load_library()
        if not torch.cuda.is_available():
            raise HipBackendError("no ROCm device visible to PyTorch: the HIP backend has no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise HipBackendError(f"HIP backend needs a cuda(ROCm) device, got {dev}")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if precision not in _PRECISIONS:
            raise HipBackendError(f"precision must be one of {sorted(_PRECISIONS)}")
        self.device = dev
        self.precision = _PREC_NAMES[_PRECISIONS[precision]]
        self._h = ctypes.c_void_p()
        _check(self._lib.cv_engine_create(dev.index, _PRECISIONS[precision], ctypes.byref(self._h)))
        # chunk = images / squares per pass (0 = library default 64 / 16384).  The activation workspace is elastic: it is
        # allocated for the largest batch seen so far (at most one chunk), so a single-image server stays under 1 GB.
        unet_chunk = int(unet_chunk or os.environ.get("CHESSVISION_HIP_UNET_CHUNK", "0") or 0)
        resnet_chunk = int(resnet_chunk or os.environ.get("CHESSVISION_HIP_RESNET_CHUNK", "0") or 0)
        if unet_chunk or resnet_chunk:
            _check(self._lib.cv_engine_set_chunk(self._h, unet_chunk, resnet_chunk))
        self.has_unet = False
        self.has_resnet = False

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.cv_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights --------------------------------------------------------------------------------
    def load_unet(self, state_dict: Mapping[str, object]) -> None:
        table, n, keep = _as_param_table(state_dict)
        _check(self._lib.cv_load_unet(self._h, table, n))
        del keep
        self.has_unet = True

    def load_resnet18(self, state_dict: Mapping[str, object]) -> None:
        table, n, keep = _as_param_table(state_dict)
        _check(self._lib.cv_load_resnet18(self._h, table, n))
        del keep
        self.has_resnet = True

    # -- forward ----------------------------------------------------------------------------------
    def _dev_f32(self, x: torch.Tensor, shape_tail) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            raise HipBackendError("input must be a torch.Tensor")
        if tuple(x.shape[1:]) != tuple(shape_tail):
            raise HipBackendError(f"input shape {tuple(x.shape)} != (N,{','.join(map(str, shape_tail))})")
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def check_numerics(self) -> None:
        """Synchronise the current stream and raise ``HipBackendError`` naming the first layer that produced a non-finite
        value since the last check (f16 range exceeded, NaN in the input); re-arms the guard."""
        _check(self._lib.cv_engine_numeric_status(self._h, _stream_ptr(self.device)))

    def export_calibration(self, model: str) -> np.ndarray:
        """The load-time range calibration of ``model`` ("unet" | "resnet18") as an int32 vector (two exponents per tensor)."""
        n = _i()
        _check(self._lib.cv_engine_export_calibration(self._h, model.encode(), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        _check(self._lib.cv_engine_export_calibration(self._h, model.encode(), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                      n.value, ctypes.byref(n)))
        return out

    def import_calibration(self, model: str, exponents: np.ndarray) -> bool:
        """Set the tensor exponents of ``model`` (a vector from ``export_calibration`` of an engine with the same precision and
        checkpoint layout); True when anything changed."""
        e = np.ascontiguousarray(exponents, dtype=np.int32)
        changed = _i()
        _check(self._lib.cv_engine_import_calibration(self._h, model.encode(), e.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                      int(e.size), ctypes.byref(changed)))
        return bool(changed.value)

    def workspace_bytes(self) -> int:
        v = ctypes.c_size_t()
        _check(self._lib.cv_engine_workspace_bytes(self._h, ctypes.byref(v)))
        return int(v.value)

    def unet_forward(self, x: torch.Tensor, check: bool = True) -> torch.Tensor:
        """(B,3,256,256) float32 in [0,1] -> (B,1,256,256) float32 logits (device tensor).  ``check`` consults the
        numeric guard (synchronises, like the ``.cpu()`` the reference does next); throughput loops pass False and call
        ``check_numerics()`` once at the end."""
        x = self._dev_f32(x, (3, 256, 256))
        out = torch.empty((x.shape[0], 1, 256, 256), dtype=torch.float32, device=self.device)
        _check(self._lib.cv_unet_forward(self._h, _ptr(x), x.shape[0], _ptr(out), _stream_ptr(self.device)))
        if check:
            self.check_numerics()
        return out

    def resnet18_forward(self, x: torch.Tensor, check: bool = True) -> torch.Tensor:
        """(N,1,64,64) float32 in [0,1] -> (N,13) float32 logits (device tensor)."""
        x = self._dev_f32(x, (1, 64, 64))
        out = torch.empty((x.shape[0], 13), dtype=torch.float32, device=self.device)
        _check(self._lib.cv_resnet18_forward(self._h, _ptr(x), x.shape[0], _ptr(out), _stream_ptr(self.device)))
        if check:
            self.check_numerics()
        return out

    def unet_forward_u8(self, x_u8: torch.Tensor, threshold: float = 0.5, want_mask: bool = True):
        """(B,256,256,3) uint8 HWC -> (logits (B,1,256,256) f32, mask (B,256,256) u8 | None)."""
        if x_u8.dtype != torch.uint8 or tuple(x_u8.shape[1:]) != (256, 256, 3):
            raise HipBackendError("unet_forward_u8 expects (B,256,256,3) uint8")
        x_u8 = x_u8.to(self.device).contiguous()
        b = x_u8.shape[0]
        logits = torch.empty((b, 1, 256, 256), dtype=torch.float32, device=self.device)
        mask = torch.empty((b, 256, 256), dtype=torch.uint8, device=self.device) if want_mask else None
        _check(self._lib.cv_unet_forward_u8(self._h, _ptr(x_u8), b, _ptr(logits), _ptr(mask) if want_mask else None,
                                            float(threshold), _stream_ptr(self.device)))
        return logits, mask

    def resnet18_forward_u8(self, squares_u8: torch.Tensor) -> torch.Tensor:
        """(N,64,64) uint8 -> (N,13) float32 softmax probabilities."""
        if squares_u8.dtype != torch.uint8 or tuple(squares_u8.shape[1:]) != (64, 64):
            raise HipBackendError("resnet18_forward_u8 expects (N,64,64) uint8")
        squares_u8 = squares_u8.to(self.device).contiguous()
        n = squares_u8.shape[0]
        out = torch.empty((n, 13), dtype=torch.float32, device=self.device)
        _check(self._lib.cv_resnet18_forward_u8(self._h, _ptr(squares_u8), n, _ptr(out), _stream_ptr(self.device)))
        return out

    def softmax13(self, logits: torch.Tensor) -> torch.Tensor:
        logits = logits.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(logits)
        _check(self._lib.cv_softmax13(self._h, _ptr(logits), logits.shape[0], _ptr(out), _stream_ptr(self.device)))
        return out

    # -- classical stages on the device ------------------------------------------
    def resize_area_u8(self, images: torch.Tensor, out_hw=(256, 256)) -> torch.Tensor:
        """(N,H,W,C) uint8 -> (N,out_h,out_w,C) uint8 with INTER_AREA semantics (reference core.py:212)."""
        if images.dtype != torch.uint8 or images.dim() != 4:
            raise HipBackendError("resize_area_u8 expects (N,H,W,C) uint8")
        images = images.to(self.device).contiguous()
        n, h, w, c = images.shape
        out = torch.empty((n, out_hw[0], out_hw[1], c), dtype=torch.uint8, device=self.device)
        _check(self._lib.cv_resize_area_u8(self._h, _ptr(images), n, h, w, c, _ptr(out), out_hw[0], out_hw[1],
                                           _stream_ptr(self.device)))
        return out

    def extract_squares_u8(self, images: torch.Tensor, inverse_maps: np.ndarray, want_boards: bool = True):
        """images (N,H,W,3) uint8 BGR on the device + N inverse homographies (board pixel -> source pixel) ->
        (squares (N*64,64,64) uint8, boards (N,512,512) uint8 | None): warp + gray + flip + split, fused."""
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
            raise HipBackendError("extract_squares_u8 expects (N,H,W,3) uint8")
        images = images.to(self.device).contiguous()
        n, h, w, _ = images.shape
        # the matrices travel through a pinned staging tensor on the current stream; nothing here blocks the host, so the
        # next job's UNet (already queued on this stream) does not hold the classifier of this job back
        if isinstance(inverse_maps, torch.Tensor):           # already staged by the caller (page-locked, float64, n x 9)
            if inverse_maps.dtype != torch.float64 or inverse_maps.numel() != n * 9:
                raise HipBackendError("extract_squares_u8 expects n x 9 float64 matrices")
            inv = inverse_maps
        else:
            inv = torch.from_numpy(np.ascontiguousarray(inverse_maps, dtype=np.float64).reshape(n, 9)).pin_memory()
        inv_dev = inv.to(self.device, non_blocking=True)
        squares = torch.empty((n * 64, 64, 64), dtype=torch.uint8, device=self.device)
        boards = torch.empty((n, 512, 512), dtype=torch.uint8, device=self.device) if want_boards else None
        _check(self._lib.cv_extract_squares_u8_dev(self._h, _ptr(images), n, h, w, _ptr(inv_dev), _ptr(squares),
                                                   _ptr(boards) if want_boards else None, _stream_ptr(self.device)))
        return squares, boards

    # -- introspection ----------------------------------------------------------------------------
    def activation(self, model: str, name: str) -> np.ndarray:
        dims = (ctypes.c_int64 * 4)()
        _check(self._lib.cv_get_activation(self._h, model.encode(), name.encode(), None, 0, dims))
        out = np.empty(tuple(int(d) for d in dims), dtype=np.float32)
        _check(self._lib.cv_get_activation(self._h, model.encode(), name.encode(), out.ctypes.data_as(_fp), out.size,
                                           dims))
        return out

    def activation_exponent(self, model: str, name: str) -> int:
        """Power-of-two exponent the tensor is stored with (stored = value * 2^-exponent), chosen by the load-time calibration."""
        v = _i()
        _check(self._lib.cv_get_activation_exponent(self._h, model.encode(), name.encode(), ctypes.byref(v)))
        return int(v.value)

    def model_macs(self, model: str) -> int:
        v = ctypes.c_int64()
        _check(self._lib.cv_model_macs(self._h, model.encode(), ctypes.byref(v)))
        return int(v.value)

    def profile(self, model: str, x: torch.Tensor, iters: int = 1):
        """Event-time every launch of `iters` forwards; returns (conv_ms, conv_launches, all_ms, entries)."""
        x = x.to(self.device, torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty((n, 1, 256, 256) if model == "unet" else (n, 13), dtype=torch.float32, device=self.device)
        conv_ms, all_ms, launches = _f(), _f(), _i()
        _check(self._lib.cv_profile_convs(self._h, model.encode(), _ptr(x), n, _ptr(out), iters,
                                          _stream_ptr(self.device), ctypes.byref(conv_ms), ctypes.byref(launches),
                                          ctypes.byref(all_ms)))
        entries = []
        idx = 0
        name, kern = ctypes.create_string_buffer(128), ctypes.create_string_buffer(160)
        ms, macs, is_conv, nbytes = _f(), ctypes.c_double(), _i(), ctypes.c_double()
        while self._lib.cv_profile_entry(self._h, idx, name, 128, ctypes.byref(ms), ctypes.byref(macs),
                                         ctypes.byref(is_conv)) == 0:
            _check(self._lib.cv_profile_entry_bytes(self._h, idx, ctypes.byref(nbytes)))
            _check(self._lib.cv_profile_entry_kernel(self._h, idx, kern, 160))
            entries.append({"name": name.value.decode(), "ms": ms.value, "macs": macs.value, "conv": bool(is_conv.value),
                            "bytes": nbytes.value, "kernel": kern.value.decode()})
            idx += 1
        return conv_ms.value, launches.value, all_ms.value, entries

    # -- single-layer entry points (parity tests) ------------------------------------------------
    def op_conv2d(self, x, w, stride=1, scale=None, shift=None, residual=None, relu=False) -> torch.Tensor:
        x = x.to(self.device, torch.float32).contiguous()
        w = _np_f32(w)
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        pad = (k - 1) // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
        y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=self.device)
        sc = _np_f32(scale) if scale is not None else None
        sh = _np_f32(shift) if shift is not None else None
        res = residual.to(self.device, torch.float32).contiguous() if residual is not None else None
        _check(self._lib.cv_op_conv2d(self._h, _ptr(x), n, cin, h, wd, w.ctypes.data_as(_fp), cout, k, stride,
                                      sc.ctypes.data_as(_fp) if sc is not None else None,
                                      sh.ctypes.data_as(_fp) if sh is not None else None,
                                      _ptr(res) if res is not None else None, int(bool(relu)), _ptr(y),
                                      _stream_ptr(self.device)))
        return y

    def op_conv_transpose2x2(self, x, w, bias) -> torch.Tensor:
        x = x.to(self.device, torch.float32).contiguous()
        w, bias = _np_f32(w), _np_f32(bias)
        n, cin, h, wd = x.shape
        cout = w.shape[1]
        y = torch.empty((n, cout, 2 * h, 2 * wd), dtype=torch.float32, device=self.device)
        _check(self._lib.cv_op_conv_transpose2x2(self._h, _ptr(x), n, cin, h, wd, w.ctypes.data_as(_fp), cout,
                                                 bias.ctypes.data_as(_fp), _ptr(y), _stream_ptr(self.device)))
        return y

    def op_outc_1x1(self, x, w, bias, threshold: float = 0.5):
        """(n,c,h,w) -> (logits (n,1,h,w) f32, mask (n,h,w) u8) through the stand-alone OutConv kernel."""
        x = x.to(self.device, torch.float32).contiguous()
        w, bias = _np_f32(w).reshape(-1), _np_f32(bias).reshape(-1)
        n, c, h, wd = x.shape
        logits = torch.empty((n, 1, h, wd), dtype=torch.float32, device=self.device)
        mask = torch.empty((n, h, wd), dtype=torch.uint8, device=self.device)
        _check(self._lib.cv_op_outc_1x1(self._h, _ptr(x), n, c, h, wd, w.ctypes.data_as(_fp), bias.ctypes.data_as(_fp),
                                        float(threshold), _ptr(logits), _ptr(mask), _stream_ptr(self.device)))
        return logits, mask

    def _op_pool(self, fn, x, ho, wo) -> torch.Tensor:
        x = x.to(self.device, torch.float32).contiguous()
        n, c, h, wd = x.shape
        y = torch.empty((n, c, ho(h), wo(wd)), dtype=torch.float32, device=self.device)
        _check(fn(self._h, _ptr(x), n, c, h, wd, _ptr(y), _stream_ptr(self.device)))
        return y

    def op_maxpool2x2(self, x):
        return self._op_pool(self._lib.cv_op_maxpool2x2, x, lambda h: h // 2, lambda w: w // 2)

    def op_maxpool3x3s2(self, x):
        return self._op_pool(self._lib.cv_op_maxpool3x3s2, x, lambda h: (h - 1) // 2 + 1, lambda w: (w - 1) // 2 + 1)

    def op_upsample_bilinear2x(self, x):
        return self._op_pool(self._lib.cv_op_upsample_bilinear2x, x, lambda h: 2 * h, lambda w: 2 * w)

    def selftest_mfma(self):
        a, b = _f(), _f()
        _check(self._lib.cv_selftest_mfma(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value


class _HipModel:
    """Common duck-typed module surface: ``__call__``, ``eval``, ``train``, ``to``, ``metadata``."""

    model_name = ""

    def __init__(self, engine: HipEngine, metadata: dict | None = None):
        self.engine = engine
        if metadata:
            self.metadata = metadata

    def eval(self):
        return self                                    # inference-only engine

    def train(self, mode: bool = True):
        if mode:
            raise HipBackendError("the HIP backend is inference-only (reference training is out of scope)")
        return self

    def to(self, device=None, *_, **__):
        if device is not None and torch.device(device).type != "cuda":
            raise HipBackendError(f"HIP model cannot move to {device}: there is no CPU fallback")
        return self

    def parameters(self):
        return iter(())


class HipBoardExtractor(_HipModel):
    """UNet(3,1) forward on MI355X; drop-in for ``ChessVision.board_extractor`` (core.py:66-72,220)."""

    model_name = "unet"

    def __call__(self, image_batch: torch.Tensor) -> torch.Tensor:
        return self.engine.unet_forward(image_batch)


class HipPieceClassifier(_HipModel):
    """ResNet-18 (1ch, 13 classes) forward on MI355X; drop-in for ``ChessVision.classifier`` (core.py:74-82,241)."""

    model_name = "resnet18"

    def __call__(self, batch: torch.Tensor) -> torch.Tensor:
        return self.engine.resnet18_forward(batch)
