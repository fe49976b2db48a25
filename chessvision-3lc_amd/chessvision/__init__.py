"""MI355X-native ChessVision CNN hot path behind the reference's own API surface.

``from chessvision import ChessVision`` works exactly as with the reference package
(reference ``chessvision/__init__.py:1-3``); the names BASELINE.json's north_star uses that do not exist in
the reference (``BoardExtractor``, ``PieceClassifier``, ``ChessVision.predict``) are provided as aliases.
Sub-modules are imported lazily so that ``import chessvision.synthetic`` does not pull in the HIP backend.
"""
from __future__ import annotations

import os as _os

# Request slots (core.py) run up to four ``process_image`` calls side by side, each on its own stream beside the engines' side
# streams; the HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two slots that
# share a queue serialise (four request threads: 2086 -> 2313 requests/s with 16 queues).  The runtime reads the variable when it
# initialises -- the first HIP call of the process, e.g. ``torch.cuda.is_available()`` -- so it is set here, at import, and only when the
# host application has not chosen a value itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__all__ = ["ChessVision", "BoardExtractor", "PieceClassifier", "HipEngine"]


def __getattr__(name: str):
    if name == "ChessVision":
        from .core import ChessVision
        return ChessVision
    if name in ("BoardExtractor", "HipBoardExtractor"):
        from .hip_backend import HipBoardExtractor
        return HipBoardExtractor
    if name in ("PieceClassifier", "HipPieceClassifier"):
        from .hip_backend import HipPieceClassifier
        return HipPieceClassifier
    if name == "HipEngine":
        from .hip_backend import HipEngine
        return HipEngine
    raise AttributeError(f"module 'chessvision' has no attribute {name!r}")
