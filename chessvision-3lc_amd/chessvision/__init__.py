"""MI355X-native ChessVision CNN hot path behind the reference's own API surface.

``from chessvision import ChessVision`` works exactly as with the reference package
(reference ``chessvision/__init__.py:1-3``); the names BASELINE.json's north_star uses that do not exist in
the reference (``BoardExtractor``, ``PieceClassifier``, ``ChessVision.predict``) are provided as aliases.
Sub-modules are imported lazily so that ``import chessvision.synthetic`` does not pull in the HIP backend.
"""
from __future__ import annotations

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
