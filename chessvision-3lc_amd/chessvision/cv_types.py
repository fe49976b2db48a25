"""Result records handed back by ``ChessVision``.

The attribute names and their order are the public contract of the reference's result objects
(``chessvision/cv_types.py:9-62`` there; read by ``app/computeroot/cv_endpoint.py:165-171`` and
``scripts/eval/evaluate.py:280-330``), so they are kept; everything else about this module is local.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from numpy.typing import NDArray

U8 = NDArray[np.uint8]
F32 = NDArray[np.float32]


def _doc(text: str):
    return field(metadata={"doc": text})


@dataclass
class ValidationFix:
    square_name: str = _doc('board coordinate, "a1".."h8"')
    original_piece: str = _doc("symbol the classifier chose")
    corrected_piece: str = _doc("symbol after the rule fired")
    rule_name: str = _doc('rule identifier, e.g. "no_pawns_on_ends"')


@dataclass
class BoardExtractionResult:
    probabilities: F32 = _doc("raw UNet LOGITS, (256,256) -- the reference stores logits under this name (core.py:287,306)")
    binary_mask: U8 = _doc("0 / 255 segmentation mask, (256,256)")
    quadrangle: F32 | None = _doc("(4,1,2) corners in input-image pixels; None when no board was found")
    board_image: U8 | None = _doc("(512,512) rectified gray board; None when no board was found")


@dataclass
class PositionResult:
    fen: str = _doc("piece placement after rule validation")
    original_fen: str = _doc("piece placement straight from the per-square argmax")
    model_probabilities: F32 = _doc("(64,13) class probabilities")
    squares: U8 | None = _doc("(64,64,64,1) square crops, a8..h1 order (None from process_images unless return_crops=True)")
    square_names: list[str] = _doc("coordinate of each row of the two arrays above")
    validation_fixes: list[ValidationFix] = _doc("rule corrections that were applied")

    @property
    def confidence_scores(self) -> list[float]:
        """Top class probability per square.  NOT a reference field: ``cv_endpoint.py:169,227`` reads it although the
        reference result lacks it (HTTP 500 there); provided so the Flask app works against this package."""
        return [float(v) for v in np.max(self.model_probabilities, axis=1)]


@dataclass
class ChessVisionResult:
    board_extraction: BoardExtractionResult = _doc("always present")
    position: PositionResult | None = _doc("None when board extraction failed")
    processing_time: float = _doc("seconds spent in process_image (per image for process_images)")


@dataclass
class ValidationMetrics:
    accuracy_before: float = _doc("square accuracy of original_fen")
    accuracy_after: float = _doc("square accuracy of fen")
    num_fixes: int = _doc("len(fixes)")
    fixes: list[ValidationFix] = _doc("the corrections")

    @property
    def accuracy_delta(self) -> float:
        return self.accuracy_after - self.accuracy_before
