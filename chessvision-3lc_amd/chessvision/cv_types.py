"""Result records of the pipeline; field names and order as the reference's ``chessvision/cv_types.py:9-62``."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from numpy.typing import NDArray


@dataclass
class ValidationFix:
    square_name: str        # e.g. "e1"
    original_piece: str     # symbol predicted by the classifier
    corrected_piece: str    # symbol after the rule fired
    rule_name: str


@dataclass
class BoardExtractionResult:
    probabilities: NDArray[np.float32]          # NB: raw UNet *logits* (reference core.py:287,306)
    binary_mask: NDArray[np.uint8]              # 0 / 255
    quadrangle: NDArray[np.float32] | None      # 4 corners in input-image pixels, None when no board
    board_image: NDArray[np.uint8] | None       # 512x512 gray, None when no board


@dataclass
class PositionResult:
    fen: str                                    # after rule validation
    original_fen: str                           # straight argmax
    model_probabilities: NDArray[np.float32]    # (64, 13)
    squares: NDArray[np.uint8]                  # (64, 64, 64, 1)
    square_names: list[str]
    validation_fixes: list[ValidationFix]

    @property
    def confidence_scores(self) -> list[float]:
        """Highest class probability per square.  Not a reference field: ``app/computeroot/cv_endpoint.py:169,227``
        reads it and the reference result lacks it (HTTP 500 there); provided so the Flask app works."""
        return [float(v) for v in np.max(self.model_probabilities, axis=1)]


@dataclass
class ChessVisionResult:
    board_extraction: BoardExtractionResult
    position: PositionResult | None
    processing_time: float


@dataclass
class ValidationMetrics:
    accuracy_before: float
    accuracy_after: float
    num_fixes: int
    fixes: list[ValidationFix]

    @property
    def accuracy_delta(self) -> float:
        return self.accuracy_after - self.accuracy_before
