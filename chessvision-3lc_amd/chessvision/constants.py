"""Sizes, label tables and default paths of the ChessVision pipeline.

Values follow the reference's ``chessvision/constants.py`` (sizes :15-23, label order :23, weight paths :46-50,
square tables :109-129); the square-name tables are generated instead of spelled out.
"""
from __future__ import annotations

import os
from pathlib import Path

# The reference resolves CVROOT to the repository that contains the package (constants.py:7).
CVROOT = os.getenv("CVROOT", Path(__file__).resolve().parent.parent.as_posix())
DATA_ROOT = Path(CVROOT) / "data"
WEIGHTS_DIR = Path(CVROOT) / "weights"

BEST_EXTRACTOR_WEIGHTS = str(WEIGHTS_DIR / "best_extractor.pth")
BEST_CLASSIFIER_WEIGHTS = str(WEIGHTS_DIR / "best_classifier.pth")
BEST_YOLO_EXTRACTOR = str(WEIGHTS_DIR / "best_yolo_extractor.pt")
BEST_YOLO_CLASSIFIER = str(WEIGHTS_DIR / "best_yolo_classifier.pt")
BLACK_BOARD_PATH = (DATA_ROOT / "board_extraction" / "black_board.png").as_posix()
BLACK_SQUARE_PATH = (DATA_ROOT / "squares" / "black_square.png").as_posix()

INPUT_SIZE = (256, 256)        # UNet input (width, height)
BOARD_SIZE = (512, 512)        # rectified board
PIECE_SIZE = (64, 64)          # one square

# class index -> FEN symbol; "f" marks an empty square
LABEL_NAMES = list("BKNPQRbknpqr") + ["f"]
NUM_CLASSES = len(LABEL_NAMES)
LABEL_INDICES = {name: i for i, name in enumerate(LABEL_NAMES)}
_PIECE_WORDS = {"B": "Bishop", "K": "King", "N": "Knight", "P": "Pawn", "Q": "Queen", "R": "Rook"}
LABEL_DESCRIPTIONS = ([f"White {_PIECE_WORDS[s]}" for s in "BKNPQR"] + [f"Black {_PIECE_WORDS[s]}" for s in "BKNPQR"] +
                      ["Empty Square", "Unknown"])
SEGMENTATION_MAP = {0: "background", 255: "chessboard"}

FILES, RANKS = "abcdefgh", "12345678"
# classifier output order seen from White's side: a8..h8, a7..h7, ..., a1..h1; and the 180-degree view
SQUARE_NAMES_NORMAL = [f + r for r in reversed(RANKS) for f in FILES]
SQUARE_NAMES_FLIPPED = list(reversed(SQUARE_NAMES_NORMAL))
DARK_SQUARES = {f + r for fi, f in enumerate(FILES) for ri, r in enumerate(RANKS) if (fi + ri) % 2 == 0}
INVALID_PAWN_SQUARES = {f + r for f in FILES for r in "18"}
