"""Independent CPU restatement of the model-facing classical stages  --  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Written from the reference's call sites and OpenCV's documented 8-bit arithmetic, sharing NO code with the product's host
path (``chessvision/classical.py``, ``chessvision/fen.py``) or its device / C++ path (``csrc/pipeline.hip``, ``position.cpp``):
a misreading of OpenCV or of the reference that both of those share would otherwise be invisible to the end-to-end check.
Deliberately the slow, literal form (loops and float64) -- small inputs only.

    stage                               reference call site                         here
    cv2.resize(INTER_AREA), 512 -> 256  chessvision/core.py:212                     resize_area_int
    sigmoid > threshold -> 0 / 255      core.py:273, utils.py:101-112               binary_mask
    cv2.cvtColor(BGR2GRAY)              core.py:299                                 bgr_to_gray
    cv2.flip(board, 1)                  core.py:300                                 flip_lr
    extract_squares                     core.py:419-439 (KAT tests/test_chessvision.py:119-146)   split_squares
    argmax -> python-chess board_fen    core.py:326-349                             placement
    pawn rule                           core.py:453-469, constants.py:88-106        pawn_rule

NOT restated independently (the end-to-end oracle takes them from the product's numpy host path and says so): the contour
chain (findContours / contourArea / boundingRect / arcLength / approxPolyDP -- pinned instead on the reference's own 631
label masks, tests/test_contour_cpp.py) and the perspective warp (getPerspectiveTransform / warpPerspective).
"""
from __future__ import annotations

import numpy as np

LABELS = ["B", "K", "N", "P", "Q", "R", "b", "k", "n", "p", "q", "r", "f"]          # reference constants.py:23 ("f" = empty)
FILES = "abcdefgh"


def square_names(flip: bool) -> list[str]:
    """Reading order of the 64 crops: a8..h8, a7..h7, ..., a1..h1; flipped boards h1..a1, ..., h8..a8 (constants.py:109-129)."""
    if not flip:
        return [FILES[f] + str(r) for r in range(8, 0, -1) for f in range(8)]
    return [FILES[f] + str(r) for r in range(1, 9) for f in range(7, -1, -1)]


def resize_area_int(image: np.ndarray, out_hw: tuple[int, int]) -> np.ndarray:
    """INTER_AREA for an integer shrink factor: every output pixel is the mean of its fy x fx box, rounded half up
    (OpenCV's integer fast path computes (sum + area/2) / area in integers)."""
    h, w, c = image.shape
    oh, ow = out_hw
    assert h % oh == 0 and w % ow == 0, "independent restatement covers integer factors only"
    fy, fx = h // oh, w // ow
    total = np.zeros((oh, ow, c), np.int64)
    for dy in range(fy):                                   # one strided view per position inside the box
        for dx in range(fx):
            total += image[dy::fy, dx::fx]
    return ((2 * total + fy * fx) // (2 * fy * fx)).astype(np.uint8)


def binary_mask(logits: np.ndarray, threshold: float) -> np.ndarray:
    """sigmoid in float32 (torch.sigmoid on the float32 logits, core.py:273), then > threshold -> 255 else 0."""
    z = logits.astype(np.float32)
    prob = (np.float32(1.0) / (np.float32(1.0) + np.exp(-z, dtype=np.float32))).astype(np.float32)
    return np.where(prob > np.float32(threshold), 255, 0).astype(np.uint8)


def bgr_to_gray(image: np.ndarray) -> np.ndarray:
    """8-bit BGR2GRAY of OpenCV 4.x (the reference pins opencv-python 4.11.0.86, uv.lock:2663): fixed point with 15 fractional
    bits, Y = (3735 B + 19235 G + 9798 R + 2^14) >> 15  (modules/imgproc/src/color_rgb.simd.hpp: BY15 / GY15 / RY15,
    gray_shift = 15).  OpenCV 3.x used 14 bits (1868 / 9617 / 4899): the two differ by one grey level on ~1 pixel in 20.
    Computed here as the rounded rational so that nothing but the three constants is shared with the product."""
    b = image[..., 0].astype(np.float64)
    g = image[..., 1].astype(np.float64)
    r = image[..., 2].astype(np.float64)
    return np.floor((3735.0 * b + 19235.0 * g + 9798.0 * r) / 32768.0 + 0.5).astype(np.uint8)


def flip_lr(board: np.ndarray) -> np.ndarray:
    out = np.empty_like(board)
    w = board.shape[1]
    for x in range(w):
        out[:, x] = board[:, w - 1 - x]
    return out


def split_squares(board: np.ndarray) -> np.ndarray:
    """(H, W) -> (64, H/8, W/8, 1): rank 8 first, file a first inside a rank."""
    h, w = board.shape
    sh, sw = h // 8, w // 8
    out = np.zeros((64, sh, sw, 1), board.dtype)
    for row in range(8):
        for col in range(8):
            out[row * 8 + col, :, :, 0] = board[row * sh:(row + 1) * sh, col * sw:(col + 1) * sw]
    return out


def placement(labels: list[str], names: list[str]) -> str:
    """FEN piece placement (python-chess ``BaseBoard.board_fen``): ranks 8 -> 1 separated by '/', files a -> h, runs of
    empty squares as digits."""
    at = {name: lab for lab, name in zip(labels, names)}
    ranks = []
    for rank in range(8, 0, -1):
        text, run = "", 0
        for f in FILES:
            lab = at.get(f + str(rank), "f")
            if lab == "f":
                run += 1
                continue
            if run:
                text += str(run)
                run = 0
            text += lab
        if run:
            text += str(run)
        ranks.append(text)
    return "/".join(ranks)


def pawn_rule(labels: list[str], probs: np.ndarray, names: list[str]):
    """A pawn on rank 1 or 8 becomes the most probable non-pawn class (core.py:453-469).  ``np.argsort`` order decides ties, as
    in the reference: ascending stable sort, walked from the end.  Returns (new labels, [(square, old, new)])."""
    out = list(labels)
    fixes = []
    for i, (lab, name) in enumerate(zip(labels, names)):
        if lab not in ("P", "p") or name[1] not in ("1", "8"):
            continue
        order = np.argsort(probs[i])
        for idx in order[::-1]:
            if LABELS[idx] not in ("P", "p"):
                out[i] = LABELS[idx]
                fixes.append((name, lab, LABELS[idx]))
                break
    return out, fixes
