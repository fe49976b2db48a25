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

    getPerspectiveTransform + warpPerspective   utils.py:115-132                    perspective_matrix, warp_perspective

NOT restated independently (the end-to-end oracle takes it from the product's numpy host path and says so): the contour chain
(findContours / contourArea / boundingRect / arcLength / approxPolyDP -- pinned instead on the reference's own 631 label masks,
tests/test_contour_cpp.py).
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


def perspective_matrix(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.getPerspectiveTransform: the 3x3 map with m33 = 1 taking four ``src`` points onto ``dst``.  OpenCV sets up the 8x8 system
    [x y 1 0 0 0 -xu -yu; 0 0 0 x y 1 -xv -yv] h = [u; v] and solves it in double precision; here by Gaussian elimination with
    partial pivoting, written out (no library solver shared with the product)."""
    s4 = np.asarray(src, np.float64).reshape(4, 2)
    d4 = np.asarray(dst, np.float64).reshape(4, 2)
    a = [[0.0] * 9 for _ in range(8)]
    for i in range(4):
        x, y = s4[i]
        u, v = d4[i]
        a[2 * i] = [x, y, 1.0, 0.0, 0.0, 0.0, -x * u, -y * u, u]
        a[2 * i + 1] = [0.0, 0.0, 0.0, x, y, 1.0, -x * v, -y * v, v]
    for col in range(8):
        piv = max(range(col, 8), key=lambda r: abs(a[r][col]))
        a[col], a[piv] = a[piv], a[col]
        for r in range(col + 1, 8):
            f = a[r][col] / a[col][col]
            for c in range(col, 9):
                a[r][c] -= f * a[col][c]
    h = [0.0] * 8
    for r in range(7, -1, -1):
        h[r] = (a[r][8] - sum(a[r][c] * h[c] for c in range(r + 1, 8))) / a[r][r]
    return np.array(h + [1.0], np.float64).reshape(3, 3)


def _invert3(m: np.ndarray) -> np.ndarray:
    (a, b, c), (d, e, f), (g, h, i) = m
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    adj = np.array([[e * i - f * h, c * h - b * i, b * f - c * e],
                    [f * g - d * i, a * i - c * g, c * d - a * f],
                    [d * h - e * g, b * g - a * h, a * e - b * d]], np.float64)
    return adj / det


def warp_perspective(image: np.ndarray, m: np.ndarray, size: tuple[int, int]) -> np.ndarray:
    """cv2.warpPerspective(image, M, (w, h)) with its defaults (INTER_LINEAR, BORDER_CONSTANT 0), in OpenCV's fixed-point form:
    the inverse map gives source coordinates in 1/32 pixel (``INTER_BITS = 5``: X = round(32 * X0 / W0)), the four bilinear weights
    are the integers (32-a)(32-b)*32 ... a*b*32 (``INTER_REMAP_COEF_BITS = 15``: they sum to 2^15 exactly) and the pixel is
    ``(sum(w * p) + 2^14) >> 15`` -- round half UP, where a float blend rounds ties to even.  Taps outside the image read 0."""
    w_out, h_out = size
    inv = _invert3(np.asarray(m, np.float64))
    img = image if image.ndim == 3 else image[:, :, None]
    h, w, ch = img.shape
    ys, xs = np.mgrid[0:h_out, 0:w_out].astype(np.float64)
    x0 = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    y0 = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    w0 = inv[2, 0] * xs + inv[2, 1] * ys + inv[2, 2]
    scale = np.where(w0 != 0, 32.0 / np.where(w0 != 0, w0, 1.0), 0.0)
    fx = np.clip(x0 * scale, -2.0 ** 31, 2.0 ** 31 - 1)
    fy = np.clip(y0 * scale, -2.0 ** 31, 2.0 ** 31 - 1)
    xi = np.rint(fx).astype(np.int64)                        # saturate_cast<int>(double): round to nearest, ties to even
    yi = np.rint(fy).astype(np.int64)
    sx, sy, ax, ay = xi >> 5, yi >> 5, xi & 31, yi & 31
    pad = np.zeros((h + 2, w + 2, ch), np.int64)
    pad[1:h + 1, 1:w + 1] = img

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return pad[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)] * ok[..., None]

    w00 = ((32 - ax) * (32 - ay) * 32)[..., None]
    w01 = (ax * (32 - ay) * 32)[..., None]
    w10 = ((32 - ax) * ay * 32)[..., None]
    w11 = (ax * ay * 32)[..., None]
    acc = w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1)
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out if image.ndim == 3 else out[:, :, 0]


def extract_board(image: np.ndarray, quad: np.ndarray, size: tuple[int, int] = (512, 512)) -> np.ndarray:
    """``utils.extract_perspective`` of the reference (utils.py:115-132): quadrangle (TR, TL, BL, BR order as the reference's
    ``_rotate_quadrangle`` leaves it) -> destination corners ((0,0), (w,0), (w,h), (0,h)) -> warp."""
    w, h = size
    dest = np.array(((0, 0), (w, 0), (w, h), (0, h)), np.float64)
    return warp_perspective(image, perspective_matrix(np.asarray(quad, np.float64).reshape(4, 2), dest), size)
