"""Independent CPU restatement of the model-facing classical stages  --  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Written from the reference's call sites and OpenCV's documented 8-bit arithmetic, sharing NO code with the product's host
path (``chessvision/classical.py``, ``chessvision/fen.py``) or its device / C++ path (``csrc/pipeline.hip``, ``position.cpp``):
a misreading of OpenCV or of the reference that both of those share would otherwise be invisible to the end-to-end check.
Deliberately the slow, literal form (loops and float64) -- small inputs only.

    stage                               reference call site                         here
    cv2.resize(INTER_AREA), 512 -> 256  chessvision/core.py:212                     resize_area_int (integer factors),
                                                                                    resize_area (any shrink: OpenCV's float32 table form),
                                                                                    resize_area_enlarge (a photo below 256 px: the
                                                                                    fixed-point bilinear path with AREA coefficients)
    sigmoid > threshold -> 0 / 255      core.py:273, utils.py:101-112               binary_mask
    cv2.cvtColor(BGR2GRAY)              core.py:299                                 bgr_to_gray
    cv2.flip(board, 1)                  core.py:300                                 flip_lr
    extract_squares                     core.py:419-439 (KAT tests/test_chessvision.py:119-146)   split_squares
    argmax -> python-chess board_fen    core.py:326-349                             placement
    pawn rule                           core.py:453-469, constants.py:88-106        pawn_rule

    getPerspectiveTransform + warpPerspective   utils.py:115-132                    perspective_matrix, warp_perspective

The contour chain (findContours / contourArea / boundingRect / arcLength / approxPolyDP, core.py:357-411) is restated in plain C:
``oracle/c_ref/contours_ref.c`` (binding ``oracle/contours_c.py``), also independent of the product.
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


def _area_table(ssize: int, dsize: int, scale: float):
    """OpenCV's ``computeResizeAreaTab`` (imgproc/src/resize.cpp): for every destination index the source indices it covers and
    their weights -- a leading partial cell (if it covers more than 1e-3 of a pixel), the whole cells at 1 / cellWidth, a trailing
    partial cell; weights computed in double and stored as FLOAT.  Returns a list (per destination index) of (source index, weight)."""
    tab = []
    for d in range(dsize):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, ssize - fs1)
        s1, s2 = int(np.ceil(fs1)), int(np.floor(fs2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        row = []
        if s1 - fs1 > 1e-3:
            row.append((s1 - 1, np.float32((s1 - fs1) / cell)))
        for sx in range(s1, s2):
            row.append((sx, np.float32(1.0 / cell)))
        if fs2 - s2 > 1e-3:
            row.append((s2, np.float32(min(min(fs2 - s2, 1.0), cell) / cell)))
        tab.append(row)
    return tab


def resize_area(image: np.ndarray, out_hw: tuple[int, int]) -> np.ndarray:
    """cv2.resize(image, (w, h), interpolation=INTER_AREA) for a SHRINK in both directions (core.py:212).  Integer factors in both
    directions are OpenCV's integer fast path (``resize_area_int``); anything else is ``ResizeArea_Invoker`` with float32 work type,
    restated with its order of operations: per source row a horizontal pass ``buf[dx] = buf[dx] + S[sx] * alpha`` over the table
    entries in order, then ``sum[dx] = beta * buf[dx]`` for the first source row of a destination row and ``sum[dx] += beta *
    buf[dx]`` for the others, and ``saturate_cast<uchar>(sum)`` = round half to even.  Every operation rounds to float32."""
    h, w, c = image.shape
    oh, ow = out_hw
    assert oh <= h and ow <= w, "INTER_AREA as a shrink only (enlarging goes through OpenCV's bilinear path)"
    if h % oh == 0 and w % ow == 0:
        return resize_area_int(image, out_hw)
    xtab, ytab = _area_table(w, ow, w / ow), _area_table(h, oh, h / oh)
    src = image.astype(np.float32)
    out = np.zeros((oh, ow, c), np.uint8)
    for dy in range(oh):
        total = None
        for sy, beta in ytab[dy]:
            buf = np.zeros((ow, c), np.float32)
            for dx in range(ow):
                acc = np.zeros(c, np.float32)
                for sx, alpha in xtab[dx]:
                    acc = (acc + src[sy, sx] * alpha).astype(np.float32)
                buf[dx] = acc
            total = (beta * buf).astype(np.float32) if total is None else (total + (beta * buf).astype(np.float32)).astype(np.float32)
        out[dy] = np.clip(np.rint(total), 0, 255).astype(np.uint8)
    return out


def _linear_area_axis(ssize: int, dsize: int):
    """The per-axis table cv2.resize builds for its bilinear path in AREA mode (imgproc/src/resize.cpp, the loop over dx / dy with
    ``area_mode = interpolation == INTER_AREA``), restated literally: sx = cvFloor(dx * scale); fx = (float)((dx + 1) - (sx + 1) *
    inv_scale); fx = fx <= 0 ? 0 : fx - cvFloor(fx); if sx >= ssize - 1: fx = 0, sx = ssize - 1; coefficients
    saturate_cast<short>((1 - fx) * 2048) and saturate_cast<short>(fx * 2048) (float products, round half to even)."""
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    tab = []
    for d in range(dsize):
        sx = int(np.floor(d * scale))
        fx = np.float32((d + 1) - (sx + 1) * inv_scale)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if sx >= ssize - 1:
            fx, sx = np.float32(0.0), ssize - 1
        c0 = int(np.rint(np.float32(np.float32(1.0) - fx) * np.float32(2048.0)))
        c1 = int(np.rint(fx * np.float32(2048.0)))
        tab.append((sx, min(max(c0, -32768), 32767), min(max(c1, -32768), 32767)))
    return tab


def resize_area_enlarge(image: np.ndarray, out_hw: tuple[int, int]) -> np.ndarray:
    """cv2.resize(image, (w, h), interpolation=INTER_AREA) when at least one direction ENLARGES (core.py:212 on a photo smaller than
    256 pixels).  OpenCV has no area algorithm for that case and runs its 8-bit bilinear resizer with the AREA coefficient rule
    (``_linear_area_axis``): ``HResizeLinear`` -- D[dx] = S[sx] * a0 + S[sx + 1] * a1 in int, the right neighbour not read where a1
    is forced to 0 -- over the two source rows sy, sy + 1 (clipped to the image), then ``VResizeLinear<uchar>``:
    dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  Pure integer arithmetic, one pixel at a time."""
    h, w, c = image.shape
    oh, ow = out_hw
    xtab, ytab = _linear_area_axis(w, ow), _linear_area_axis(h, oh)
    out = np.zeros((oh, ow, c), np.uint8)
    for dy in range(oh):
        sy, b0, b1 = ytab[dy]
        r0, r1 = image[min(sy, h - 1)], image[min(sy + 1, h - 1)]
        for dx in range(ow):
            sx, a0, a1 = xtab[dx]
            sx1 = sx + 1 if sx + 1 < w else sx
            for ch in range(c):
                d0 = int(r0[sx, ch]) * a0 + int(r0[sx1, ch]) * a1
                d1 = int(r1[sx, ch]) * a0 + int(r1[sx1, ch]) * a1
                out[dy, dx, ch] = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2
    return out


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
    """cv2.getPerspectiveTransform (imgproc/src/imgwarp.cpp): the 3x3 map with m33 = 1 taking four ``src`` points onto ``dst``.
    OpenCV fills rows i and i + 4 of an 8x8 system with [x y 1 0 0 0 -xu -yu] / [0 0 0 x y 1 -xv -yv] and calls
    ``solve(A, B, X, DECOMP_LU)``; below the LAPACK size threshold that is its own ``LUImpl`` (core/src/matrix_decomp.cpp): partial
    pivoting on the FIRST largest magnitude, elimination with ``alpha = A[j][i] * (-1 / A[i][i])`` and ``A[j][k] += alpha * A[i][k]``,
    back substitution ``s -= A[i][k] * x[k]``, ``x[i] = s / A[i][i]``.  The ORDER OF OPERATIONS is part of the result (the last bit of
    the matrix decides 1/32-pixel ties in the warp), so it is followed literally, in scalar Python floats (IEEE double, no fused
    multiply-add), sharing no code with the product.  A singular system returns the zero matrix (OpenCV leaves X untouched)."""
    s4 = np.asarray(src, np.float32).reshape(4, 2)           # Point2f: the products below are formed in FLOAT, as in OpenCV
    d4 = np.asarray(dst, np.float32).reshape(4, 2)
    a = [[0.0] * 8 for _ in range(8)]
    b = [0.0] * 8
    for i in range(4):
        x, y, u, v = s4[i][0], s4[i][1], d4[i][0], d4[i][1]   # np.float32 scalars
        xu, yu, xv, yv = float(-x * u), float(-y * u), float(-x * v), float(-y * v)
        a[i] = [float(x), float(y), 1.0, 0.0, 0.0, 0.0, xu, yu]
        a[i + 4] = [0.0, 0.0, 0.0, float(x), float(y), 1.0, xv, yv]
        b[i], b[i + 4] = float(u), float(v)
    eps = 2.220446049250313e-16 * 100.0
    for i in range(8):
        k = i
        for j in range(i + 1, 8):
            if abs(a[j][i]) > abs(a[k][i]):
                k = j
        if abs(a[k][i]) < eps:
            return np.zeros((3, 3), np.float64)
        if k != i:
            for j in range(i, 8):
                a[i][j], a[k][j] = a[k][j], a[i][j]
            b[i], b[k] = b[k], b[i]
        d = -1.0 / a[i][i]
        for j in range(i + 1, 8):
            alpha = a[j][i] * d
            for c in range(i + 1, 8):
                a[j][c] += alpha * a[i][c]
            b[j] += alpha * b[i]
    for i in range(7, -1, -1):
        s = b[i]
        for c in range(i + 1, 8):
            s -= a[i][c] * b[c]
        b[i] = s / a[i][i]
    return np.array(b + [1.0], np.float64).reshape(3, 3)


def _invert3(m: np.ndarray) -> np.ndarray:
    """cv::invert of a 3x3 double matrix (core/src/lapack.cpp, the closed-form branch warpPerspective takes for its matrix): the
    cofactors times the RECIPROCAL of the determinant, ``det3`` expanded along the first row.  Singular -> zeros, as OpenCV."""
    (a, b, c), (d, e, f), (g, h, i) = [[float(v) for v in row] for row in m]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    if det == 0.0:
        return np.zeros((3, 3), np.float64)
    r = 1.0 / det
    return np.array([[(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r],
                     [(f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r],
                     [(d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r]], np.float64)


def warp_perspective(image: np.ndarray, m: np.ndarray, size: tuple[int, int]) -> np.ndarray:
    """cv2.warpPerspective(image, M, (w, h)) with its defaults (INTER_LINEAR, BORDER_CONSTANT 0), in OpenCV's fixed-point form
    (imgwarp.cpp: WarpPerspectiveInvoker + remapBilinear).  The destination is walked in blocks of bw0 x bh0 pixels, bh0 = min(16, h), bw0 = min(1024 / bh0, w)
    (WarpPerspectiveInvoker's BLOCK_SZ is 32 -- 64 x 16 blocks on the 512-px board; BLOCK_SZ = 64 / 128 x 32 blocks are warpAffine's);
    for the block starting at column ``bx`` and the row ``y``: X0 = M0*bx + M1*y + M2 (likewise Y0, W0), and for the pixel ``x1``
    columns into the block W = W0 + M6*x1, W = 32 / W (0 when W is 0), X = round_half_even(clamp((X0 + M0*x1) * W, INT_MIN, INT_MAX))
    -- source coordinates in 1/32 pixel (``INTER_BITS = 5``).  The association (block start first, then the in-block column) is
    followed literally: the last bit decides ties.  The integer pixel X >> 5 saturates to int16 (the map is CV_16SC2), the four
    bilinear weights are the integers (32-a)(32-b)*32 ... a*b*32 (``INTER_REMAP_COEF_BITS = 15``: they sum to 2^15 exactly) and the
    pixel is ``(sum(w * p) + 2^14) >> 15`` -- round half UP.  Taps outside the image read 0."""
    w_out, h_out = size
    inv = _invert3(np.asarray(m, np.float64))
    img = image if image.ndim == 3 else image[:, :, None]
    h, w, ch = img.shape
    block_rows = min(32 // 2, h_out)                         # const int BLOCK_SZ = 32; bh0 = min(BLOCK_SZ/2, height)
    bw = min((32 * 32) // block_rows, w_out)                 # bw0 = min(BLOCK_SZ*BLOCK_SZ/bh0, width)
    cols = np.arange(w_out)
    bx = ((cols // bw) * bw).astype(np.float64)[None, :]              # block start column of every destination column
    x1 = (cols % bw).astype(np.float64)[None, :]
    ys = np.arange(h_out, dtype=np.float64)[:, None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        x0 = (inv[0, 0] * bx + inv[0, 1] * ys) + inv[0, 2]
        y0 = (inv[1, 0] * bx + inv[1, 1] * ys) + inv[1, 2]
        w0 = (inv[2, 0] * bx + inv[2, 1] * ys) + inv[2, 2]
        wq = w0 + inv[2, 0] * x1
        scale = np.where(wq != 0, 32.0 / np.where(wq != 0, wq, 1.0), 0.0)
        fx = np.maximum(-2147483648.0, np.minimum(2147483647.0, (x0 + inv[0, 0] * x1) * scale))
        fy = np.maximum(-2147483648.0, np.minimum(2147483647.0, (y0 + inv[1, 0] * x1) * scale))
    xi = np.rint(fx).astype(np.int64)                        # saturate_cast<int>(double): round to nearest, ties to even
    yi = np.rint(fy).astype(np.int64)
    sx = np.clip(xi >> 5, -32768, 32767)                     # saturate_cast<short>
    sy = np.clip(yi >> 5, -32768, 32767)
    ax, ay = xi & 31, yi & 31
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


def warp_perspective_float(image: np.ndarray, m: np.ndarray, size: tuple[int, int], rows: range | None = None) -> np.ndarray:
    """The OTHER reading of cv2.warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) for 8-bit images: the float-coordinate linear kernels
    of the recent 4.x line (around 4.11, the version the reference pins) instead of the fixed-point walk.  Restated as the formula
    they publish, one pixel at a time in numpy float32 scalars (each operation rounds to float32; no fused multiply-add): the inverse
    matrix (double) cast to float; w = x*M6 + y*M7 + M8; sx = (x*M0 + y*M1 + M2) / w; ix = floor(sx), alpha = sx - ix; likewise y; taps
    outside the image are 0; v = lerp(lerp(p00, p01, alpha), lerp(p10, p11, alpha), beta) with lerp(a, b, t) = a + t*(b - a); round half
    to even.  Which of the two readings an installed cv2 follows cannot be determined here; CV_WARP selects.
    ``rows``: compute only these destination rows (the others stay 0) -- the scalar loop takes ~15 s per 512 x 512 board."""
    w_out, h_out = size
    f = np.float32
    M = [f(v) for v in _invert3(np.asarray(m, np.float64)).reshape(9)]
    img = image if image.ndim == 3 else image[:, :, None]
    h, w, ch = img.shape
    out = np.zeros((h_out, w_out, ch), np.uint8)

    def px(yy, xx, c):
        return f(img[yy, xx, c]) if 0 <= yy < h and 0 <= xx < w else f(0.0)

    with np.errstate(all="ignore"):
        for y in (rows if rows is not None else range(h_out)):
            fy_ = f(y)
            for x in range(w_out):
                fx_ = f(x)
                den = f(f(fx_ * M[6]) + f(fy_ * M[7])) + M[8]
                sx = f(f(f(fx_ * M[0]) + f(fy_ * M[1])) + M[2]) / den
                sy = f(f(f(fx_ * M[3]) + f(fy_ * M[4])) + M[5]) / den
                if not (np.isfinite(sx) and np.isfinite(sy)) or abs(sx) >= 1e9 or abs(sy) >= 1e9:
                    continue                                  # maps nowhere: border value
                ix, iy = int(np.floor(sx)), int(np.floor(sy))
                al, be = f(sx - f(ix)), f(sy - f(iy))
                for c in range(ch):
                    p00, p01, p10, p11 = px(iy, ix, c), px(iy, ix + 1, c), px(iy + 1, ix, c), px(iy + 1, ix + 1, c)
                    top = f(p00 + f(al * f(p01 - p00)))
                    bot = f(p10 + f(al * f(p11 - p10)))
                    v = f(top + f(be * f(bot - top)))
                    out[y, x, c] = int(min(255.0, max(0.0, np.rint(v))))
    return out if image.ndim == 3 else out[:, :, 0]


def extract_board(image: np.ndarray, quad: np.ndarray, size: tuple[int, int] = (512, 512), mode: str = "fixed") -> np.ndarray:
    """``utils.extract_perspective`` of the reference (utils.py:115-132): quadrangle (TR, TL, BL, BR order as the reference's
    ``_rotate_quadrangle`` leaves it) -> destination corners ((0,0), (w,0), (w,h), (0,h)) -> warp."""
    w, h = size
    dest = np.array(((0, 0), (w, 0), (w, h), (0, h)), np.float64)
    warp = warp_perspective_float if mode == "float" else warp_perspective
    return warp(image, perspective_matrix(np.asarray(quad, np.float32).reshape(4, 2), dest), size)
