"""Classical-CV stages between the two CNNs, restated on numpy (no OpenCV in this image or on the GPU box).

These are the SURVEY.md section 8(f) "next" rows on the CPU side of the hot path; the reference calls OpenCV
for all of them (``chessvision/core.py:212,299-300,360,373-374,394,398``; ``chessvision/utils.py:131-132``).
OpenCV 4.11 is not importable here, so nothing below can be bit-pinned against it; each function states the
OpenCV behaviour it follows, and ``tests/test_classical.py`` pins the mask -> quadrangle chain on the reference's
own fixtures (``data/board_extraction/masks/*.png`` against ``coordinates.json``).
"""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray
from scipy import ndimage

# 8-neighbourhood in clockwise order starting at west, (dy, dx) in image coordinates (y down)
_NBR = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]
_NBR_INDEX = {d: i for i, d in enumerate(_NBR)}


# ---- resize (cv2.resize(..., interpolation=cv2.INTER_AREA), core.py:212) --------------------------------
def _area_tab(src: int, dst: int):
    """OpenCV's ``computeResizeAreaTab`` for one axis: per destination index up to ``n`` (source index, float32 weight) pairs --
    a leading partial cell when it covers more than 1e-3 of a pixel, the whole cells at 1 / cellWidth, a trailing partial cell.
    Returned as dense arrays (dst, n) padded with weight-0 entries plus the per-row entry count."""
    scale = src / dst
    rows = []
    for d in range(dst):
        lo = d * scale
        hi = lo + scale
        cell = min(scale, src - lo)
        s1, s2 = int(np.ceil(lo)), int(np.floor(hi))
        s2 = min(s2, src - 1)
        s1 = min(s1, s2)
        ent = []
        if s1 - lo > 1e-3:
            ent.append((s1 - 1, (s1 - lo) / cell))
        ent += [(sx, 1.0 / cell) for sx in range(s1, s2)]
        if hi - s2 > 1e-3:
            ent.append((s2, min(min(hi - s2, 1.0), cell) / cell))
        rows.append(ent)
    n = max(len(r) for r in rows)
    idx = np.zeros((dst, n), dtype=np.int64)
    wgt = np.zeros((dst, n), dtype=np.float32)
    cnt = np.array([len(r) for r in rows], dtype=np.int64)
    for d, r in enumerate(rows):
        for k, (sx, a) in enumerate(r):
            idx[d, k], wgt[d, k] = sx, np.float32(a)
    return idx, wgt, cnt


def _linear_weights(src: int, dst: int) -> NDArray[np.float64]:
    scale = src / dst
    w = np.zeros((dst, src), dtype=np.float64)
    for d in range(dst):
        f = (d + 0.5) * scale - 0.5
        i0 = int(np.floor(f))
        t = f - i0
        a, b = min(max(i0, 0), src - 1), min(max(i0 + 1, 0), src - 1)
        w[d, a] += 1.0 - t
        w[d, b] += t
    return w


def resize_area(image: NDArray[np.uint8], size: tuple[int, int]) -> NDArray[np.uint8]:
    """INTER_AREA resize to ``size = (width, height)`` as cv2.resize computes it.

    Integer shrink factors in both directions: the exact box mean with round-half-up (OpenCV's ResizeAreaFast: ``(sum + n/2) /
    n``).  Any other shrink: ``ResizeArea_Invoker`` with float32 arithmetic in OpenCV's order of operations -- horizontal pass
    ``buf = buf + S * alpha`` over the table entries of a destination column, vertical pass ``sum = beta * buf`` for the first source
    row and ``sum += beta * buf`` for the others, round half to even -- so that host, device (``pipeline.hip``) and the independent
    oracle agree bit for bit (round 4; before, a coverage-weighted mean in double: one grey level apart on 2.5 % of the pixels).
    Enlarging (an image smaller than the target) goes through a plain bilinear blend, an approximation of OpenCV's fixed-point
    bilinear path that INTER_AREA takes for scale < 1."""
    w_out, h_out = size
    img = image if image.ndim == 3 else image[:, :, None]
    h, w, _ = img.shape
    if (h, w) == (h_out, w_out):
        out = img.copy()
    elif h % h_out == 0 and w % w_out == 0:
        fy, fx = h // h_out, w // w_out
        acc = img.reshape(h_out, fy, w_out, fx, -1).astype(np.uint32).sum(axis=(1, 3))
        out = ((acc + (fy * fx) // 2) // (fy * fx)).astype(np.uint8)
    elif h_out <= h and w_out <= w:
        xi, xw, xc = _area_tab(w, w_out)
        yi, yw, yc = _area_tab(h, h_out)
        src = img.astype(np.float32)
        rows = np.zeros((h, w_out, img.shape[2]), dtype=np.float32)          # horizontal pass of every source row
        for k in range(xi.shape[1]):
            live = xc > k
            rows[:, live] = rows[:, live] + src[:, xi[live, k]] * xw[live, k][None, :, None]
        total = np.zeros((h_out, w_out, img.shape[2]), dtype=np.float32)
        for j in range(yi.shape[1]):
            live = yc > j
            term = yw[live, j][:, None, None] * rows[yi[live, j]]
            total[live] = term if j == 0 else total[live] + term
        out = np.clip(np.rint(total), 0, 255).astype(np.uint8)
    else:
        wy = _linear_weights(h, h_out)
        wx = _linear_weights(w, w_out)
        acc = np.einsum("ys,swc->ywc", wy, img.astype(np.float64))
        acc = np.einsum("xw,ywc->yxc", wx, acc)
        out = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out if image.ndim == 3 else out[:, :, 0]


# ---- contours (cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_*), core.py:360) -------------------------
def _trace_border(f: NDArray[np.bool_], start: tuple[int, int], prev: tuple[int, int]) -> NDArray[np.int32]:
    """Suzuki-Abe border following (step 3 of Algorithm 1) from ``start`` with ``prev`` the background pixel
    the raster scan came from.  Returns the border as (x, y) points in traversal order."""
    h, w = f.shape

    def at(y, x):
        return 0 <= y < h and 0 <= x < w and f[y, x]

    i, j = start
    # 3.1: clockwise around (i, j) starting from prev, find a foreground pixel
    k0 = _NBR_INDEX[(prev[0] - i, prev[1] - j)]
    first = None
    for s in range(8):
        dy, dx = _NBR[(k0 + s) % 8]
        if at(i + dy, j + dx):
            first = (i + dy, j + dx)
            break
    if first is None:
        return np.array([[j, i]], dtype=np.int32)
    pts = []
    i2, j2 = first
    i3, j3 = i, j
    while True:
        # 3.3: counter-clockwise around (i3, j3) starting after (i2, j2)
        k = _NBR_INDEX[(i2 - i3, j2 - j3)]
        for s in range(1, 9):
            dy, dx = _NBR[(k - s) % 8]
            if at(i3 + dy, j3 + dx):
                i4, j4 = i3 + dy, j3 + dx
                break
        pts.append((j3, i3))
        if (i4, j4) == (i, j) and (i3, j3) == first:
            break
        i2, j2 = i3, j3
        i3, j3 = i4, j4
    return np.array(pts, dtype=np.int32)


def find_contours(mask: NDArray[np.uint8]) -> list[NDArray[np.int32]]:
    """Outer borders of the 8-connected foreground components, then hole borders (RETR_CCOMP returns both
    levels as one flat list).  Every border pixel is returned (CHAIN_APPROX_NONE); the reference asks for
    TC89_KCOS chain compression, which only thins the point list that ``approx_poly_dp`` consumes next."""
    f = np.asarray(mask) != 0
    out: list[NDArray[np.int32]] = []
    lab, n = ndimage.label(f, structure=np.ones((3, 3), dtype=bool))
    if n:
        firsts = ndimage.minimum_position(np.arange(f.size).reshape(f.shape), lab, index=np.arange(1, n + 1))
        for (y, x) in firsts:
            out.append(_trace_border(f, (int(y), int(x)), (int(y), int(x) - 1)).reshape(-1, 1, 2))
    # holes: 4-connected background components that do not touch the frame
    blab, bn = ndimage.label(~f)
    if bn:
        frame = set(np.unique(np.concatenate([blab[0], blab[-1], blab[:, 0], blab[:, -1]]))) - {0}
        ids = [k for k in range(1, bn + 1) if k not in frame]
        if ids:
            firsts = ndimage.minimum_position(np.arange(f.size).reshape(f.shape), blab, index=ids)
            for (y, x) in firsts:       # (y, x) = first hole pixel; the pixel to its left is foreground
                out.append(_trace_border(f, (int(y), int(x) - 1), (int(y), int(x))).reshape(-1, 1, 2))
    return out


def contour_area(contour: NDArray[np.int32]) -> float:
    p = contour.reshape(-1, 2).astype(np.float64)
    if len(p) < 3:
        return 0.0
    x, y = p[:, 0], p[:, 1]
    return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) * 0.5)


def bounding_rect(contour: NDArray[np.int32]) -> tuple[int, int, int, int]:
    p = contour.reshape(-1, 2)
    x0, y0 = int(p[:, 0].min()), int(p[:, 1].min())
    return x0, y0, int(p[:, 0].max()) - x0 + 1, int(p[:, 1].max()) - y0 + 1


def arc_length(contour: NDArray[np.int32], closed: bool = True) -> float:
    p = contour.reshape(-1, 2).astype(np.float64)
    d = np.diff(np.vstack([p, p[:1]]) if closed else p, axis=0)
    return float(np.sqrt((d * d).sum(axis=1)).sum())


def approx_poly_dp(contour: NDArray[np.int32], epsilon: float) -> NDArray[np.int32]:
    """Douglas-Peucker for a CLOSED curve, following cv2.approxPolyDP's strategy: (1) three farthest-point hops
    from point 0 pick the initial split, (2) stack-driven refinement against ``epsilon``, (3) one clean-up sweep
    that drops vertices lying within sqrt(0.5)*epsilon of the chord between their neighbours."""
    src = contour.reshape(-1, 2).astype(np.int64)
    count = len(src)
    if count == 0:
        return np.zeros((0, 1, 2), dtype=np.int32)
    eps2 = float(epsilon) * float(epsilon)
    pos, right_start, le_eps = 0, 0, False
    for _ in range(3):
        pos = (pos + right_start) % count
        d = ((np.roll(src, -pos, axis=0) - src[pos]) ** 2).sum(axis=1)
        j = int(np.argmax(d[1:])) + 1 if count > 1 else 0
        right_start = j
        le_eps = float(d[j]) <= eps2
    if le_eps:
        return src[pos].reshape(1, 1, 2).astype(np.int32)
    a, b = pos % count, (right_start + pos) % count
    stack = [(b, a), (a, b)]
    dst: list[tuple[int, int]] = []
    while stack:
        s, e = stack.pop()
        start, end = src[s], src[e]
        if (s + 1) % count != e:
            idx = np.arange(s + 1, e if e > s else e + count) % count
            dx, dy = float(end[0] - start[0]), float(end[1] - start[1])
            dist = np.abs((src[idx, 1] - start[1]) * dx - (src[idx, 0] - start[0]) * dy)
            m = int(np.argmax(dist))
            le = float(dist[m]) ** 2 <= eps2 * (dx * dx + dy * dy)
            split = int(idx[m])
        else:
            le, split = True, s
        if le:
            dst.append((int(start[0]), int(start[1])))
        else:
            stack.append((split, e))
            stack.append((s, split))
    # clean-up sweep over the closed result: drop a vertex when it is (nearly) on the chord of its neighbours
    pts = list(dst)
    i = 0
    while len(pts) > 2 and i < len(pts):
        start, cur, end = pts[i - 1], pts[i], pts[(i + 1) % len(pts)]
        dx, dy = end[0] - start[0], end[1] - start[1]
        dist = abs((cur[0] - start[0]) * dy - (cur[1] - start[1]) * dx)
        inner = (cur[0] - start[0]) * (end[0] - cur[0]) + (cur[1] - start[1]) * (end[1] - cur[1])
        if dist * dist <= 0.5 * eps2 * (dx * dx + dy * dy) and dx != 0 and dy != 0 and inner >= 0:
            del pts[i]
        else:
            i += 1
    return np.array(pts, dtype=np.int32).reshape(-1, 1, 2)


# ---- perspective (cv2.getPerspectiveTransform + cv2.warpPerspective, utils.py:131-132) -----------------
# The last bit of the matrices decides which way a source coordinate that is an exact .5 tie in 1/32 pixels rounds, so these three
# functions follow the arithmetic OpenCV 4.x publishes operation by operation (IEEE double, no fused multiply-add); the native
# product path (csrc/homography.cpp, csrc/pipeline.hip) and the independent oracle (oracle/classical_ref.py) do the same and the
# tests require all of them to agree bit for bit.
def get_perspective_transform(src: NDArray[np.float32], dst: NDArray[np.float32]) -> NDArray[np.float64]:
    """3x3 homography mapping the four ``src`` points onto ``dst`` (h33 = 1), as cv2.getPerspectiveTransform computes it: rows i
    and i + 4 of an 8x8 system (the -x*u products in float32: Point2f operands), ``solve(.., DECOMP_LU)`` = OpenCV's own LUImpl at
    this size -- partial pivoting on the first largest magnitude, ``alpha = A[j][i] * (-1 / A[i][i])``, back substitution
    ``s -= A[i][k] * x[k]``, ``x[i] = s / A[i][i]``.  Degenerate points give the zero matrix."""
    s4 = np.asarray(src, dtype=np.float32).reshape(4, 2)
    d4 = np.asarray(dst, dtype=np.float32).reshape(4, 2)
    x, y, u, v = s4[:, 0], s4[:, 1], d4[:, 0], d4[:, 1]
    a = np.zeros((8, 8), dtype=np.float64)
    b = np.concatenate([u, v]).astype(np.float64)
    a[:4, 0] = a[4:, 3] = x
    a[:4, 1] = a[4:, 4] = y
    a[:4, 2] = a[4:, 5] = 1.0
    a[:4, 6], a[:4, 7], a[4:, 6], a[4:, 7] = -x * u, -y * u, -x * v, -y * v        # float32 products, widened on assignment
    for i in range(8):
        k = i + int(np.argmax(np.abs(a[i:, i])))                                   # argmax keeps the FIRST maximum, as `>` does
        if abs(a[k, i]) < np.finfo(np.float64).eps * 100:
            return np.zeros((3, 3), dtype=np.float64)
        if k != i:
            a[[i, k], i:] = a[[k, i], i:]
            b[[i, k]] = b[[k, i]]
        d = -1.0 / a[i, i]
        alpha = a[i + 1:, i] * d
        a[i + 1:, i + 1:] += alpha[:, None] * a[i, i + 1:][None, :]
        b[i + 1:] += alpha * b[i]
    for i in range(7, -1, -1):
        acc = b[i]
        for k in range(i + 1, 8):
            acc -= a[i, k] * b[k]
        b[i] = acc / a[i, i]
    return np.append(b, 1.0).reshape(3, 3)


def get_perspective_transforms(src: NDArray[np.float32], dst: NDArray[np.float32]) -> NDArray[np.float64]:
    """``get_perspective_transform`` for N quadrangles: src (N,4,2) -> (N,3,3) (the batched pipeline uses the native
    ``hip_backend.board_homographies`` instead; this is its readable checker)."""
    src = np.asarray(src, dtype=np.float32).reshape(-1, 4, 2)
    return np.stack([get_perspective_transform(q, dst) for q in src]) if len(src) else np.zeros((0, 3, 3))


def invert3(m: NDArray[np.float64]) -> NDArray[np.float64]:
    """cv::invert of a 3x3 double matrix (what cv2.warpPerspective applies to its argument): cofactors times the reciprocal of the
    determinant expanded along the first row; zeros when the determinant is 0."""
    (a, b, c), (d, e, f), (g, h, i) = np.asarray(m, dtype=np.float64).tolist()
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    if det == 0.0:
        return np.zeros((3, 3), dtype=np.float64)
    r = 1.0 / det
    return np.array([[(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r],
                     [(f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r],
                     [(d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r]], dtype=np.float64)


def warp_perspective(image: NDArray[np.uint8], m: NDArray[np.float64], size: tuple[int, int]) -> NDArray[np.uint8]:
    """cv2.warpPerspective(image, M, size) with its defaults (INTER_LINEAR, BORDER_CONSTANT 0) in OpenCV's fixed-point form
    (``imgwarp.cpp``: WarpPerspectiveInvoker + remapBilinear).  The destination is walked in blocks of min(128, w) columns; with
    ``blk`` the block's first column and ``x1`` the column inside it: X0 = M0*blk + M1*y + M2, W = W0 + M6*x1, W = 32 / W (0 if
    W == 0), X = round_half_even(clamp((X0 + M0*x1) * W)) -- source coordinates in 1/32 pixel (``INTER_BITS = 5``); the integer
    pixel X >> 5 saturates to int16; integer bilinear weights (32-a)(32-b)*32 ... a*b*32 that sum to 2^15
    (``INTER_REMAP_COEF_BITS``), pixel = (sum + 2^14) >> 15, i.e. round half UP; taps outside the image read 0."""
    w_out, h_out = size
    inv = invert3(m)
    bw = min(128, w_out)
    cols = np.arange(w_out)
    blk = ((cols // bw) * bw).astype(np.float64)[None, :]
    x1 = (cols % bw).astype(np.float64)[None, :]
    ys = np.arange(h_out, dtype=np.float64)[:, None]
    with np.errstate(all="ignore"):
        x0 = (inv[0, 0] * blk + inv[0, 1] * ys) + inv[0, 2]
        y0 = (inv[1, 0] * blk + inv[1, 1] * ys) + inv[1, 2]
        w0 = (inv[2, 0] * blk + inv[2, 1] * ys) + inv[2, 2]
        den = w0 + inv[2, 0] * x1
        scale = np.divide(32.0, den, out=np.zeros_like(den), where=den != 0)
        lim = (-2.0 ** 31, 2.0 ** 31 - 1)
        xi = np.rint(np.clip((x0 + inv[0, 0] * x1) * scale, *lim)).astype(np.int64)
        yi = np.rint(np.clip((y0 + inv[1, 0] * x1) * scale, *lim)).astype(np.int64)
    px, py = np.clip(xi >> 5, -32768, 32767), np.clip(yi >> 5, -32768, 32767)
    ax, ay = (xi & 31)[..., None], (yi & 31)[..., None]
    img = image if image.ndim == 3 else image[:, :, None]
    h, w, c = img.shape
    padded = np.zeros((h + 2, w + 2, c), dtype=np.int64)
    padded[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= h) & (xx >= -1) & (xx <= w)                    # the zero frame itself is "outside"
        return padded[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)] * ok[..., None]

    acc = ((32 - ax) * (32 - ay) * 32 * tap(py, px) + ax * (32 - ay) * 32 * tap(py, px + 1) +
           (32 - ax) * ay * 32 * tap(py + 1, px) + ax * ay * 32 * tap(py + 1, px + 1))
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out if image.ndim == 3 else out[:, :, 0]


def bgr_to_gray(image: NDArray[np.uint8]) -> NDArray[np.uint8]:
    """cv2.cvtColor(BGR2GRAY) for 8-bit images as OpenCV 4.x computes it (the reference pins opencv-python 4.11.0.86,
    ``uv.lock:2663``): 15 fractional bits, ``(3735 B + 19235 G + 9798 R + 2^14) >> 15`` (``color_rgb.simd.hpp``: ``BY15``,
    ``GY15``, ``RY15``, ``gray_shift = 15``).  OpenCV 3.x used 14 bits (1868 / 9617 / 4899, ``>> 14``), which is what rounds
    1-2 of this package did; the two agree except for one grey level on about one pixel in twenty.  OpenCV is not installed
    here, so the constants are restated from its source as published, not measured."""
    b, g, r = (image[..., i].astype(np.int32) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def flip_horizontal(image: NDArray[np.uint8]) -> NDArray[np.uint8]:
    return np.ascontiguousarray(image[:, ::-1])
