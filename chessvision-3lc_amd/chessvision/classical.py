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


def _area_linear_coeffs(src: int, dst: int):
    """One axis of the bilinear form cv2.resize falls back to when INTER_AREA has to ENLARGE (OpenCV ``resize.cpp``: "true area
    interpolation is only implemented for scale_x >= 1 && scale_y >= 1; in other cases it is emulated using some variant of bilinear
    interpolation"): source index ``s = floor(d * scale)``, fraction ``f = (float)((d + 1) - (s + 1) * inv_scale)`` reduced to
    [0, 1) (0 when not positive), both clamped at the last source pixel, and the 11-bit fixed-point pair
    ``(short)round((1 - f) * 2048), (short)round(f * 2048)`` (INTER_RESIZE_COEF_BITS).  Returns (s0, s1, a0, a1) as int64 arrays."""
    inv_scale = dst / src
    scale = 1.0 / inv_scale
    d = np.arange(dst, dtype=np.float64)
    s0 = np.floor(d * scale).astype(np.int64)
    f = ((d + 1.0) - (s0 + 1).astype(np.float64) * inv_scale).astype(np.float32)
    f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
    last = s0 >= src - 1
    f = np.where(last, np.float32(0), f).astype(np.float32)
    s0 = np.where(last, src - 1, s0)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s0, np.minimum(s0 + 1, src - 1), a0, a1


def resize_area(image: NDArray[np.uint8], size: tuple[int, int]) -> NDArray[np.uint8]:
    """INTER_AREA resize to ``size = (width, height)`` as cv2.resize computes it.

    Integer shrink factors in both directions: the exact box mean with round-half-up (OpenCV's ResizeAreaFast: ``(sum + n/2) /
    n``).  Any other shrink: ``ResizeArea_Invoker`` with float32 arithmetic in OpenCV's order of operations -- horizontal pass
    ``buf = buf + S * alpha`` over the table entries of a destination column, vertical pass ``sum = beta * buf`` for the first source
    row and ``sum += beta * buf`` for the others, round half to even -- so that host, device (``pipeline.hip``) and the independent
    oracle agree bit for bit (round 4; before, a coverage-weighted mean in double: one grey level apart on 2.5 % of the pixels).
    Enlarging in either direction (a photo smaller than the target): OpenCV's fixed-point bilinear path with the AREA coefficient
    rule (``_area_linear_coeffs``): horizontal pass ``S[s0] * a0 + S[s1] * a1`` in int32, vertical pass
    ``(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2`` (``VResizeLinear`` for 8-bit images) -- exact integers, so
    host, device and the independent oracle agree byte for byte (round 5; before, a plain double-precision bilinear blend)."""
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
        x0, x1, a0, a1 = _area_linear_coeffs(w, w_out)
        y0, y1, b0, b1 = _area_linear_coeffs(h, h_out)
        src = img.astype(np.int64)
        rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]           # (h, w_out, c), 11 fractional bits
        acc = ((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16)
        out = ((acc + 2) >> 2).astype(np.uint8)
    return out if image.ndim == 3 else out[:, :, 0]


# ---- contours (cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_TC89_KCOS), core.py:360) ----------------
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


# OpenCV's direction codes (0 = east, then counter-clockwise on screen: NE, N, NW, W, SW, S, SE) by (dy, dx)
_CODE = {(0, 1): 0, (-1, 1): 1, (-1, 0): 2, (-1, -1): 3, (0, -1): 4, (1, -1): 5, (1, 0): 6, (1, 1): 7}
_ABS_DIFF = (1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1)


def _f32_bits(value: float) -> int:
    """Bit pattern of ``(float)value`` as a signed 32-bit integer (OpenCV compares its k-cosines through ``Cv32suf.i``)."""
    return int(np.array([value], dtype=np.float32).view(np.int32)[0])


def tc89_kcos(points: NDArray[np.int32]) -> NDArray[np.int32]:
    """``CHAIN_APPROX_TC89_KCOS`` as cv2.findContours applies it to a traced chain (OpenCV ``icvApproximateChainTC89``; reference
    call site core.py:360).  ``points`` (n, 2) = every border pixel in tracing order.  Pass 0 keeps the points where the chain
    code changes.  Pass 1 gives each its Teh-Chin region of support k -- grown while the chord p[i-k] p[i+k] lengthens and the
    ratio (distance of p[i] to the chord) / (chord length) rises -- and its k-cosine: cos of the angle p[i-j] p[i] p[i+j] plus 1.1,
    rounded to float32 and compared through its bit pattern, walked from j = k downwards while it grows.  Pass 2 suppresses points
    that have a larger measure within k/2 chain steps (a suppressed point counts as 0 for the points after it).  Pass 3 removes
    points of support 1 that do not beat both chain neighbours."""
    pts = np.asarray(points).reshape(-1, 2).astype(np.int64)
    n = len(pts)
    if n <= 1:
        return pts.astype(np.int32)
    px, py = pts[:, 0].tolist(), pts[:, 1].tolist()
    code = [_CODE[(py[(i + 1) % n] - py[i], px[(i + 1) % n] - px[i])] for i in range(n)]
    s = [_ABS_DIFF[code[i] - code[i - 1] + 7] for i in range(n)]
    kept = [i for i in range(n) if s[i] != 0]
    if not kept:
        return pts.astype(np.int32)
    k = [0] * n
    for i in kept:
        x0, y0 = px[i], py[i]
        kk, l, d_num = 1, 0, 0
        while True:
            i1, i2 = (i - kk) % n, (i + kk) % n
            dx, dy = px[i2] - px[i1], py[i2] - py[i1]
            lk = dx * dx + dy * dy
            dk_num = (x0 - px[i1]) * dy - (y0 - py[i1]) * dx
            d = d_num * lk - dk_num * l                       # exact integers; OpenCV's float cast keeps the sign
            if kk > 1 and (l >= lk or (d_num > 0 and d <= 0) or (d_num < 0 and d >= 0)):
                break
            d_num, l = dk_num, lk
            kk += 1
            if kk > n:
                return pts.astype(np.int32)                   # OpenCV asserts k <= len
        kk -= 1
        k[i] = kk
        sv = 0
        for j in range(kk, 0, -1):
            i1, i2 = (i - j) % n, (i + j) % n
            dx1, dy1, dx2, dy2 = px[i1] - x0, py[i1] - y0, px[i2] - x0, py[i2] - y0
            if (dx1 == 0 and dy1 == 0) or (dx2 == 0 and dy2 == 0):
                break
            cos = float(np.float32(float(dx1 * dx2 + dy1 * dy2) /
                                   float(np.sqrt(np.float64((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2))))))
            sk = _f32_bits(cos + 1.1)
            if j < kk and sk <= sv:
                break
            sv = sk
        s[i] = sv
    survivors = []
    for i in kept:                                            # pass 2
        k2 = k[i] >> 1
        if any(s[(i - j) % n] > s[i] or s[(i + j) % n] > s[i] for j in range(1, k2 + 1)):
            s[i] = 0
        else:
            survivors.append(i)
    final = []
    for i in survivors:                                       # pass 3
        if k[i] == 1 and (s[i] <= s[(i - 1) % n] or s[i] <= s[(i + 1) % n]):
            s[i] = 0
        else:
            final.append(i)
    return pts[final].astype(np.int32)


def find_contours(mask: NDArray[np.uint8], tc89: bool = True, with_holes_flag: bool = False):
    """``cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_TC89_KCOS)[0]`` (reference core.py:360; ``tc89=False``: CHAIN_APPROX_NONE):
    the outer border of every 8-connected foreground component and the border of every hole (4-connected background that does not
    reach the frame), traced from OpenCV's start pixel in OpenCV's direction, in OpenCV's order -- it links a new contour in front
    of its parent's children and lists the two-level tree in pre-order: outer borders from the LAST found in raster order to the
    first, each followed by its holes, last found first."""
    f = np.asarray(mask) != 0
    lab, n = ndimage.label(f, structure=np.ones((3, 3), dtype=bool))
    index = np.arange(f.size).reshape(f.shape)
    outer = ndimage.minimum_position(index, lab, index=np.arange(1, n + 1)) if n else []
    holes_of: dict[int, list[tuple[int, int]]] = {}
    blab, bn = ndimage.label(~f)
    if bn:
        frame = set(np.unique(np.concatenate([blab[0], blab[-1], blab[:, 0], blab[:, -1]]))) - {0}
        ids = [k for k in range(1, bn + 1) if k not in frame]
        if ids:
            for (y, x) in ndimage.minimum_position(index, blab, index=ids):      # raster order of the holes' first pixels
                holes_of.setdefault(int(lab[y, x - 1]), []).append((int(y), int(x)))
    out, flags = [], []
    for comp in range(n, 0, -1):
        y, x = outer[comp - 1]
        out.append(_trace_border(f, (int(y), int(x)), (int(y), int(x) - 1)))
        flags.append(False)
        for (hy, hx) in reversed(holes_of.get(comp, [])):   # (hy, hx) = first hole pixel; the pixel to its left is foreground
            out.append(_trace_border(f, (hy, hx - 1), (hy, hx)))
            flags.append(True)
    out = [(tc89_kcos(c) if tc89 else c).reshape(-1, 1, 2) for c in out]
    return (out, flags) if with_holes_flag else out


def contour_area(contour: NDArray[np.int32]) -> float:
    """cv2.contourArea: the shoelace sum in double (exact for pixel coordinates)."""
    p = contour.reshape(-1, 2).astype(np.float64)
    if len(p) == 0:
        return 0.0
    x, y = p[:, 0], p[:, 1]
    return float(abs(np.dot(np.roll(x, 1), y) - np.dot(np.roll(y, 1), x)) * 0.5)


def bounding_rect(contour: NDArray[np.int32]) -> tuple[int, int, int, int]:
    p = contour.reshape(-1, 2)
    x0, y0 = int(p[:, 0].min()), int(p[:, 1].min())
    return x0, y0, int(p[:, 0].max()) - x0 + 1, int(p[:, 1].max()) - y0 + 1


def arc_length(contour: NDArray[np.int32], closed: bool = True) -> float:
    """cv2.arcLength: OpenCV converts the points to Point2f and takes every segment's length in FLOAT (``std::sqrt`` of a float),
    summing in double -- for a closed curve starting with the segment from the last point to the first."""
    p = contour.reshape(-1, 2).astype(np.float32)
    if len(p) <= 1:
        return 0.0
    d = p - np.roll(p, 1, axis=0)
    if not closed:
        d = d[1:]
    seg = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
    total = 0.0
    for v in seg.tolist():                                    # sequential double sum, OpenCV's order
        total += v
    return total


def approx_poly_dp(contour: NDArray[np.int32], epsilon: float) -> NDArray[np.int32]:
    """cv2.approxPolyDP(contour, epsilon, closed=True) for integer points (OpenCV ``approxPolyDP_<int>``): (1) three farthest-point
    hops from point 0 pick the first split, (2) a stack of (start, end) slices drives Douglas-Peucker -- a slice's start point is
    emitted when nothing inside is farther than epsilon from its chord, else it splits at the farthest point, (3) ONE clean-up pass
    over the result, in place as OpenCV does it: a vertex within sqrt(0.5)*epsilon of the chord of its neighbours (chord not
    axis-parallel, vertex between them) is dropped and its successor kept without being examined."""
    src = [(int(x), int(y)) for x, y in np.asarray(contour).reshape(-1, 2)]
    count = len(src)
    if count == 0:
        return np.zeros((0, 1, 2), dtype=np.int32)
    eps = float(epsilon) * float(epsilon)
    pos, right_start, le_eps = 0, 0, False
    start = (-1000000, -1000000)
    for _ in range(3):
        max_dist = 0.0
        pos = (pos + right_start) % count
        start = src[pos]
        for j in range(1, count):
            pt = src[(pos + j) % count]
            dist = float((pt[0] - start[0]) ** 2 + (pt[1] - start[1]) ** 2)
            if dist > max_dist:
                max_dist, right_start = dist, j
        le_eps = max_dist <= eps
    dst: list[tuple[int, int]] = []
    stack: list[tuple[int, int]] = []
    if not le_eps:
        a = pos % count
        b = (right_start + a) % count
        stack += [(b, a), (a, b)]
    else:
        dst.append(start)
    split = right_start
    while stack:
        s, e = stack.pop()
        start, end = src[s], src[e]
        pos = (s + 1) % count
        if pos != e:
            max_dist = 0.0
            dx, dy = float(end[0] - start[0]), float(end[1] - start[1])
            while pos != e:
                pt = src[pos]
                dist = abs((pt[1] - start[1]) * dx - (pt[0] - start[0]) * dy)
                if dist > max_dist:
                    max_dist, split = dist, pos
                pos = (pos + 1) % count
            le = max_dist * max_dist <= eps * (dx * dx + dy * dy)
        else:
            le = True
        if le:
            dst.append(start)
        else:
            stack += [(split, e), (s, split)]
    cnt = new_count = len(dst)
    rpos = cnt - 1

    def read():
        nonlocal rpos
        p = dst[rpos]
        rpos = rpos + 1 if rpos + 1 < cnt else 0
        return p

    start = read()
    wpos = rpos
    pt = read()
    i = 0
    while i < cnt and new_count > 2:
        end = read()
        dx, dy = float(end[0] - start[0]), float(end[1] - start[1])
        dist = abs((pt[0] - start[0]) * dy - (pt[1] - start[1]) * dx)
        inner = (pt[0] - start[0]) * (end[0] - pt[0]) + (pt[1] - start[1]) * (end[1] - pt[1])
        if dist * dist <= 0.5 * eps * (dx * dx + dy * dy) and dx != 0 and dy != 0 and inner >= 0:
            new_count -= 1
            dst[wpos] = start = end
            wpos = wpos + 1 if wpos + 1 < cnt else 0
            pt = read()
            i += 2
            continue
        dst[wpos] = start = pt
        wpos = wpos + 1 if wpos + 1 < cnt else 0
        pt = end
        i += 1
    return np.array(dst[:new_count], dtype=np.int32).reshape(-1, 1, 2)


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


def warp_mode() -> str:
    """Which reading of cv2.warpPerspective the pipeline follows: ``fixed`` (default) = the classic fixed-point ``WarpPerspectiveInvoker`` +
    ``remapBilinear`` walk that every OpenCV release up to 4.10 runs for 8-bit images; ``float`` = the float-coordinate linear kernels that
    joined the 4.x line around 4.11 (the reference pins opencv-python 4.11.0.86): coordinates, weights and blend in float32.  Set by
    the environment variable CV_WARP, read by the host path here, the device kernel (csrc/pipeline.hip) and the oracle alike;
    INTEGRATION.md section D shows how a maintainer with cv2 installed finds out which one their build matches."""
    import os

    mode = os.environ.get("CV_WARP", "fixed").strip().lower() or "fixed"
    if mode not in ("fixed", "float"):
        raise ValueError(f"CV_WARP must be 'fixed' or 'float', not {mode!r}")
    return mode


def warp_perspective_float(image: NDArray[np.uint8], m: NDArray[np.float64], size: tuple[int, int]) -> NDArray[np.uint8]:
    """The ``float`` reading (see ``warp_mode``): the inverse of M (double, cofactor form) is rounded to float32; per destination pixel
    w = (x*M6 + y*M7) + M8, sx = ((x*M0 + y*M1) + M2) / w, sy likewise -- every operation rounded to float32, no fused multiply-add;
    ix = floor(sx), a = sx - ix (same for y); the four taps (BORDER_CONSTANT 0 outside the image) blend as v0 = p00 + a*(p01 - p00),
    v1 = p10 + a*(p11 - p10), v = v0 + b*(v1 - v0) in float32; the pixel is v rounded half to even and saturated."""
    w_out, h_out = size
    f = np.float32
    inv = invert3(m).astype(np.float32)
    xs = np.arange(w_out, dtype=np.float32)[None, :]
    ys = np.arange(h_out, dtype=np.float32)[:, None]
    with np.errstate(all="ignore"):
        den = (xs * inv[2, 0] + ys * inv[2, 1]).astype(f) + inv[2, 2]
        sx = (((xs * inv[0, 0] + ys * inv[0, 1]).astype(f) + inv[0, 2]).astype(f) / den).astype(f)
        sy = (((xs * inv[1, 0] + ys * inv[1, 1]).astype(f) + inv[1, 2]).astype(f) / den).astype(f)
        ok = np.isfinite(sx) & np.isfinite(sy) & (np.abs(sx) < 1e9) & (np.abs(sy) < 1e9)
        sx = np.where(ok, sx, f(-4.0)).astype(f)
        sy = np.where(ok, sy, f(-4.0)).astype(f)
    fx, fy = np.floor(sx), np.floor(sy)
    a, b = (sx - fx).astype(f)[..., None], (sy - fy).astype(f)[..., None]
    ix, iy = fx.astype(np.int64), fy.astype(np.int64)
    img = image if image.ndim == 3 else image[:, :, None]
    h, w, c = img.shape
    padded = np.zeros((h + 2, w + 2, c), dtype=np.float32)
    padded[1:-1, 1:-1] = img

    def tap(yy, xx):
        inside = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return padded[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)] * inside[..., None].astype(f)

    p00, p01, p10, p11 = tap(iy, ix), tap(iy, ix + 1), tap(iy + 1, ix), tap(iy + 1, ix + 1)
    v0 = (p00 + (a * (p01 - p00).astype(f)).astype(f)).astype(f)
    v1 = (p10 + (a * (p11 - p10).astype(f)).astype(f)).astype(f)
    v = (v0 + (b * (v1 - v0).astype(f)).astype(f)).astype(f)
    out = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return out if image.ndim == 3 else out[:, :, 0]


def warp_perspective(image: NDArray[np.uint8], m: NDArray[np.float64], size: tuple[int, int], mode: str | None = None) -> NDArray[np.uint8]:
    """cv2.warpPerspective(image, M, size) with its defaults (INTER_LINEAR, BORDER_CONSTANT 0); ``mode`` (default: ``warp_mode()``) picks
    the reading -- ``float``: ``warp_perspective_float``; ``fixed``: OpenCV's fixed-point form
    (``imgwarp.cpp``: WarpPerspectiveInvoker + remapBilinear).  The destination is walked in blocks of min(1024 / min(16, h), w)
    columns (BLOCK_SZ = 32: 64 x 16 blocks on a 512-px board; the 128 x 32 blocks belong to warpAffine); with
    ``blk`` the block's first column and ``x1`` the column inside it: X0 = M0*blk + M1*y + M2, W = W0 + M6*x1, W = 32 / W (0 if
    W == 0), X = round_half_even(clamp((X0 + M0*x1) * W)) -- source coordinates in 1/32 pixel (``INTER_BITS = 5``); the integer
    pixel X >> 5 saturates to int16; integer bilinear weights (32-a)(32-b)*32 ... a*b*32 that sum to 2^15
    (``INTER_REMAP_COEF_BITS``), pixel = (sum + 2^14) >> 15, i.e. round half UP; taps outside the image read 0."""
    if (mode or warp_mode()) == "float":
        return warp_perspective_float(image, m, size)
    w_out, h_out = size
    inv = invert3(m)
    bh = min(16, h_out)                                     # WarpPerspectiveInvoker: BLOCK_SZ = 32, bh0 = min(BLOCK_SZ / 2, height),
    bw = min(1024 // bh, w_out)                             # bw0 = min(BLOCK_SZ * BLOCK_SZ / bh0, width): 64 columns for the 512-px board
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
