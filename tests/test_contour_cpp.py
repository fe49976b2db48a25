"""CPU: the C++ mask -> quadrangle stage (csrc/contour.cpp, cv_find_quadrangle) against the numpy restatement
(chessvision/classical.py via ChessVision._find_quadrangle), which is itself pinned on the reference's mask/corner
fixtures in test_classical.py.  The two must agree exactly."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

from chessvision.core import ChessVision

G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def find_quadrangle():
    import __graft_entry__ as ge

    ge.build()
    from chessvision.hip_backend import find_quadrangle as fq

    return fq


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return np.array_equal(np.asarray(a).reshape(4, 2), np.asarray(b).reshape(4, 2))


@pytest.mark.parametrize("name", sorted(json.load(open(G / "masks" / "corners.json"))))
def test_reference_masks(find_quadrangle, name):
    mask = np.array(Image.open(G / "masks" / f"{name}.png").convert("L"))
    got = find_quadrangle(mask)
    assert got is not None and got.dtype == np.int32 and got.shape == (4, 1, 2)
    assert _same(got, ChessVision._find_quadrangle(mask))


def _polygon_mask(rng, size=256):
    """Random convex-ish quadrilateral (plus optional speck / hole / second blob) rasterised by half-plane tests."""
    c = np.array([size / 2, size / 2]) + rng.uniform(-10, 10, 2)
    half = rng.uniform(80, 105)
    base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64) * half
    pts = c + base + rng.uniform(-22, 22, (4, 2))
    yy, xx = np.mgrid[0:size, 0:size]
    inside = np.ones((size, size), bool)
    x, y = pts[:, 0], pts[:, 1]
    orient = np.sign(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    for i in range(4):
        a, b = pts[i], pts[(i + 1) % 4]
        inside &= orient * ((b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])) >= 0
    mask = np.where(inside, 255, 0).astype(np.uint8)
    kind = rng.integers(0, 4)
    if kind == 1:
        y, x = rng.integers(0, size - 6, 2)
        mask[y:y + 5, x:x + 5] = 255                       # speck -> exercises the >1-contour filter
    elif kind == 2:
        cy, cx = int(c[1]), int(c[0])
        mask[cy - 4:cy + 4, cx - 4:cx + 4] = 0            # hole -> RETR_CCOMP inner border
    elif kind == 3:
        mask[rng.random((size, size)) < 0.002] = 255       # salt noise
    return mask


def test_random_masks_agree_with_numpy_restatement(find_quadrangle):
    rng = np.random.default_rng(123)
    found = 0
    for _ in range(60):
        mask = _polygon_mask(rng)
        a, b = find_quadrangle(mask), ChessVision._find_quadrangle(mask)
        assert _same(a, b), (a, b)
        found += a is not None
    assert found >= 20


def test_batched_threads_equal_single(find_quadrangle):
    from chessvision.hip_backend import find_quadrangles

    rng = np.random.default_rng(9)
    masks = np.stack([_polygon_mask(rng) for _ in range(24)] + [np.where(rng.random((256, 256)) < 0.5, 255, 0).astype(np.uint8)])
    many = find_quadrangles(masks, n_threads=4)
    for m, q in zip(masks, many):
        assert _same(q, find_quadrangle(m))
    assert _same(many[-1], ChessVision._find_quadrangle(masks[-1]))      # pure noise: thousands of specks
    assert find_quadrangles(np.zeros((0, 8, 8), np.uint8)) == []


def test_degenerate_masks(find_quadrangle):
    assert find_quadrangle(np.zeros((256, 256), np.uint8)) is None
    full = np.full((64, 64), 255, np.uint8)
    assert _same(find_quadrangle(full), ChessVision._find_quadrangle(full))
    one = np.zeros((32, 32), np.uint8)
    one[5, 7] = 255
    assert find_quadrangle(one) is None and ChessVision._find_quadrangle(one) is None


def test_all_reference_label_masks_give_their_annotated_corners(find_quadrangle):
    """Every one of the reference's 631 label masks (data/board_extraction/masks, bit-packed in golden/masks_all.npz by
    make_golden.py) against its annotated corners (coordinates.json): all four corners within 2 px, each matched to a
    different annotation, order top-right, top-left, bottom-left, bottom-right.  The numpy restatement must agree
    bit for bit on a sample of them."""
    from chessvision.hip_backend import find_quadrangles

    z = np.load(G / "masks_all.npz")
    masks = (np.unpackbits(z["bits"], axis=-1)[..., :256] * 255).astype(np.uint8)
    corners = z["corners"] * 256.0
    assert masks.shape == (631, 256, 256) and corners.shape == (631, 4, 2)
    quads = find_quadrangles(masks)
    worst = 0.0
    for q, ann in zip(quads, corners):
        assert q is not None
        pts = q.reshape(4, 2).astype(np.float64)
        d = np.sqrt(((pts[:, None, :] - ann[None, :, :]) ** 2).sum(-1))
        assert sorted(d.argmin(axis=1).tolist()) == [0, 1, 2, 3]
        worst = max(worst, float(d.min(axis=1).max()))
        cx, cy = pts.mean(axis=0)
        assert [("T" if y < cy else "B") + ("L" if x < cx else "R") for x, y in pts] == ["TR", "TL", "BL", "BR"]
    assert worst <= 2.0, worst
    for i in range(0, 631, 40):
        assert _same(quads[i], ChessVision._find_quadrangle(masks[i]))


# ---- borderline cases of the reference's filter / simplification rules (core.py:357-411), C++ == numpy restatement --------------
# Both implementations follow the reference's call (core.py:360: RETR_CCOMP, CHAIN_APPROX_TC89_KCOS) since round 5: the border is
# traced pixel by pixel, then compressed to its Teh-Chin dominant points, and contourArea / arcLength / approxPolyDP see the
# compressed chain (csrc/contour.cpp, classical.py; against the independent plain-C oracle in tests/test_contour_parity.py).  The
# cases below sit where the filter / simplification rules are borderline and pin what the C++ and the numpy form do there.
def _blank(size=256):
    return np.zeros((size, size), np.uint8)


def _both(find_quadrangle, mask):
    a, b = find_quadrangle(mask), ChessVision._find_quadrangle(mask)
    assert _same(a, b)
    return a


def test_area_share_threshold_applies_only_with_several_contours(find_quadrangle):
    size = 256
    for side, expect in ((154, True), (150, False)):         # 154^2 / 256^2 = 0.362 >= 0.35 ; 150^2 / 256^2 = 0.343 < 0.35
        m = _blank(size)
        m[20:20 + side, 30:30 + side] = 255
        m[230:236, 230:236] = 255                             # a second contour switches the area / box filter on
        assert (_both(find_quadrangle, m) is not None) == expect
        m[230:236, 230:236] = 0                               # alone, the same small square is accepted: no filter with one contour
        assert _both(find_quadrangle, m) is not None


def test_box_ratio_threshold(find_quadrangle):
    for h, w, expect in ((170, 250, True), (146, 250, False)):    # 170/250 = 0.68 >= 0.6 ; 146/250 = 0.584 < 0.6 (area share 0.56)
        m = _blank()
        m[10:10 + h, 3:3 + w] = 255
        m[240:246, 120:126] = 255
        assert (_both(find_quadrangle, m) is not None) == expect


def test_cut_corner_becomes_a_fifth_vertex_only_when_it_exceeds_epsilon(find_quadrangle):
    yy, xx = np.mgrid[0:256, 0:256]
    for cut, expect in ((40, True), (120, False)):            # perimeter ~800 -> epsilon ~80 px: a 40 px bevel is absorbed, a 120 px one is not
        m = _blank()
        m[28:228, 28:228] = 255
        m[(xx - 28) + (yy - 28) < cut] = 0                    # bevel the top-left corner
        q = _both(find_quadrangle, m)
        assert (q is not None) == expect


def test_rotated_board_vertex_order(find_quadrangle):
    """A board rotated by ~30 degrees: four vertices, list starts at the right-most of the two top corners and runs
    counter-clockwise on screen (reference _rotate_quadrangle, core.py:399-411)."""
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float64)
    t = np.deg2rad(30.0)
    u = (xx - 128) * np.cos(t) + (yy - 128) * np.sin(t)
    v = -(xx - 128) * np.sin(t) + (yy - 128) * np.cos(t)
    m = np.where((np.abs(u) < 80) & (np.abs(v) < 80), 255, 0).astype(np.uint8)
    q = _both(find_quadrangle, m).reshape(4, 2)
    assert q[0, 0] >= q[2, 0]                                  # the rotation rule's postcondition
    cross = sum(q[i, 0] * q[(i + 1) % 4, 1] - q[(i + 1) % 4, 0] * q[i, 1] for i in range(4))
    assert cross < 0                                           # counter-clockwise on screen (y down)
    corners = np.array([[128 + 80 * (sx * np.cos(t) - sy * np.sin(t)), 128 + 80 * (sx * np.sin(t) + sy * np.cos(t))]
                        for sx, sy in ((1, -1), (-1, -1), (-1, 1), (1, 1))])
    for c in corners:
        assert np.min(np.hypot(*(q - c).T)) <= 3.0


def test_run_based_labelling_agrees_with_the_numpy_restatement_on_adversarial_masks(find_quadrangle):
    """Round 4 replaced the two flood fills by a run-based union-find (csrc/contour.cpp).  Component order, hole detection
    (4-connected background that does not touch the frame), bounding-box pruning and the traced borders must be unchanged: the C++
    result equals the numpy restatement (scipy labelling) on noise of several densities, checkerboards (thousands of one-pixel
    components), nested frames (holes inside components inside holes), widths that are not multiples of 8 / 64 (the bit packing's
    tail) and degenerate rows."""
    rng = np.random.default_rng(0)
    checked = 0
    for t in range(72):
        h, w = [(256, 256), (64, 64), (37, 91), (128, 70), (9, 200), (65, 129)][t % 6]
        kind = t % 8
        if kind == 0:
            m = rng.random((h, w)) < 0.5
        elif kind == 1:
            m = rng.random((h, w)) < 0.92
        elif kind == 2:
            m = np.zeros((h, w), bool); m[2:h - 2, 2:w - 2] = True
            m[h // 3:h // 3 + 4, w // 3:w // 3 + 5] = False; m[h // 2, w // 2] = False
        elif kind == 3:
            m = np.ones((h, w), bool); m[rng.integers(0, h, 20), rng.integers(0, w, 20)] = False
        elif kind == 4:
            yy, xx = np.mgrid[0:h, 0:w]; m = ((yy // 3 + xx // 3) % 2 == 0)
        elif kind == 5:
            m = np.zeros((h, w), bool); m[1:h - 1, 1:w - 1] = True; m[3:h - 3, 3:w - 3] = False; m[5:h - 5, 5:w - 5] = True
        elif kind == 6:
            m = rng.random((h, w)) < 0.08
        else:
            m = np.zeros((h, w), bool); m[h // 4:3 * h // 4, w // 5:4 * w // 5] = True          # one clean quadrilateral
            m ^= rng.random((h, w)) < 0.002                                                       # ... with salt-and-pepper noise
        mask = (m * 255).astype(np.uint8)
        got, want = find_quadrangle(mask), ChessVision._find_quadrangle(mask)
        assert (got is None) == (want is None), (t, got, want)
        if got is not None:
            assert np.array_equal(got, want), (t, got, want)
            checked += 1
    assert checked >= 6
