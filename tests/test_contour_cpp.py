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
