"""CPU, OPTIONAL: every classical stage of the path against the REAL OpenCV, for whoever has `cv2` installed (the build container and the
GPU boxes of this repository do not: the whole module is skipped there, and nothing in the repository may depend on it).  This is
the check ADVICE r04 asked for and the only thing that can turn "matches our restatement of OpenCV" into "matches OpenCV":
run `pip install opencv-python-headless==4.11.0.86 && python -m pytest tests/test_against_cv2_if_installed.py -q` on any machine.

Reference call sites: chessvision/core.py:212 (resize), 360-375 (findContours / arcLength / approxPolyDP), 394-398 (contourArea /
boundingRect), utils.py:131-132 (getPerspectiveTransform / warpPerspective), core.py:299-300 (cvtColor / flip)."""
from __future__ import annotations

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from chessvision import classical  # noqa: E402
from chessvision.core import ChessVision  # noqa: E402

from ragged import ragged_set  # noqa: E402


def test_resize_inter_area_all_three_regimes():
    rng = np.random.default_rng(0)
    for h, w in ((512, 512), (768, 1024), (300, 400), (257, 641), (100, 80), (200, 300), (255, 255)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(classical.resize_area(img, (256, 256)), cv2.resize(img, (256, 256), interpolation=cv2.INTER_AREA)), (h, w)


def test_find_contours_tc89_order_and_descriptors():
    for mask, i, kind in ragged_set(120, seed=5):
        ours, _ = classical.find_contours(mask, True, True)
        theirs, _ = cv2.findContours(mask, cv2.RETR_CCOMP, cv2.CHAIN_APPROX_TC89_KCOS)
        assert len(ours) == len(theirs), (i, kind)
        for a, b in zip(ours, theirs):
            assert np.array_equal(a, b), (i, kind)
        for c in theirs[:3]:
            assert classical.arc_length(c, True) == cv2.arcLength(c, True)
            assert classical.contour_area(c) == cv2.contourArea(c)
            assert classical.bounding_rect(c) == tuple(cv2.boundingRect(c))
            eps = 0.1 * cv2.arcLength(c, True)
            assert np.array_equal(classical.approx_poly_dp(c, eps), cv2.approxPolyDP(c, eps, True))


def test_find_quadrangle_equals_the_reference_chain():
    def reference(mask):                                           # chessvision/core.py:357-411, verbatim calls
        contours, _ = cv2.findContours(mask, cv2.RETR_CCOMP, cv2.CHAIN_APPROX_TC89_KCOS)
        if len(contours) > 1:
            keep = []
            area = float(mask.shape[0] * mask.shape[1])
            for c in contours:
                a = cv2.contourArea(c) / area
                if a < 0.35 or a > 1.0:
                    continue
                _, _, w, h = cv2.boundingRect(c)
                if (min(h, w) / float(max(h, w)) if h and w else -1) < 0.6:
                    continue
                keep.append(c)
            contours = keep
        for c in contours:
            cand = cv2.approxPolyDP(c, 0.1 * cv2.arcLength(c, True), True)
            if len(cand) == 4:
                return cand[[3, 0, 1, 2], :, :] if cand[0, 0, 0] < cand[2, 0, 0] else cand
        return None

    for mask, i, kind in ragged_set(200, seed=6):
        want, got = reference(mask), ChessVision._find_quadrangle(mask)
        assert (want is None) == (got is None) and (want is None or np.array_equal(want, got)), (i, kind)


def test_perspective_matrix_gray_and_one_of_the_two_warp_readings():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (384, 512, 3), dtype=np.uint8)
    dest = np.array(((0, 0), (512, 0), (512, 512), (0, 512)), np.float32)
    quads = [np.array([[400, 60], [90, 40], [60, 330], [430, 350]], np.float32),
             np.array([[500, 10], [20, 5], [-30, 370], [530, 400]], np.float32),
             np.array([[255, 0], [0, 0], [0, 255], [255, 255]], np.float32) * np.float32(384 / 256.0)]      # dyadic ties
    assert np.array_equal(classical.bgr_to_gray(img), cv2.cvtColor(img, cv2.COLOR_BGR2GRAY))
    matches = {"fixed": 0, "float": 0}
    for q in quads:
        m = classical.get_perspective_transform(q, dest)
        assert np.array_equal(m, cv2.getPerspectiveTransform(q, dest))
        ref = cv2.warpPerspective(img, cv2.getPerspectiveTransform(q, dest), (512, 512))
        for mode in matches:
            matches[mode] += int(np.array_equal(classical.warp_perspective(img, m, (512, 512), mode=mode), ref))
    # ONE of the two readings must be this build's warpPerspective, on all three quadrangles (export CV_WARP to it)
    assert 3 in matches.values(), (cv2.__version__, matches)
