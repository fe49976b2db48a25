"""CPU: the mask -> quadrangle chain of the reference (chessvision/core.py:357-411) -- findContours(RETR_CCOMP,
CHAIN_APPROX_TC89_KCOS), contourArea / boundingRect filter, arcLength, approxPolyDP, rotation -- in three implementations that share
no code: the product's C++ (csrc/contour.cpp: run-based components, borders traced from component starts), the product's numpy
host form (chessvision/classical.py: scipy labels) and the independent oracle (oracle/c_ref/contours_ref.c: the published
raster-scan relabelling, literal).  Integer output: the bar is bit-exact."""
from __future__ import annotations

import numpy as np
import pytest

from chessvision import classical
from chessvision.core import ChessVision
from oracle import contours_c as oc

from ragged import label_masks, ragged_set


@pytest.fixture(scope="module")
def hb():
    import __graft_entry__ as ge

    ge.build()
    from chessvision import hip_backend

    return hip_backend


def _same_lists(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def _same_quad(a, b):
    if a is None or b is None:
        return a is None and b is None
    return np.array_equal(np.asarray(a).reshape(4, 2), np.asarray(b).reshape(4, 2))


def test_tc89_known_answers():
    """What cv2.findContours(.., CHAIN_APPROX_TC89_KCOS) is known to return: a filled axis-parallel rectangle comes back as its four
    corners, top-left first, then DOWN the left side (OpenCV traces outer borders counter-clockwise on screen); an isolated pixel
    as itself; the border of a hole runs on the surrounding foreground pixels."""
    m = np.zeros((20, 20), np.uint8)
    m[3:10, 4:12] = 255
    c, holes = oc.find_contours(m, oc.TC89_KCOS)
    assert holes == [False] and c[0].reshape(-1, 2).tolist() == [[4, 3], [4, 9], [11, 9], [11, 3]]
    full, _ = oc.find_contours(m, oc.NONE)
    assert len(full[0]) == 2 * (7 + 8) - 4 and full[0].reshape(-1, 2)[:3].tolist() == [[4, 3], [4, 4], [4, 5]]
    one = np.zeros((9, 9), np.uint8)
    one[4, 5] = 1
    c, _ = oc.find_contours(one, oc.TC89_KCOS)
    assert [x.reshape(-1, 2).tolist() for x in c] == [[[5, 4]]]
    m[5:8, 6:9] = 0                                            # a 3 x 3 hole: its border = the 12 foreground pixels 4-adjacent to it
    c, holes = oc.find_contours(m, oc.NONE)
    assert holes == [False, True] and len(c[1]) == 12 and c[1].reshape(-1, 2)[0].tolist() == [5, 5]
    ring = {(x, y) for x, y in c[1].reshape(-1, 2).tolist()}
    assert ring == {(6, 4), (7, 4), (8, 4), (9, 5), (9, 6), (9, 7), (8, 8), (7, 8), (6, 8), (5, 7), (5, 6), (5, 5)}


def test_ccomp_order_is_newest_outer_first_each_followed_by_its_holes(hb):
    m = np.zeros((60, 60), np.uint8)
    m[2:20, 2:30] = 255                                        # component A (found first) with two holes
    m[5:8, 5:8] = 0
    m[5:8, 15:18] = 0
    m[30:55, 10:50] = 255                                      # component B (found second) with one hole that holds an island
    m[35:50, 15:45] = 0
    m[40:44, 25:30] = 255                                      # island C (found last): a top-level contour under RETR_CCOMP
    for finder in (lambda x: oc.find_contours(x, oc.NONE), lambda x: hb.find_contours(x, False),
                   lambda x: classical.find_contours(x, False, True)):
        c, holes = finder(m)
        firsts = [tuple(x.reshape(-1, 2)[0].tolist()) for x in c]
        assert holes == [False, False, True, False, True, True]
        assert firsts == [(25, 40), (10, 30), (14, 35), (2, 2), (14, 5), (4, 5)]     # C, B, B's hole, A, A's second hole, A's first


def test_arc_length_takes_float_segments_and_starts_with_the_closing_one():
    c = np.array([[0, 0], [1, 2], [4, 3], [2, 7]], np.int32)
    segs = [np.float32(np.sqrt(np.float32(dx * dx + dy * dy))) for dx, dy in ((2, 7), (1, 2), (3, 1), (2, 4))]
    want = 0.0
    for s in segs:
        want += float(s)
    assert oc.arc_length(c) == classical.arc_length(c) == want
    assert want != float(np.sqrt([53.0, 5.0, 10.0, 20.0]).sum())              # the double-precision perimeter differs in the last bits


def test_product_equals_the_independent_oracle_on_ragged_masks(hb):
    """>= 1000 masks with real-UNet-like damage: every contour list (CHAIN_APPROX_NONE and TC89_KCOS, OpenCV's order, hole flags) and
    every quadrangle of the C++ product equals the oracle's."""
    found = compared = 0
    for mask, i, kind in ragged_set(1040):
        for method in (oc.NONE, oc.TC89_KCOS):
            a, ha = oc.find_contours(mask, method)
            b, hb_ = hb.find_contours(mask, bool(method))
            assert ha == hb_ and _same_lists(a, b), (i, kind, method, len(a), len(b))
        qa, qb = oc.find_quadrangle(mask), hb.find_quadrangle(mask)
        assert _same_quad(qa, qb), (i, kind, qa, qb)
        compared += 1
        found += qa is not None
    assert compared == 1040 and found >= 700, (compared, found)


def test_numpy_host_form_equals_the_oracle(hb):
    n = 0
    for mask, i, kind in ragged_set(48, seed=7):
        for method in (oc.NONE, oc.TC89_KCOS):
            a, ha = oc.find_contours(mask, method)
            b, hb_ = classical.find_contours(mask, bool(method), True)
            assert ha == hb_ and _same_lists(a, b), (i, kind, method)
        for c in a[:2]:
            assert oc.arc_length(c) == classical.arc_length(c) and oc.contour_area(c) == classical.contour_area(c)
            for eps in (0.7, 3.0, 0.02 * oc.arc_length(c), 0.1 * oc.arc_length(c)):
                assert np.array_equal(oc.approx_poly_dp(c, eps), classical.approx_poly_dp(c, eps)), (i, kind, eps)
        assert _same_quad(oc.find_quadrangle(mask), ChessVision._find_quadrangle(mask)), (i, kind)
        n += 1
    assert n == 48


def test_label_masks_all_three_agree_and_hit_the_annotations(hb):
    masks = label_masks()
    quads = hb.find_quadrangles(masks)
    for i in range(len(masks)):
        assert _same_quad(quads[i], oc.find_quadrangle(masks[i])), i
    for i in range(0, len(masks), 25):
        assert _same_quad(quads[i], ChessVision._find_quadrangle(masks[i])), i


def test_compression_changes_epsilon_but_not_the_label_mask_quadrangles(hb):
    """What CHAIN_APPROX_TC89_KCOS changes downstream (rounds 1-4 kept every border pixel): the perimeter arcLength sees is shorter
    -- so is epsilon = 0.1 * perimeter -- and approxPolyDP can only pick dominant points.  On the reference's clean label masks the
    quadrangle is the same either way; on ragged masks it is not always: the count is reported."""
    masks = label_masks()
    ratio = []
    for i in range(0, len(masks), 7):
        full, _ = oc.find_contours(masks[i], oc.NONE)
        tc, _ = oc.find_contours(masks[i], oc.TC89_KCOS)
        ratio.append(oc.arc_length(full[0]) / oc.arc_length(tc[0]))
        assert len(tc[0]) < len(full[0]) / 4
    assert 1.0 <= min(ratio) and max(ratio) < 1.09, (min(ratio), max(ratio))
    differ = total = 0
    for mask, i, kind in ragged_set(400, seed=11):
        full, _ = oc.find_contours(mask, oc.NONE)
        big = max(full, key=len)
        q_none = oc.approx_poly_dp(big, 0.1 * oc.arc_length(big))
        q_tc = oc.find_quadrangle(mask)
        if q_tc is None or len(q_none) != 4:
            differ += (q_tc is None) != (len(q_none) != 4)
        else:
            differ += not np.array_equal(np.sort(q_none.reshape(4, 2), axis=0), np.sort(q_tc.reshape(4, 2), axis=0))
        total += 1
    print(f"quadrangle differs between CHAIN_APPROX_NONE and TC89_KCOS on {differ} of {total} ragged masks")
    assert total == 400


def test_enlarging_inter_area_host_equals_oracle():
    """INTER_AREA on a photo smaller than the 256-px target (core.py:212): OpenCV's fixed-point bilinear path with the AREA coefficient
    rule; the product's host form and the independent oracle agree byte for byte (the device form: tests/test_gpu_pipeline.py)."""
    from oracle import classical_ref as cref

    rng = np.random.default_rng(3)
    sizes = [(100, 80), (200, 300), (255, 255), (300, 200), (17, 33), (256, 100)] + \
            [tuple(int(v) for v in rng.integers(20, 256, 2)) for _ in range(4)]
    for h, w in sizes:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(classical.resize_area(img, (256, 256)), cref.resize_area_enlarge(img, (256, 256))), (h, w)
    flat = np.full((90, 70, 3), 201, np.uint8)
    assert np.unique(classical.resize_area(flat, (256, 256))).tolist() == [201]
    ramp = np.repeat(np.arange(128, dtype=np.uint8)[None, :, None], 128, axis=0).repeat(3, axis=2)
    up = classical.resize_area(ramp, (256, 256))
    assert np.all(np.diff(up[0, :, 0].astype(int)) >= 0) and up[0, 0, 0] == 0 and up[0, -1, 0] == 127


def test_degenerate_contours_and_capacity_errors(hb):
    """Isolated pixels, one-pixel lines, a two-pixel blob, a full frame, an empty mask: the three implementations agree on every
    contour list; `cv_find_contours` reports a too-small output buffer as an error instead of writing past it."""
    import ctypes

    cases = []
    m = np.zeros((12, 12), np.uint8); m[3, 4] = 1; m[8, 8] = 255; cases.append(m)                 # two isolated pixels
    m = np.zeros((12, 12), np.uint8); m[5, 2:9] = 1; cases.append(m)                               # horizontal line
    m = np.zeros((12, 12), np.uint8); m[2:10, 6] = 1; cases.append(m)                              # vertical line
    m = np.zeros((12, 12), np.uint8); m[np.arange(2, 9), np.arange(3, 10)] = 1; cases.append(m)   # diagonal (8-connected)
    m = np.zeros((12, 12), np.uint8); m[4, 4] = m[4, 5] = 1; cases.append(m)                       # two pixels
    cases.append(np.full((9, 13), 255, np.uint8))                                                  # everything set: the border is the frame
    cases.append(np.zeros((7, 7), np.uint8))                                                       # nothing
    m = np.full((16, 16), 255, np.uint8); m[1:15, 1:15] = 0; m[4:12, 4:12] = 255; m[6:10, 6:10] = 0; cases.append(m)   # frame > hole > island > hole
    for k, m in enumerate(cases):
        for method in (oc.NONE, oc.TC89_KCOS):
            a, ha = oc.find_contours(m, method)
            b, hbf = hb.find_contours(m, bool(method))
            c, hc = classical.find_contours(m, bool(method), True)
            assert ha == hbf == hc and _same_lists(a, b) and _same_lists(a, c), (k, method)
        assert _same_quad(oc.find_quadrangle(m), hb.find_quadrangle(m)) and _same_quad(oc.find_quadrangle(m), ChessVision._find_quadrangle(m)), k
    lib = hb.load_library()
    m = cases[-1]
    xy = np.zeros((4, 2), np.int32); counts = np.zeros(8, np.int32); holes = np.zeros(8, np.int32); n = ctypes.c_int64(0)
    i32p = ctypes.POINTER(ctypes.c_int32)
    rc = lib.cv_find_contours(m.ctypes.data_as(ctypes.c_void_p), 16, 16, 0, xy.ctypes.data_as(i32p), 4, counts.ctypes.data_as(i32p),
                              holes.ctypes.data_as(i32p), 8, ctypes.byref(n))
    assert rc != 0 and b"capacity" in lib.cv_last_error()
    rc = lib.cv_find_contours(m.ctypes.data_as(ctypes.c_void_p), 16, 16, 7, xy.ctypes.data_as(i32p), 4, counts.ctypes.data_as(i32p),
                              holes.ctypes.data_as(i32p), 8, ctypes.byref(n))
    assert rc != 0                                                                                  # unknown approximation method


def test_random_small_masks_all_three_agree(hb):
    """600 random masks of 5 x 5 to 40 x 40 at densities 0.1 .. 0.9 (every local border configuration: thin lines, diagonal touches,
    pixels that belong to an outer border and a hole border at once, nested holes): C++ == oracle on all of them, numpy host form ==
    oracle on every fifth, contour lists in both approximation modes and quadrangles."""
    rng = np.random.default_rng(77)
    n_contours = 0
    for t in range(600):
        h, w = int(rng.integers(5, 41)), int(rng.integers(5, 41))
        mask = ((rng.random((h, w)) < rng.uniform(0.1, 0.9)) * 255).astype(np.uint8)
        for method in (oc.NONE, oc.TC89_KCOS):
            a, ha = oc.find_contours(mask, method)
            b, hb_ = hb.find_contours(mask, bool(method))
            assert ha == hb_ and _same_lists(a, b), (t, h, w, method)
            if t % 5 == 0:
                c, hc = classical.find_contours(mask, bool(method), True)
                assert ha == hc and _same_lists(a, c), (t, h, w, method, "numpy")
        n_contours += len(a)
        assert _same_quad(oc.find_quadrangle(mask), hb.find_quadrangle(mask)), (t, h, w)
    assert n_contours > 5000


def test_committed_golden_vectors_of_the_label_masks(hb):
    """tests/golden/contours_tc89.npz (written by the oracle, make_contours.py): per label mask the quadrangle, the contour count, the
    point counts of the first contour with and without CHAIN_APPROX_TC89_KCOS and a position-weighted checksum of the compressed
    points.  The product's C++ (all 631) and its numpy host form (every 20th) reproduce them; so does the oracle as built today."""
    from ragged import G

    z = np.load(G / "contours_tc89.npz")
    masks = label_masks()
    assert z["quads"].shape == (631, 4, 2) and z["stats"].shape == (631, 4)

    def stat(contours_tc, contours_full):
        pts = contours_tc[0].reshape(-1, 2).astype(np.int64)
        return (len(contours_tc), len(pts), len(contours_full[0]), int((pts[:, 0] * 1009 + pts[:, 1] * 9176 + np.arange(len(pts)) * 31).sum()))

    for i, m in enumerate(masks):
        assert np.array_equal(hb.find_quadrangle(m).reshape(4, 2), z["quads"][i]), i
        assert stat(hb.find_contours(m, True)[0], hb.find_contours(m, False)[0]) == tuple(z["stats"][i]), i
        if i % 20 == 0:
            assert np.array_equal(ChessVision._find_quadrangle(m).reshape(4, 2), z["quads"][i]), i
            assert stat(classical.find_contours(m, True), classical.find_contours(m, False)) == tuple(z["stats"][i]), i
            assert np.array_equal(oc.find_quadrangle(m).reshape(4, 2), z["quads"][i]), i
