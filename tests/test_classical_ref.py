"""CPU: the product's host-side classical stages against the oracle's INDEPENDENT restatements (oracle/classical_ref.py).

The end-to-end oracle (oracle/pipeline_ref.py) runs resize, sigmoid/threshold, gray, flip, the 64-way split, FEN and the pawn
rule through oracle/classical_ref.py, which shares no code with chessvision/classical.py, chessvision/fen.py, the ChessVision
statics, csrc/pipeline.hip or csrc/position.cpp.  Here the two host-side readings are compared directly on random data and on
the reference's own known-answer cases, so a disagreement is found without a GPU."""
from __future__ import annotations

import numpy as np

from chessvision import ChessVision, classical, constants, utils
from chessvision.fen import board_fen
from oracle import classical_ref as cref


def test_label_and_square_tables_match_the_reference_constants():
    assert cref.LABELS == constants.LABEL_NAMES                                   # reference constants.py:23
    assert cref.square_names(False) == constants.SQUARE_NAMES_NORMAL              # constants.py:109-118
    assert cref.square_names(True) == constants.SQUARE_NAMES_FLIPPED              # constants.py:120-129
    assert {n for n in cref.square_names(False) if n[1] in "18"} == set(constants.INVALID_PAWN_SQUARES)


def test_gray_flip_split_resize_agree():
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    assert np.array_equal(classical.bgr_to_gray(img), cref.bgr_to_gray(img))
    every = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing="ij"), -1).reshape(1, -1, 3).astype(np.uint8)
    assert np.array_equal(classical.bgr_to_gray(every), cref.bgr_to_gray(every))  # a lattice over the colour cube
    assert np.array_equal(classical.resize_area(img, (256, 256)), cref.resize_area_int(img, (256, 256)))
    assert np.array_equal(classical.resize_area(img[:, :256], (64, 128)), cref.resize_area_int(img[:, :256], (128, 64)))
    # fractional shrink factors (a 4:3 photo, odd sizes, one integer and one fractional axis): OpenCV's float32 table form, restated
    # twice (vectorised numpy in the product, scalar loops in the oracle) -- identical bytes
    for shape in ((384, 512, 3), (300, 400, 3), (257, 300, 3), (480, 641, 3), (512, 384, 3)):
        photo = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(classical.resize_area(photo, (256, 256)), cref.resize_area(photo, (256, 256))), shape
    flat = np.full((300, 400, 3), 77, np.uint8)
    assert (cref.resize_area(flat, (256, 256)) == 77).all()                        # the weights of a cell sum to 1 (to float32 rounding)
    gray = cref.bgr_to_gray(img)
    assert np.array_equal(classical.flip_horizontal(gray), cref.flip_lr(gray))
    assert np.array_equal(ChessVision.extract_squares(gray), cref.split_squares(gray))


def test_extract_squares_known_answer_of_the_reference():
    """tests/test_chessvision.py:119-146 of the reference: square (r, c) of a board filled with r * 8 + c is constant."""
    board = np.zeros((512, 512), np.uint8)
    for r in range(8):
        for c in range(8):
            board[r * 64:(r + 1) * 64, c * 64:(c + 1) * 64] = r * 8 + c
    sq = cref.split_squares(board)
    assert sq.shape == (64, 64, 64, 1)
    assert all(np.all(sq[i] == i) for i in range(64))


def test_threshold_semantics_at_the_edge():
    logits = np.array([[0.0, 1e-7, -1e-7, 20.0, -20.0, np.log(3.0)]], np.float32)   # sigmoid = .5, >.5, <.5, 1, 0, .75
    for thr in (0.5, 0.75, 0.25):
        prob = 1.0 / (1.0 + np.exp(-logits.astype(np.float32)))
        want = utils.create_binary_mask(prob.astype(np.float32), thr)
        assert np.array_equal(cref.binary_mask(logits, thr), want)


def test_fen_and_pawn_rule_agree_on_random_probabilities():
    rng = np.random.default_rng(5)
    for trial in range(200):
        flip = bool(trial & 1)
        names = cref.square_names(flip)
        probs = rng.dirichlet(np.full(13, 0.3), size=64).astype(np.float32)
        if trial % 3 == 0:                                                        # force pawns onto the back ranks, with ties
            for i in rng.choice(64, 6, replace=False):
                probs[i] = 0.0
                probs[i, 3 if trial % 2 else 9] = 0.5
                probs[i, rng.integers(0, 13)] += 0.25
                probs[i, rng.integers(0, 13)] += 0.25
        labels = [cref.LABELS[int(i)] for i in probs.argmax(1)]
        assert cref.placement(labels, names) == board_fen(labels, names)
        mine, my_fixes = cref.pawn_rule(labels, probs, names)
        theirs, their_fixes = ChessVision.validate_position(list(labels), probs, names)
        assert mine == theirs
        assert my_fixes == [(f.square_name, f.original_piece, f.corrected_piece) for f in their_fixes]
        got = ChessVision.process_position_probabilities(probs, names, np.zeros((64, 64, 64, 1), np.uint8))
        assert got.fen == cref.placement(mine, names) and got.original_fen == cref.placement(labels, names)


def test_start_position_reads_as_the_standard_fen():
    back = "rnbqkbnr"
    labels = list(back) + ["p"] * 8 + ["f"] * 32 + ["P"] * 8 + list(back.upper())
    assert cref.placement(labels, cref.square_names(False)) == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"
    assert cref.placement(labels[::-1], cref.square_names(True)) == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"


def test_perspective_matrix_and_warp_agree_with_the_product_bit_for_bit():
    """getPerspectiveTransform + warpPerspective: the product's host path (numpy rows), its native path (csrc/homography.cpp) and
    the oracle's scalar-Python restatement follow OpenCV's ORDER OF OPERATIONS (LUImpl pivoting / elimination, float32 -x*u
    products, 3x3 inverse as cofactors x 1/det, block-start + in-block-column association of the warp coordinates), so the
    matrices and the warped images are IDENTICAL -- random quadrangles, quadrangles that leave the frame, and the dyadic
    whole-image fallback quadrangle, whose 1/32-pixel coordinates are exact .5 ties that any last-bit difference would flip
    (VERDICT r03 'weak' 2: until round 3 the product inverted with LAPACK and the tests tolerated 8 grey levels on 25 % of the pixels)."""
    from chessvision.hip_backend import board_homographies

    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (384, 512, 3), dtype=np.uint8)
    smooth = np.clip(np.add.outer(np.arange(384), np.arange(512))[..., None] * np.array([0.2, 0.25, 0.3]), 0, 255).astype(np.uint8)
    dest = np.array(((0, 0), (512, 0), (512, 512), (0, 512)), np.float32)
    quads = [np.array([[430, 40], [60, 55], [45, 340], [470, 350]], np.float32) + rng.uniform(-25, 25, (4, 2)).astype(np.float32)
             for _ in range(6)]
    quads.append(np.array([[540, -20], [-30, 10], [-10, 400], [530, 390]], np.float32))           # leaves the frame: zero border
    quads.append(np.array([[255, 0], [0, 0], [0, 255], [255, 255]], np.float32) * np.float32(384 / 256.0))   # fallback quadrangle, 384 rows
    quads.append(np.array([[255, 0], [0, 0], [0, 255], [255, 255]], np.float32) * np.float32(2.0))           # ... of a 512-row photo
    quads.append(np.array([[300.25, 10.5], [11.125, 20.75], [5.5, 370.375], [480.0625, 300.5]], np.float32))  # fractional corners
    inv_native, fwd_native = board_homographies(np.stack(quads), (512, 512), want_forward=True)
    for k, quad in enumerate(quads):
        m_prod = classical.get_perspective_transform(quad, dest)
        m_ref = cref.perspective_matrix(quad, dest)
        assert np.array_equal(m_prod, m_ref) and np.array_equal(fwd_native[k], m_ref)
        assert np.array_equal(classical.invert3(m_prod), cref._invert3(m_ref)) and np.array_equal(inv_native[k], cref._invert3(m_ref))
        assert np.abs(m_ref @ np.append(quad[0], 1.0) / (m_ref @ np.append(quad[0], 1.0))[2] - [0, 0, 1]).max() < 1e-6   # it IS the map
        for src in (img, smooth):
            a = utils.extract_perspective(src, quad.reshape(4, 1, 2), (512, 512))
            b = cref.extract_board(src, quad, (512, 512))
            assert np.array_equal(a, b), (k, int(np.abs(a.astype(int) - b.astype(int)).max()), float((a != b).mean()))
    out = cref.extract_board(img, quads[6], (512, 512))
    assert (out[:4, :4] == 0).all()                                   # constant-zero border outside the photo
    # degenerate quadrangle: OpenCV leaves the solution untouched (zeros) -> every pixel reads source (0,0)
    flat = np.array([[1, 1], [2, 2], [3, 3], [4, 4]], np.float32)
    assert not classical.get_perspective_transform(flat, dest).any() and not cref.perspective_matrix(flat, dest).any()
    assert not board_homographies(flat[None], (512, 512)).any()


def test_native_homographies_match_the_oracle_on_random_and_degenerate_quadrangles():
    """Property test of `cv_board_homographies` (C++) against the oracle's scalar restatement: random convex and non-convex
    quadrangles, huge and tiny coordinates, repeated points and collinear points (singular system -> zero matrices on both sides),
    other output sizes -- identical doubles, not merely close."""
    from chessvision.hip_backend import board_homographies

    rng = np.random.default_rng(99)
    quads = [rng.uniform(-50, 900, (4, 2)).astype(np.float32) for _ in range(300)]
    quads += [np.array([[1e4, 3], [2, 5], [7, 9e3], [8e3, 8e3]], np.float32), np.array([[0.125, 0.25], [0.5, 0.75], [0.875, 1], [1.5, 0.0625]], np.float32)]
    quads += [np.array([[5, 5], [5, 5], [9, 1], [1, 9]], np.float32), np.array([[0, 0], [1, 1], [2, 2], [3, 3]], np.float32),
              np.array([[0, 0], [10, 0], [20, 0], [5, 7]], np.float32)]
    for size in ((512, 512), (256, 384)):
        dest = np.array(((0, 0), (size[0], 0), (size[0], size[1]), (0, size[1])), np.float32)
        inv, fwd = board_homographies(np.stack(quads), size, want_forward=True)
        for k, q in enumerate(quads):
            m = cref.perspective_matrix(q, dest)
            assert np.array_equal(fwd[k], m), (k, q)
            assert np.array_equal(inv[k], cref._invert3(m)), (k, q)


def test_fractional_inter_area_on_random_sizes_including_near_integer_factors():
    """The table form has data-dependent structure: a leading / trailing partial cell only when it covers more than 1e-3 of a pixel,
    a clamped last cell, one to three (shrink < 2) or many (shrink > 3) entries per destination pixel.  Random source sizes --
    including sizes one pixel off an integer factor, where the partial cells are tiny -- give the same bytes in the product's
    vectorised form and in the oracle's scalar form, for 1- and 3-channel images and non-square targets."""
    rng = np.random.default_rng(123)
    shapes = [(257, 511, 3), (511, 513, 3), (513, 257, 1), (767, 769, 3), (1023, 640, 3), (256, 300, 3), (300, 256, 3)]
    shapes += [(int(rng.integers(256, 700)), int(rng.integers(256, 700)), 3) for _ in range(5)]
    for shape in shapes:
        photo = rng.integers(0, 256, shape, dtype=np.uint8)
        got = classical.resize_area(photo, (256, 256))
        assert np.array_equal(got, cref.resize_area(photo, (256, 256))), shape
    photo = rng.integers(0, 256, (333, 444, 3), dtype=np.uint8)
    assert np.array_equal(classical.resize_area(photo, (200, 100)), cref.resize_area(photo, (100, 200)))     # (width, height) vs (h, w)


def test_float_reading_of_warp_perspective_host_equals_oracle():
    """CV_WARP=float: the float-coordinate reading of cv2.warpPerspective next to the fixed-point one.  The host
    form (vectorised numpy float32) and the oracle (scalar float32, one pixel at a time) agree byte for byte on interior, out-of-frame
    and whole-image quadrangles; the two READINGS differ from each other (that is the point of making the choice explicit)."""
    from chessvision import classical

    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    dest = np.array(((0, 0), (48, 0), (48, 48), (0, 48)), np.float32)
    quads = [np.array([[100, 15], [22, 10], [15, 82], [108, 88]], np.float32),
             np.array([[125, 3], [5, 1], [-8, 92], [133, 100]], np.float32),              # leaves the frame: BORDER_CONSTANT taps
             np.array([[127, 0], [0, 0], [0, 95], [127, 95]], np.float32)]                # whole image (the fallback quadrangle's shape)
    differ = 0
    for q in quads:
        m = classical.get_perspective_transform(q, dest)
        a = cref.warp_perspective_float(img, m, (48, 48))
        b = classical.warp_perspective(img, m, (48, 48), mode="float")
        assert np.array_equal(a, b)
        differ += int((b != classical.warp_perspective(img, m, (48, 48), mode="fixed")).sum())
    assert differ > 0
    flat = np.full((40, 40, 3), 77, np.uint8)
    m = classical.get_perspective_transform(np.array([[30, 5], [6, 4], [5, 33], [32, 35]], np.float32), dest)
    assert np.unique(classical.warp_perspective(flat, m, (48, 48), mode="float")).tolist() == [77]
