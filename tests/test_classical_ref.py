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


def test_perspective_matrix_and_warp_agree_with_the_product():
    """Two readings of getPerspectiveTransform + warpPerspective: the product blends in floating point after snapping the source
    coordinates to 1/32 pixel ... until round 3; both now use OpenCV's integer weights and round-half-up, written independently
    (matrix by library solve / inverse here, by hand-written elimination / adjugate there).  Same matrix to 1e-9; same image except
    where the last bit of the inverse moves a coordinate across a 1/64-pixel boundary (< 1e-4 of the pixels)."""
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (384, 512, 3), dtype=np.uint8)
    smooth = np.clip(np.add.outer(np.arange(384), np.arange(512))[..., None] * np.array([0.2, 0.25, 0.3]) , 0, 255).astype(np.uint8)
    dest = np.array(((0, 0), (512, 0), (512, 512), (0, 512)), np.float32)
    for trial in range(6):
        quad = np.array([[430, 40], [60, 55], [45, 340], [470, 350]], np.float32) + rng.uniform(-25, 25, (4, 2)).astype(np.float32)
        m_prod = classical.get_perspective_transform(quad, dest)
        m_ref = cref.perspective_matrix(quad, dest)
        assert np.abs(m_prod - m_ref).max() <= 1e-9 * max(1.0, np.abs(m_ref).max())
        for src in (img, smooth):
            a = utils.extract_perspective(src, quad.reshape(4, 1, 2), (512, 512))
            b = cref.extract_board(src, quad, (512, 512))
            diff = np.abs(a.astype(int) - b.astype(int))
            assert diff.max() <= 8, diff.max()
            assert (diff > 0).mean() <= 1e-4, float((diff > 0).mean())      # matrix inverses differ in the last bit: a coordinate may land on the other side of a 1/64 boundary
    # a quadrangle reaching outside the image: constant-zero border on both sides
    quad = np.array([[540, -20], [-30, 10], [-10, 400], [530, 390]], np.float32)
    a = utils.extract_perspective(img, quad.reshape(4, 1, 2), (512, 512))
    b = cref.extract_board(img, quad, (512, 512))
    diff = np.abs(a.astype(int) - b.astype(int))
    # the two matrix inverses differ in the last bit, so a coordinate may round to the neighbouring 1/32-pixel step: on noise that is
    # up to 255 / 32 grey levels, on a handful of pixels
    assert diff.max() <= 8 and (diff > 0).mean() <= 1e-4, (diff.max(), float((diff > 0).mean()))
    assert (b[:4, :4] == 0).all() and (a[:4, :4] == 0).all()
