"""CPU: host-side stages between the CNNs, pinned on the reference's own fixtures:
its one pipeline known-answer test (tests/test_chessvision.py:119-146), its mask dataset with annotated corners,
and the label / FEN conventions of constants.py:23 and scripts/eval/evaluate.py:62-86."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

from chessvision import classical, constants, utils
from chessvision.core import ChessVision
from chessvision.fen import board_fen

G = Path(__file__).resolve().parent / "golden"


def test_extract_squares_known_answer():
    board = np.zeros((512, 512), dtype=np.uint8)
    for rank in range(8):
        for file in range(8):
            board[rank * 64:(rank + 1) * 64, file * 64:(file + 1) * 64] = rank * 8 + file
    squares = ChessVision.extract_squares(board)
    assert squares.shape == (64, 64, 64, 1)
    for i in (0, 7, 8, 15, 16, 23, 56, 63):        # a8, h8, a7, h7, a6, h6, a1, h1
        assert squares[i, 0, 0, 0] == i
    assert all((squares[i] == i).all() for i in range(64))


@pytest.mark.parametrize("name", sorted(json.load(open(G / "masks" / "corners.json"))))
def test_reference_masks_give_their_annotated_corners(name):
    corners = np.array(json.load(open(G / "masks" / "corners.json"))[name]) * 256.0
    mask = np.array(Image.open(G / "masks" / f"{name}.png").convert("L"))
    assert set(np.unique(mask)) <= {0, 255}
    quad = ChessVision._find_quadrangle(mask)
    assert quad is not None and quad.shape == (4, 1, 2)
    pts = quad.reshape(4, 2).astype(np.float64)
    # every detected corner within 2 px of an annotated one (tolerance documented in SURVEY.md section 8f-2)
    d = np.sqrt(((pts[:, None, :] - corners[None, :, :]) ** 2).sum(-1))
    assert d.min(axis=1).max() <= 2.0, d
    assert sorted(d.argmin(axis=1).tolist()) == [0, 1, 2, 3]
    # order after _rotate_quadrangle: top-right, top-left, bottom-left, bottom-right (counter-clockwise on screen)
    cx, cy = pts.mean(axis=0)
    quadrant = [("T" if y < cy else "B") + ("L" if x < cx else "R") for x, y in pts]
    assert quadrant == ["TR", "TL", "BL", "BR"]


def test_quadrangle_filters():
    m = np.zeros((256, 256), np.uint8)
    assert ChessVision._find_quadrangle(m) is None
    m[40:220, 30:210] = 255
    m[5:9, 5:9] = 255                               # speck: rejected by the 35 % area rule once there are 2 contours
    q = ChessVision._find_quadrangle(m)
    assert q is not None
    assert sorted(map(tuple, q.reshape(4, 2).tolist())) == [(30, 40), (30, 219), (209, 40), (209, 219)]
    thin = np.zeros((256, 256), np.uint8)
    thin[10:250, 100:140] = 255
    thin[0:3, 0:3] = 255
    assert ChessVision._find_quadrangle(thin) is None      # fails both area share and bounding-box ratio
    s = ChessVision._scale_quadrangle(q, (512, 1024))
    assert s.dtype == np.float32 and np.allclose(s, q * 2.0)   # height-only factor (reference core.py:416)


def test_binary_mask_rule():
    p = np.array([[0.5, 0.50001, 0.49999, 1.0, 0.0]], dtype=np.float32)
    assert utils.create_binary_mask(p, 0.5).tolist() == [[0, 255, 0, 255, 0]]
    with pytest.raises(AssertionError):
        utils.create_binary_mask(p.astype(np.float64), 0.5)
    assert utils.ratio(3, 4) == 0.75 and utils.ratio(0, 4) == -1


def test_resize_area_box_mean_and_identity():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    out = classical.resize_area(img, (256, 256))
    ref = (img.reshape(256, 2, 256, 2, 3).astype(np.uint32).sum(axis=(1, 3)) + 2) // 4
    assert np.array_equal(out, ref.astype(np.uint8))
    assert np.array_equal(classical.resize_area(out, (256, 256)), out)
    frac = classical.resize_area(img[:300, :400], (256, 256))
    assert frac.shape == (256, 256, 3) and abs(float(frac.mean()) - float(img[:300, :400].mean())) < 1.0


def test_perspective_roundtrip_and_gray():
    src = np.array([[10, 20], [200, 30], [220, 210], [15, 190]], np.float32)
    dst = np.array([[0, 0], [512, 0], [512, 512], [0, 512]], np.float32)
    m = classical.get_perspective_transform(src, dst)
    mapped = (m @ np.c_[src, np.ones(4)].T).T
    assert np.allclose(mapped[:, :2] / mapped[:, 2:], dst, atol=1e-6)
    img = np.zeros((256, 256, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 100, 200            # B, G, R
    board = utils.extract_perspective(img, src, (512, 512))
    assert board.shape == (512, 512, 3) and (board[5:500, 5:500] == [10, 100, 200]).all()
    gray = classical.bgr_to_gray(board)
    assert int(gray[256, 256]) == (10 * 3735 + 100 * 19235 + 200 * 9798 + 16384) >> 15      # OpenCV 4.x: 15-bit coefficients
    assert np.array_equal(classical.flip_horizontal(gray), gray[:, ::-1])


def test_label_and_fen_conventions():
    assert constants.LABEL_NAMES == ["B", "K", "N", "P", "Q", "R", "b", "k", "n", "p", "q", "r", "f"]
    assert constants.SQUARE_NAMES_NORMAL[:3] == ["a8", "b8", "c8"] and constants.SQUARE_NAMES_NORMAL[-1] == "h1"
    assert constants.SQUARE_NAMES_FLIPPED[0] == "h1" and constants.SQUARE_NAMES_FLIPPED[-1] == "a8"
    assert len(constants.DARK_SQUARES) == 32 and "a1" in constants.DARK_SQUARES and "h1" not in constants.DARK_SQUARES
    start = list("rnbqkbnr") + ["p"] * 8 + ["f"] * 32 + ["P"] * 8 + list("RNBQKBNR")
    assert board_fen(start, constants.SQUARE_NAMES_NORMAL) == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"
    assert board_fen(start[::-1], constants.SQUARE_NAMES_FLIPPED) == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"
    assert board_fen(["f"] * 64, constants.SQUARE_NAMES_NORMAL) == "8/8/8/8/8/8/8/8"
    with pytest.raises(ValueError):
        board_fen(["x"] + ["f"] * 63, constants.SQUARE_NAMES_NORMAL)


def test_position_from_probabilities_and_pawn_rule():
    probs = np.full((64, 13), 0.01, dtype=np.float32)
    probs[:, constants.LABEL_INDICES["f"]] = 0.5
    e1, a8 = constants.SQUARE_NAMES_NORMAL.index("e1"), 0
    probs[e1, constants.LABEL_INDICES["P"]] = 0.9         # white pawn on rank 1: illegal
    probs[e1, constants.LABEL_INDICES["K"]] = 0.6         # best non-pawn alternative
    probs[a8, constants.LABEL_INDICES["p"]] = 0.9         # black pawn on rank 8: illegal -> falls back to empty
    d4 = constants.SQUARE_NAMES_NORMAL.index("d4")
    probs[d4, constants.LABEL_INDICES["P"]] = 0.9         # legal pawn stays
    squares = np.zeros((64, 64, 64, 1), np.uint8)
    res = ChessVision.process_position_probabilities(probs, constants.SQUARE_NAMES_NORMAL, squares)
    assert res.original_fen == "p7/8/8/8/3P4/8/8/4P3"
    assert res.fen == "8/8/8/8/3P4/8/8/4K3"
    assert [(f.square_name, f.original_piece, f.corrected_piece, f.rule_name) for f in res.validation_fixes] == [
        ("a8", "p", "f", "no_pawns_on_ends"), ("e1", "P", "K", "no_pawns_on_ends")]
    assert (len(res.validation_fixes) > 0) == (res.original_fen != res.fen)
    assert len(res.confidence_scores) == 64 and abs(res.confidence_scores[e1] - 0.9) < 1e-6


def test_board_extraction_from_logits_without_models():
    logits = np.full((256, 256), -8.0, dtype=np.float32)
    yy, xx = np.mgrid[0:256, 0:256]
    inside = (xx > 40 + 0.1 * yy) & (xx < 215 - 0.05 * yy) & (yy > 35) & (yy < 225)
    logits[inside] = 8.0
    image = np.zeros((512, 512, 3), np.uint8)
    image[..., 1] = 120
    res = ChessVision.process_board_extraction_logits(logits, image, 0.5)
    assert res.binary_mask.dtype == np.uint8 and set(np.unique(res.binary_mask)) == {0, 255}
    assert res.quadrangle is not None and res.quadrangle.dtype == np.float32 and res.quadrangle.shape == (4, 1, 2)
    assert res.board_image is not None and res.board_image.shape == (512, 512) and res.board_image.dtype == np.uint8
    assert res.probabilities is logits                          # "probabilities" carries the raw logits
    none = ChessVision.process_board_extraction_logits(np.full((256, 256), -8.0, np.float32), image, 0.5)
    assert none.board_image is None and none.quadrangle is None and none.binary_mask.max() == 0
