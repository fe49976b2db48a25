"""End-to-end CPU oracle: the reference's ``ChessVision.process_image`` chain with the oracle's CNNs at the model seam.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The reference pipeline (``chessvision/core.py:152-195``) is
    resize INTER_AREA -> /255, HWC->CHW -> board_extractor(batch)[0] -> sigmoid > threshold -> contours -> quadrangle
    -> perspective warp -> gray -> flip -> 64 squares -> /255 -> classifier(batch) -> softmax -> argmax -> FEN + pawn rule
and keeps the two models behind opaque callables (``core.py:53-54``).  Here it is written out stage by stage:

* the two CNNs are ``oracle.unet_ref.UNet`` / ``oracle.resnet_ref.ResNet18`` on torch CPU fp32;
* resize (integer factors, fractional shrinks, enlarging), sigmoid / threshold, the perspective matrix and warp (OpenCV's arithmetic in OpenCV's order of
  operations: LU solve, cofactor inverse, block-wise 1/32-pixel coordinates, integer weights, round half up -- the product's
  device warp must equal it byte for byte), gray, flip, the 64-way split, soft-max, arg-max, FEN and the pawn rule are the
  INDEPENDENT restatements of ``oracle/classical_ref.py`` -- they share no code with the product's host path
  (``chessvision/classical.py``, ``fen.py``, ``ChessVision`` statics) nor with its device / C++ path;
* the mask -> quadrangle chain (findContours RETR_CCOMP + CHAIN_APPROX_TC89_KCOS, contourArea / boundingRect filter, arcLength,
  approxPolyDP, rotation; ``core.py:357-411``) is ``oracle/c_ref/contours_ref.c`` through ``oracle/contours_c.py``: a literal
  raster-scan restatement of the published algorithms in plain C that shares nothing with the product's run-based C++ or its
  scipy-label numpy form (round 5; until round 4 this stage was borrowed from the product).

``fallback_quad`` mirrors the option of ``process_images``: boards whose mask yields no quadrangle are classified through
the whole-image quadrangle (TR, TL, BL, BR of the 256x256 mask) so that random-init weights still exercise the classifier.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import classical_ref as cref
from . import contours_c


def process_image(unet, resnet, image: np.ndarray, threshold: float = 0.5, flip: bool = False, fallback_quad: bool = False):
    t0 = time.time()
    h, w = image.shape[:2]
    if h >= 256 and w >= 256:
        small = cref.resize_area(image, (256, 256))                            # integer and fractional shrinks, independent of the product
    else:
        small = cref.resize_area_enlarge(image, (256, 256))                    # a photo below 256 px: OpenCV's fixed-point bilinear path
    x = torch.from_numpy(small.astype(np.float32) / np.float32(255.0)).permute(2, 0, 1)[None]   # core.py:215-216
    with torch.no_grad():
        logits = unet(x)[0, 0].numpy().astype(np.float32)
    mask = cref.binary_mask(logits, threshold)
    return process_from_mask(resnet, image, mask, logits, flip, fallback_quad, t0)


def process_from_mask(resnet, image: np.ndarray, mask: np.ndarray, logits: np.ndarray, flip: bool = False,
                      fallback_quad: bool = False, t0: float | None = None):
    """The chain downstream of the binary mask (reference core.py:277-307 + classify_position).  Also called by the end-to-end
    tests with the PRODUCT's mask when a pixel whose logit sits within the logit tolerance of the threshold flipped, so that
    quadrangle, warp, classifier and FEN of that board are still compared instead of skipped."""
    from chessvision.cv_types import BoardExtractionResult, ChessVisionResult     # result records only

    t0 = time.time() if t0 is None else t0
    h = image.shape[0]
    quad = contours_c.find_quadrangle(mask)                                    # independent plain-C restatement
    if quad is None and fallback_quad:
        quad = np.array([[[255, 0]], [[0, 0]], [[0, 255]], [[255, 255]]], dtype=np.int32)
    if quad is None:
        ext = BoardExtractionResult(board_image=None, binary_mask=mask, quadrangle=None, probabilities=logits)
        return ChessVisionResult(board_extraction=ext, position=None, processing_time=time.time() - t0)
    scaled = np.array(quad * (h / 256.0), dtype=np.float32)                     # height only, core.py:416
    import os

    warp_mode = os.environ.get("CV_WARP", "fixed").strip().lower() or "fixed"      # which reading of cv2.warpPerspective
    board = cref.flip_lr(cref.bgr_to_gray(cref.extract_board(image, scaled, (512, 512), mode=warp_mode)))
    position = classify_board(resnet, board, flip)
    ext = BoardExtractionResult(board_image=board, binary_mask=mask, quadrangle=scaled, probabilities=logits)
    return ChessVisionResult(board_extraction=ext, position=position, processing_time=time.time() - t0)


def classify_board(resnet, board: np.ndarray, flip: bool = False):
    """The second half of the chain on a GIVEN rectified board (reference ``classify_position``, core.py:225-249, and
    ``process_position_probabilities``, core.py:309-355): split, /255, classifier, soft-max, arg-max, FEN, pawn rule."""
    from chessvision.cv_types import PositionResult, ValidationFix

    squares = cref.split_squares(board)
    batch = torch.from_numpy(squares.astype(np.float32)).permute(0, 3, 1, 2) / 255.0            # core.py:236-237
    with torch.no_grad():
        probs = torch.softmax(resnet(batch), dim=1).numpy()
    names = cref.square_names(flip)
    labels = [cref.LABELS[int(i)] for i in np.argmax(probs, axis=1)]
    original = cref.placement(labels, names)
    fixed, fixes = cref.pawn_rule(labels, probs, names)
    return PositionResult(fen=cref.placement(fixed, names), original_fen=original, model_probabilities=probs, squares=squares,
                          square_names=names,
                          validation_fixes=[ValidationFix(square_name=s, original_piece=o, corrected_piece=n, rule_name="no_pawns_on_ends")
                                            for s, o, n in fixes])


def process_images(unet, resnet, images, threshold: float = 0.5, flip: bool = False, fallback_quad: bool = False):
    unet, resnet = unet.eval(), resnet.eval()
    with torch.no_grad():
        return [process_image(unet, resnet, im, threshold, flip, fallback_quad) for im in images]
