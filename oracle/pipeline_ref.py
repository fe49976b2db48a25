"""End-to-end CPU oracle: the reference's ``ChessVision.process_image`` chain with the oracle's CNNs at the model seam.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The reference pipeline (``chessvision/core.py:152-195``) is
    resize INTER_AREA -> /255, HWC->CHW -> board_extractor(batch)[0] -> sigmoid > threshold -> contours -> quadrangle
    -> perspective warp -> gray -> flip -> 64 squares -> /255 -> classifier(batch) -> softmax -> argmax -> FEN + pawn rule
and keeps the two models behind opaque callables (``core.py:53-54``).  Here those callables are ``oracle.unet_ref.UNet``
and ``oracle.resnet_ref.ResNet18`` on torch CPU fp32, and every classical stage is the host-side numpy restatement that
the per-image API of the package runs (``chessvision/classical.py``; OpenCV is not installed here).  The batched GPU
path under test (``ChessVision.process_images``: device resize, fused u8 UNet entry, C++ contours, fused device warp,
u8 classifier entry, C++ FEN) shares none of that code except the model-independent static helpers.

``fallback_quad`` mirrors the option of ``process_images``: boards whose mask yields no quadrangle are classified through
the whole-image quadrangle (TR, TL, BL, BR of the 256x256 mask) so that random-init weights still exercise the classifier.
"""
from __future__ import annotations

import numpy as np
import torch


def make_oracle_chessvision(unet: torch.nn.Module, resnet: torch.nn.Module):
    """A ``ChessVision`` whose two model objects are the oracle's torch modules, pinned to the CPU."""
    from chessvision import ChessVision

    cv = ChessVision()
    cv.device = torch.device("cpu")
    cv._board_extractor = unet.eval()
    cv._classifier = resnet.eval()
    return cv


def process_image(cv, image: np.ndarray, threshold: float = 0.5, flip: bool = False, fallback_quad: bool = False):
    """``cv.process_image`` (reference order of operations); with ``fallback_quad`` a missing quadrangle is replaced by the
    whole-image one before the warp, exactly as ``process_images`` does on the device path."""
    from chessvision import classical, constants, utils
    from chessvision.cv_types import BoardExtractionResult, ChessVisionResult

    result = cv.process_image(image, threshold, flip)
    if result.position is not None or not fallback_quad:
        return result
    ext = result.board_extraction
    quad = np.array([[[255, 0]], [[0, 0]], [[0, 255]], [[255, 255]]], dtype=np.int32)
    scaled = cv._scale_quadrangle(quad, (image.shape[0], image.shape[1]))
    board = utils.extract_perspective(image, scaled, constants.BOARD_SIZE)
    board = classical.flip_horizontal(classical.bgr_to_gray(board))
    ext = BoardExtractionResult(board_image=board, binary_mask=ext.binary_mask, quadrangle=scaled, probabilities=ext.probabilities)
    return ChessVisionResult(board_extraction=ext, position=cv.classify_position(board, flip),
                             processing_time=result.processing_time)


def process_images(unet, resnet, images, threshold: float = 0.5, flip: bool = False, fallback_quad: bool = False):
    cv = make_oracle_chessvision(unet, resnet)
    with torch.no_grad():
        return [process_image(cv, im, threshold, flip, fallback_quad) for im in images]
