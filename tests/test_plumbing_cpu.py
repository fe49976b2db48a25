"""the whole ``process_image`` pipeline on the CPU, with the oracle's modules plugged into
the reference's model seam (``_board_extractor`` / ``_classifier`` are opaque callables, core.py:53-54).  Asserts the
result contract the reference's own tests assert (tests/test_chessvision.py:45-116); no GPU, no HIP library."""
from __future__ import annotations

import re

import numpy as np
import torch

from chessvision import ChessVision, constants
from chessvision.cv_types import BoardExtractionResult, ChessVisionResult, PositionResult
from oracle import synth


def _oracle_extractor():
    """The oracle's UNet(3,1) with the "segmenting" random-init weights (chessvision/synthetic.py: make_segmenting): the real
    module tree at the model seam, whose logits follow the brightness of the photo closely enough to yield real contours."""
    from chessvision import synthetic
    from oracle.unet_ref import UNet

    net = UNet(3, 1, False)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.unet_state_dict(1, segmenting=True).items()}, strict=False)
    return net.eval()


def _photo_with_board():
    img = np.full((512, 512, 3), 20, np.uint8)
    yy, xx = np.mgrid[0:512, 0:512]
    quad = (xx + 0.15 * yy > 90) & (xx - 0.1 * yy < 430) & (yy + 0.1 * xx > 70) & (yy - 0.05 * xx < 420)   # convex, slightly skewed
    img[quad] = 200
    checker = (((xx // 40) + (yy // 40)) % 2 == 0) & quad
    img[checker] = 235
    return img


def test_process_image_contract_on_cpu():
    cv = ChessVision()
    assert cv.device.type == "cpu"
    cv._board_extractor = _oracle_extractor()
    cv._classifier = synth.make_resnet(seed=2).eval()
    result = cv.process_image(_photo_with_board())
    assert isinstance(result, ChessVisionResult) and isinstance(result.board_extraction, BoardExtractionResult)
    ext = result.board_extraction
    assert ext.binary_mask.shape == (256, 256) and ext.binary_mask.dtype == np.uint8
    assert set(np.unique(ext.binary_mask)) <= {0, 255}
    assert ext.probabilities.shape == (256, 256) and ext.probabilities.dtype == np.float32      # raw logits (core.py:287)
    assert ext.quadrangle is not None and ext.quadrangle.shape == (4, 1, 2) and ext.quadrangle.dtype == np.float32
    assert ext.board_image.shape == (constants.BOARD_SIZE[1], constants.BOARD_SIZE[0]) and ext.board_image.dtype == np.uint8
    pos = result.position
    assert isinstance(pos, PositionResult)
    assert pos.model_probabilities.shape == (64, constants.NUM_CLASSES)
    np.testing.assert_allclose(pos.model_probabilities.sum(axis=1), 1.0, atol=1e-5)
    assert pos.squares.shape == (64, 64, 64, 1) and len(pos.square_names) == 64
    for fen in (pos.fen, pos.original_fen):
        ranks = fen.split("/")
        assert len(ranks) == 8
        for r in ranks:
            assert re.fullmatch(r"[prnbqkPRNBQK1-8]+", r)
            assert sum(int(ch) if ch.isdigit() else 1 for ch in r) == 8
    assert not any(ch in "pP" for ch in pos.fen.split("/")[0] + pos.fen.split("/")[7])    # pawn rule (core.py:453-469)
    assert result.processing_time > 0


def test_no_board_gives_no_position_on_cpu():
    cv = ChessVision()
    cv._board_extractor = _oracle_extractor()
    cv._classifier = synth.make_resnet(seed=2).eval()
    result = cv.process_image(np.full((512, 512, 3), 10, np.uint8))
    assert result.board_extraction.board_image is None and result.board_extraction.quadrangle is None
    assert result.position is None


def test_oracle_pipeline_with_fallback_quadrangle_on_cpu():
    """oracle/pipeline_ref.py (the checker of tests/test_gpu_e2e.py): a frame without a board is classified through the
    whole-image quadrangle when fallback_quad is set, and a synthetic board photo is found for real."""
    from chessvision import synthetic
    from oracle import pipeline_ref

    unet, resnet = _oracle_extractor(), synth.make_resnet(seed=2)
    dark = np.full((512, 512, 3), 10, np.uint8)
    res = pipeline_ref.process_images(unet, resnet, [dark, synthetic.board_photo(3)], fallback_quad=True)
    assert res[0].position is not None and res[0].board_extraction.quadrangle.tolist() == [[[510.0, 0.0]], [[0.0, 0.0]], [[0.0, 510.0]], [[510.0, 510.0]]]
    q = res[1].board_extraction.quadrangle.reshape(4, 2)
    assert res[1].position is not None and 250 < float(np.ptp(q[:, 0])) < 420 and 250 < float(np.ptp(q[:, 1])) < 420
    assert pipeline_ref.process_images(unet, resnet, [dark])[0].position is None
