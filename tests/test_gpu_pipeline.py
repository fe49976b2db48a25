"""GPU: the reference's own integration tests (tests/test_chessvision.py:45-116) restated against the HIP-backed
ChessVision with random-init checkpoints in the reference's file formats, plus the batched API."""
from __future__ import annotations

import numpy as np
import pytest

from chessvision import ChessVision, constants, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv_model(tmp_path_factory):
    d = tmp_path_factory.mktemp("weights")
    pe, pc = synthetic.save_checkpoints(d)
    return ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), lazy_load=True)


def _board_photo(seed=0):
    """512x512 BGR: a bright quadrilateral 'board' with an 8x8 checker texture on dark noise (BASELINE config 1)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 40, (512, 512, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:512, 0:512]
    inside = (xx > 90 + 0.05 * yy) & (xx < 430 - 0.04 * yy) & (yy > 70) & (yy < 440)
    checker = (((xx - 90) // 43 + (yy - 70) // 46) % 2).astype(np.uint8)
    img[inside] = (150 + 80 * checker[inside])[:, None]
    return img


def test_lazy_state_then_models_are_hip_objects(cv_model):
    assert cv_model._board_extractor is None and cv_model._classifier is None
    from chessvision.hip_backend import HipBoardExtractor, HipPieceClassifier

    assert isinstance(cv_model.board_extractor, HipBoardExtractor)
    assert isinstance(cv_model.classifier, HipPieceClassifier)
    assert cv_model._classifier_model_id == "resnet18"                 # fallback branch of core.py:121-130
    assert cv_model.board_extractor.metadata["synthetic"] is True      # checkpoint metadata is carried over


def test_process_image_result_contract(cv_model):
    result = cv_model.process_image(_board_photo())
    assert result.board_extraction is not None
    assert isinstance(result.board_extraction.binary_mask, np.ndarray)
    assert result.board_extraction.binary_mask.dtype == np.uint8
    assert result.board_extraction.probabilities.shape == (256, 256)
    if result.board_extraction.board_image is not None:
        assert result.board_extraction.board_image.shape == (512, 512)
        assert result.position is not None and result.position.fen.count("/") == 7
        assert result.position.model_probabilities.shape == (64, 13)
        assert (len(result.position.validation_fixes) > 0) == (result.position.original_fen != result.position.fen)
    else:
        assert result.position is None
    assert result.processing_time > 0


def test_classify_position_on_a_given_board(cv_model):
    board = np.random.default_rng(1).integers(0, 256, (512, 512), dtype=np.uint8)
    res = cv_model.classify_position(board)
    assert res.squares.shape == (64, 64, 64, 1) and res.model_probabilities.shape == (64, 13)
    assert np.allclose(res.model_probabilities.sum(axis=1), 1.0, atol=1e-5)
    assert res.square_names == constants.SQUARE_NAMES_NORMAL
    flipped = cv_model.classify_position(board, flip=True)
    assert flipped.square_names == constants.SQUARE_NAMES_FLIPPED
    assert np.allclose(flipped.model_probabilities, res.model_probabilities)
    for fix in res.validation_fixes:
        assert fix.square_name in res.square_names and fix.corrected_piece in constants.LABEL_NAMES


def test_batched_api_equals_per_image_api(cv_model):
    images = [_board_photo(s) for s in range(3)]
    single = [cv_model.process_image(im) for im in images]
    batched = cv_model.process_images(images)
    assert len(batched) == 3
    for a, b in zip(single, batched):
        assert np.abs(a.board_extraction.probabilities - b.board_extraction.probabilities).max() <= 1e-4
        assert np.array_equal(a.board_extraction.binary_mask, b.board_extraction.binary_mask)
        assert (a.position is None) == (b.position is None)
        if a.position is not None:
            assert a.position.fen == b.position.fen
            assert np.abs(a.position.model_probabilities - b.position.model_probabilities).max() <= 1e-5
    assert cv_model.process_images([]) == []
