"""CPU: the ChessVision API surface the reference's callers rely on (tests/test_chessvision.py:25-42,
scripts/eval/evaluate.py:357-358, app/computeroot/cv_endpoint.py:131-133) -- no model is built here."""
from __future__ import annotations

import inspect

import pytest
import torch

import chessvision
from chessvision import ChessVision, constants


def test_constructor_signature_and_lazy_state():
    params = list(inspect.signature(ChessVision.__init__).parameters)
    assert params[:6] == ["self", "board_extractor_weights", "board_extractor_model_id", "classifier_weights",
                          "classifier_model_id", "lazy_load"]
    cv = ChessVision()
    assert cv._board_extractor is None and cv._classifier is None          # lazy
    assert cv._board_extractor_weights is None and cv._classifier_weights is None
    assert cv._board_extractor_model_id is None and cv._classifier_model_id is None
    assert isinstance(cv.device, torch.device)
    cv = ChessVision(board_extractor_weights="path/to/extractor.pth", classifier_weights="path/to/classifier.pth")
    assert cv._board_extractor_weights == "path/to/extractor.pth"
    assert cv._classifier_weights == "path/to/classifier.pth"


def test_public_surface_matches_reference():
    for name in ("process_image", "extract_board", "classify_position", "process_board_extraction_logits",
                 "process_position_probabilities", "extract_squares", "validate_position", "_find_quadrangle",
                 "_filter_contours", "_rotate_quadrangle", "_scale_quadrangle", "_initialize_board_extractor",
                 "_initialize_classifier", "predict", "process_images"):
        assert hasattr(ChessVision, name), name
    assert isinstance(inspect.getattr_static(ChessVision, "board_extractor"), property)
    assert isinstance(inspect.getattr_static(ChessVision, "classifier"), property)
    for name in ("process_board_extraction_logits", "process_position_probabilities", "extract_squares",
                 "validate_position", "_find_quadrangle", "_filter_contours", "_rotate_quadrangle", "_scale_quadrangle"):
        assert isinstance(inspect.getattr_static(ChessVision, name), staticmethod), name
    sig = inspect.signature(ChessVision.process_image)
    assert list(sig.parameters)[1:] == ["image", "threshold", "flip"]
    assert sig.parameters["threshold"].default == 0.5 and sig.parameters["flip"].default is False
    assert chessvision.BoardExtractor.__name__ == "HipBoardExtractor"
    assert chessvision.PieceClassifier.__name__ == "HipPieceClassifier"
    assert constants.BEST_EXTRACTOR_WEIGHTS.endswith("weights/best_extractor.pth")


def test_input_validation_happens_before_any_model_work():
    cv = ChessVision()
    with pytest.raises(AssertionError, match="numpy array"):
        cv.process_image([[1, 2, 3]])
    import numpy as np
    with pytest.raises(AssertionError, match="uint8"):
        cv.process_image(np.zeros((8, 8, 3), np.float32))
    with pytest.raises(AssertionError, match="3-dimensional"):
        cv.process_image(np.zeros((8, 8), np.uint8))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_model_access_without_gpu_raises_instead_of_falling_back(tmp_path):
    from chessvision import synthetic
    from chessvision.hip_backend import HipBackendError

    pe, pc = synthetic.save_checkpoints(tmp_path)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), classifier_model_id="resnet18")
    with pytest.raises(HipBackendError):
        _ = cv.board_extractor
    with pytest.raises(ImportError):
        _ = ChessVision(classifier_model_id="yolo").classifier


def test_checkpoint_layouts(tmp_path):
    from chessvision import utils

    sd = {"fc.bias": torch.zeros(13)}
    for i, blob in enumerate([{"model_state_dict": sd, "metadata": {"epoch": 3}}, {"state_dict": sd}, {"model": sd}, sd]):
        p = tmp_path / f"c{i}.pth"
        torch.save(blob, p)
        state, meta = utils.read_checkpoint(p)
        assert set(state) == {"fc.bias"}
        assert meta == ({"epoch": 3} if i == 0 else {})
    with pytest.raises(AssertionError, match="Checkpoint not found"):
        utils.read_checkpoint(tmp_path / "missing.pth")


class _Payload:                                           # an arbitrary (non-tensor) object: needs full unpickling
    def __init__(self):
        self.note = "not a tensor"


def test_checkpoints_are_loaded_without_unpickling_arbitrary_objects(tmp_path, monkeypatch):
    """A checkpoint path is user input (ChessVision kwargs, evaluate.py): the loader uses torch's weights_only mode, which
    the reference's formats satisfy, and refuses anything that needs arbitrary unpickling unless explicitly opted in."""
    from chessvision import utils

    p = tmp_path / "odd.pth"
    torch.save({"model_state_dict": {"fc.bias": torch.zeros(13)}, "metadata": {"obj": _Payload()}}, p)
    monkeypatch.delenv("CHESSVISION_ALLOW_PICKLE", raising=False)
    with pytest.raises(RuntimeError, match="weights_only"):
        utils.read_checkpoint(p)
    monkeypatch.setenv("CHESSVISION_ALLOW_PICKLE", "1")
    state, meta = utils.read_checkpoint(p)
    assert set(state) == {"fc.bias"} and isinstance(meta["obj"], _Payload)
