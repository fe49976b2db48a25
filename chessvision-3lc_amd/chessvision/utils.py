"""Model plumbing and small helpers behind ``ChessVision`` (counterpart of the reference's ``chessvision/utils.py``).

What differs from the reference: ``get_classifier_model`` / ``get_board_extractor_model`` do not build torch
modules (timm / Pytorch-UNet) -- they return the HIP model objects of ``hip_backend`` -- and checkpoint loading
yields a plain state dict that is packed by ``cv_load_unet`` / ``cv_load_resnet18`` instead of
``module.load_state_dict``.  The four checkpoint layouts accepted are those of ``utils.py:57-80``.
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Any, Mapping

import numpy as np
import torch
from numpy.typing import NDArray

from . import classical, constants

logger = logging.getLogger(__name__)


def get_device() -> torch.device:
    """cuda (ROCm) when visible -- the only device the HIP backend runs on -- else cpu (reference utils.py:20-29)."""
    if torch.cuda.is_available():
        logger.info("Using CUDA (ROCm) device")
        return torch.device("cuda")
    logger.info("Using CPU device")
    return torch.device("cpu")


def read_checkpoint(checkpoint_path: str | os.PathLike) -> tuple[Mapping[str, Any], dict]:
    """Return (state_dict, metadata) from any of the reference's checkpoint layouts:
    {"model_state_dict", "metadata"?} | {"state_dict", ...} | {"model", ...} | a bare state dict."""
    assert checkpoint_path is not None and Path(checkpoint_path).exists(), f"Checkpoint not found: {checkpoint_path}"
    logger.info(f"Loading checkpoint from {checkpoint_path}")
    # weights_only=True (the torch >= 2.6 default the reference's plain torch.load gets, utils.py:51): a checkpoint path is user
    # input, and unpickling arbitrary objects executes code.  The reference's formats (tensors + a plain metadata dict) load in
    # this mode; a checkpoint that really needs full unpickling must be opted in explicitly.
    try:
        blob = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    except Exception as exc:
        if os.environ.get("CHESSVISION_ALLOW_PICKLE") != "1":
            raise RuntimeError(f"{checkpoint_path} does not load with weights_only=True ({type(exc).__name__}: {exc}); "
                               "set CHESSVISION_ALLOW_PICKLE=1 to unpickle it fully if you trust the file") from exc
        blob = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    metadata: dict = {}
    if isinstance(blob, dict):
        for key in ("model_state_dict", "state_dict", "model"):
            if key in blob:
                metadata = blob.get("metadata", {}) or {}
                return blob[key], metadata
    return blob, metadata


def load_model_checkpoint(model, checkpoint_path: str, device: torch.device | None = None):
    """Reference-compatible entry point (utils.py:42-86): ``model`` is a HIP model object; its engine receives the
    packed weights, ``model.metadata`` is set when the checkpoint carries one."""
    state, metadata = read_checkpoint(checkpoint_path)
    if model.model_name == "unet":
        model.engine.load_unet(state)
    else:
        model.engine.load_resnet18(state)
    if metadata:
        model.metadata = metadata
    return model


def get_classifier_model(model_id: str = "resnet18", engine=None):
    """The piece classifier architecture (reference utils.py:32-39: timm ``model_id``, 13 classes, 1 input channel).
    Only ``resnet18`` has a HIP implementation."""
    from .hip_backend import HipBackendError, HipPieceClassifier

    if model_id not in ("resnet18", "", None):
        raise HipBackendError(f"classifier architecture {model_id!r} has no HIP implementation (resnet18 only)")
    return HipPieceClassifier(engine)


def get_board_extractor_model(engine=None):
    from .hip_backend import HipBoardExtractor

    return HipBoardExtractor(engine)


def ratio(a: float, b: float) -> float:
    """min/max, or -1 when either side is 0 (reference utils.py:89-93)."""
    if a == 0 or b == 0:
        return -1
    return min(a, b) / float(max(a, b))


def listdir_nohidden(path: str) -> list[str]:
    """Directory entries that do not start with a dot (public helper of the reference's ``utils.py:96-98``; its eval and
    data scripts import it from here)."""
    return [name for name in os.listdir(path) if name[:1] != "."]


def create_binary_mask(mask: NDArray[np.float32], threshold: float = 0.5) -> NDArray[np.uint8]:
    """probability > threshold -> 255 else 0 (reference utils.py:101-112)."""
    assert isinstance(mask, np.ndarray), "Mask must be a numpy array"
    assert mask.dtype == np.float32, "Mask must be float32"
    assert 0 <= threshold <= 1, "Threshold must be between 0 and 1"
    return np.where(mask > threshold, 255, 0).astype(np.uint8)


def extract_perspective(image: NDArray[np.uint8], approx: NDArray[np.float32], out_size: tuple[int, int]) -> NDArray[np.uint8]:
    """Rectify the quadrangle ``approx`` onto an ``out_size`` image: corners go to (0,0),(w,0),(w,h),(0,h)
    (reference utils.py:115-132)."""
    assert isinstance(image, np.ndarray), "Image must be a numpy array"
    assert image.dtype == np.uint8, "Image must be uint8"
    assert isinstance(approx, np.ndarray), "Approx must be a numpy array"
    assert approx.dtype == np.float32, "Approx must be float32"
    assert len(approx) == 4, "Approx must contain exactly 4 points"
    w, h = out_size
    dest = np.array(((0, 0), (w, 0), (w, h), (0, h)), np.float32)
    coeffs = classical.get_perspective_transform(np.asarray(approx, np.float32).reshape(4, 2), dest)
    return classical.warp_perspective(image, coeffs, out_size)
