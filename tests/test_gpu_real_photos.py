"""GPU: eight of the reference's own test photos (data/test/*/raw, committed as tests/golden/photos8.npz by make_photos.py)
through both CNNs and the batched pipeline, against the CPU oracle on the same arrays.

Everything else in the suite feeds uniform noise or synthetic boards; these are real 512x512 camera images -- smooth regions,
saturated highlights, near-black borders -- through the load-time range calibration of the f16-based engines.  The weights are
still synthetic (the reference ships no checkpoint), so the ground-truth FENs of the fixture are only carried, not scored."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from chessvision import ChessVision, classical, synthetic
from oracle import classical_ref as cref
from oracle import pipeline_ref, synth
from oracle.resnet_ref import ResNet18
from oracle.unet_ref import UNet

pytestmark = pytest.mark.gpu
PHOTOS = Path(__file__).resolve().parent / "golden" / "photos8.npz"


@pytest.fixture(scope="module")
def photos():
    z = np.load(PHOTOS)
    assert z["bgr"].shape == (8, 512, 512, 3) and z["bgr"].dtype == np.uint8
    return [np.ascontiguousarray(im) for im in z["bgr"]]


@pytest.mark.parametrize("prec,tol", [("f16x3", 1e-3), ("f32", 1e-3)])
def test_unet_on_real_photos(photos, prec, tol):
    from chessvision.hip_backend import HipEngine

    net = synth.make_unet(seed=1)
    x = torch.from_numpy(np.stack([cref.resize_area_int(im, (256, 256)) for im in photos]).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        ref = net(x)
    eng = HipEngine(precision=prec, unet_chunk=8)
    eng.load_unet(net.state_dict())
    out = eng.unet_forward(x).cpu()
    u8 = torch.from_numpy(np.stack([classical.resize_area(im, (256, 256)) for im in photos])).cuda()
    fused, mask = eng.unet_forward_u8(u8)
    eng.check_numerics()
    eng.close()
    err = float((out - ref).abs().max())
    assert err <= tol, err
    assert torch.equal(fused.cpu(), out)                                        # u8 entry = float entry, bit for bit
    sure = ref[:, 0].abs() > 1e-4
    assert torch.equal((mask.cpu() > 0)[sure], (ref[:, 0] > 0)[sure])


@pytest.mark.parametrize("prec", ["f16x3", "f32", "f16r"])
def test_resnet18_on_squares_of_real_photos(photos, prec):
    from chessvision.hip_backend import HipEngine

    net = synth.make_resnet(seed=2)
    squares = np.concatenate([cref.split_squares(cref.flip_lr(cref.bgr_to_gray(im))) for im in photos])     # (512,64,64,1)
    x = torch.from_numpy(squares.astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        ref = net(x)
    eng = HipEngine(precision=prec, resnet_chunk=512)
    eng.load_resnet18(net.state_dict())
    out = eng.resnet18_forward(x).cpu()
    eng.close()
    p, p_ref = torch.softmax(out, 1), torch.softmax(ref, 1)
    if prec == "f16r":                                     # the classifier's fp16 mode: the bar is on probabilities (configs[2])
        assert float((p - p_ref).abs().max()) <= 1e-3
    else:
        assert float((out - ref).abs().max()) <= 1e-3
    assert bool((p.argmax(1) == p_ref.argmax(1)).all())


def test_process_images_on_real_photos_matches_the_oracle_pipeline(photos, tmp_path):
    pe, pc = synthetic.save_checkpoints(tmp_path, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    usd = {k: torch.from_numpy(v) for k, v in synthetic.unet_state_dict(1, segmenting=True).items()}
    rsd = {k: torch.from_numpy(v) for k, v in synthetic.resnet18_state_dict(2).items()}
    unet, resnet = UNet(3, 1, False), ResNet18()
    unet.load_state_dict(usd, strict=False)
    resnet.load_state_dict(rsd, strict=False)
    got = cv.process_images(photos, fallback_quad=True)
    ref = pipeline_ref.process_images(unet, resnet, photos, fallback_quad=True)
    checked = 0
    for g, r, photo in zip(got, ref, photos):
        ge, re_ = g.board_extraction, r.board_extraction
        assert np.abs(ge.probabilities - re_.probabilities).max() <= 1e-3
        unsure = np.abs(re_.probabilities) < 1e-4
        assert np.array_equal(ge.binary_mask[~unsure], re_.binary_mask[~unsure])
        if not np.array_equal(ge.binary_mask, re_.binary_mask):
            # a flipped pixel inside the tolerance band may move a contour: the oracle continues from the product's mask
            r = pipeline_ref.process_from_mask(resnet, photo, ge.binary_mask, re_.probabilities, False, True)
            re_ = r.board_extraction
        assert (ge.quadrangle is None) == (re_.quadrangle is None) and g.position is not None and r.position is not None
        assert np.array_equal(ge.quadrangle, re_.quadrangle)
        assert np.array_equal(ge.board_image, re_.board_image)         # byte work is bit-exact, fallback quadrangle included
        assert np.abs(g.position.model_probabilities - r.position.model_probabilities).max() <= 1e-3
        top2 = np.sort(r.position.model_probabilities, axis=1)[:, -2:]
        if ((top2[:, 1] - top2[:, 0]) > 2e-3).all():
            assert g.position.fen == r.position.fen and g.position.original_fen == r.position.original_fen
        assert np.array_equal(g.position.squares, cref.split_squares(ge.board_image))
        checked += 1
    assert checked >= 6, checked
