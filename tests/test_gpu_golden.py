"""GPU: the HIP path against the committed golden vectors (tests/golden/*.npz, made by make_golden.py), through the
C ABI.  f32 engine: 1e-3 (north_star tolerance); the f16 engine is held to its documented rounding floor."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from chessvision import synthetic

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
TOL = {"f32": 1e-3, "f16": 5e-3, "f16x3": 1e-3}


def _fold(bn):
    g, b, m, v = bn
    scale = g / np.sqrt(v + 1e-5)
    return scale, b - m * scale


@pytest.mark.parametrize("prec", ["f32", "f16", "f16x3"])
def test_op_vectors(engines, prec):
    eng, d, tol = engines[prec], np.load(G / "ops.npz"), TOL[prec]
    T = lambda k: torch.from_numpy(d[k])  # noqa: E731

    def close(got, key):
        ref = d[key]
        err = np.abs(got.cpu().numpy() - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), (key, err)

    sc, sh = _fold((d["conv3_g"], d["conv3_b"], d["conv3_m"], d["conv3_v"]))
    close(eng.op_conv2d(T("conv3_x"), d["conv3_w"], scale=sc, shift=sh, relu=True), "conv3_y")
    sc1, sh1 = _fold(d["blk_bn1"])
    y1 = eng.op_conv2d(T("blk_x"), d["blk_w1"], stride=2, scale=sc1, shift=sh1, relu=True)
    close(y1, "blk_y1")
    scd, shd = _fold(d["blk_bnd"])
    yd = eng.op_conv2d(T("blk_x"), d["blk_wd"], stride=2, scale=scd, shift=shd, relu=False)
    close(yd, "blk_yd")
    sc2, sh2 = _fold(d["blk_bn2"])
    close(eng.op_conv2d(y1, d["blk_w2"], scale=sc2, shift=sh2, residual=yd, relu=True), "blk_y")
    close(eng.op_conv_transpose2x2(T("convT_x"), d["convT_w"], d["convT_b"]), "convT_y")
    close(eng.op_upsample_bilinear2x(T("up_x")), "up_y")
    if prec == "f32":
        assert np.array_equal(eng.op_maxpool2x2(T("mp_x")).cpu().numpy(), d["mp2_y"])
        assert np.array_equal(eng.op_maxpool3x3s2(T("mp_x")).cpu().numpy(), d["mp3_y"])
    close(eng.softmax13(T("head_logits")), "head_probs")


@pytest.mark.parametrize("prec", ["f32", "f16", "f16x3"])
@pytest.mark.parametrize("tag,bilinear", [("convT", False), ("bilinear", True)])
def test_unet_vectors(prec, tag, bilinear):
    from chessvision.hip_backend import HipEngine

    d = np.load(G / "unet.npz")
    eng = HipEngine(precision=prec, unet_chunk=2)
    eng.load_unet(synthetic.unet_state_dict(1, bilinear))
    u8 = synthetic.random_u8(3, "unet_in", (2, 256, 256, 3))
    logits, mask = eng.unet_forward_u8(torch.from_numpy(u8))
    flat = logits.cpu().reshape(2, -1)
    err = np.abs(flat[:, torch.from_numpy(d["sample_idx"])].numpy() - d[f"{tag}_samples"]).max()
    scale = max(1.0, np.abs(d[f"{tag}_samples"]).max()) if prec == "f16" else 1.0
    assert err <= TOL[prec] * scale, err
    if prec != "f16":
        assert np.abs(flat.double().sum(1).numpy() - d[f"{tag}_sum"]).max() <= 0.25         # checksum of 65536 logits: a mean bias of 4e-6 per logit
        assert np.abs((mask.cpu() > 0).reshape(2, -1).sum(1).numpy() - d[f"{tag}_mask_count"]).max() <= 4
    eng.close()


@pytest.mark.parametrize("prec", ["f32", "f16", "f16x3"])
def test_resnet_vectors(prec):
    from chessvision.hip_backend import HipEngine

    d = np.load(G / "resnet18.npz")
    eng = HipEngine(precision=prec, resnet_chunk=128)
    eng.load_resnet18(synthetic.resnet18_state_dict(2))
    u8 = synthetic.random_u8(4, "squares_in", (128, 64, 64))
    probs = eng.resnet18_forward_u8(torch.from_numpy(u8)).cpu().numpy()
    assert np.abs(probs - d["probs"]).max() <= (2e-2 if prec == "f16" else 1e-3)
    x = torch.from_numpy(u8).float().unsqueeze(1)
    x /= 255.0
    logits = eng.resnet18_forward(x).cpu().numpy()
    scale = max(1.0, np.abs(d["logits"]).max()) if prec == "f16" else 1.0
    assert np.abs(logits - d["logits"]).max() <= TOL[prec] * scale
    eng.close()


def test_stem_and_head_ops_through_the_model(engines):
    """The 7x7 stem and the avgpool+fc head have no stand-alone C entry point; their golden vectors are checked via
    the ResNet taps: `act1` (stem+BN+ReLU) and the final logits, with layer weights taken from the fixture."""
    from chessvision.hip_backend import HipEngine

    d = np.load(G / "ops.npz")
    sd = synthetic.resnet18_state_dict(2)
    sd["conv1.weight"] = d["stem_w"]
    g, b, m, v = d["stem_bn"]
    sd["bn1.weight"], sd["bn1.bias"], sd["bn1.running_mean"], sd["bn1.running_var"] = g, b, m, v
    eng = HipEngine(precision="f32", resnet_chunk=128)
    eng.load_resnet18(sd)
    eng.resnet18_forward(torch.from_numpy(d["stem_x"]))
    act1 = eng.activation("resnet18", "act1")
    assert np.abs(act1 - d["stem_y"]).max() <= 1e-4
    eng.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3", "f16"])
def test_outconv_golden_vector_and_edge_logits(prec):
    """ops.npz `outc_*` (conv 1x1 64 -> 1 + bias) and the edge logits of the sigmoid/threshold rule (core.py:273, utils.py:101-112)
    through the stand-alone OutConv kernel.  The edge logits travel in channel 0 with weight 1 (exact in the f32 engine); |logit|
    <= 1e-7 sits within one ulp of sigmoid = 0.5, where the device's exp-based sigmoid may round to the other side of the threshold."""
    from chessvision.hip_backend import HipEngine

    d = np.load(G / "ops.npz")
    eng = HipEngine(precision=prec)
    logits, _ = eng.op_outc_1x1(torch.from_numpy(d["outc_x"]), d["outc_w"], d["outc_b"])
    tol = 1e-5 if prec != "f16" else 2e-2
    assert np.abs(logits.cpu().numpy() - d["outc_y"]).max() <= tol
    edge = d["edge_logits"]
    x = torch.zeros(1, 8, 1, len(edge))
    x[0, 0, 0] = torch.from_numpy(edge)
    w = np.zeros(8, np.float32)
    w[0] = 1.0
    lg, mk = eng.op_outc_1x1(x, w, np.zeros(1, np.float32), threshold=0.5)
    eng.close()
    decided = np.abs(edge) > 1e-7                                  # 0.0 itself is decided too: sigmoid(0) == 0.5 exactly -> 0
    decided[0] = True
    assert mk.cpu().numpy().reshape(-1)[decided].tolist() == d["edge_mask"][decided].tolist()
    if prec == "f32":
        assert np.array_equal(lg.cpu().numpy().reshape(-1), edge)
