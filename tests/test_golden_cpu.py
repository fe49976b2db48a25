"""CPU: the oracle reproduces the committed golden vectors (regression pin across torch builds / CPU ISAs)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from oracle import synth

G = Path(__file__).resolve().parent / "golden"


def _bn(x, p):
    g, b, m, v = (torch.from_numpy(a) for a in p)
    return F.batch_norm(x, m, v, g, b, training=False, eps=1e-5)


def test_op_vectors():
    d = np.load(G / "ops.npz")
    T = lambda k: torch.from_numpy(d[k])  # noqa: E731
    y = F.relu(F.batch_norm(F.conv2d(T("conv3_x"), T("conv3_w"), padding=1), T("conv3_m"), T("conv3_v"), T("conv3_g"),
                            T("conv3_b"), training=False, eps=1e-5))
    assert np.abs(y.numpy() - d["conv3_y"]).max() <= 1e-5
    y1 = F.relu(_bn(F.conv2d(T("blk_x"), T("blk_w1"), stride=2, padding=1), d["blk_bn1"]))
    yd = _bn(F.conv2d(T("blk_x"), T("blk_wd"), stride=2), d["blk_bnd"])
    yb = F.relu(_bn(F.conv2d(y1, T("blk_w2"), padding=1), d["blk_bn2"]) + yd)
    assert np.abs(yb.numpy() - d["blk_y"]).max() <= 1e-5
    assert np.abs(F.conv_transpose2d(T("convT_x"), T("convT_w"), T("convT_b"), stride=2).numpy() - d["convT_y"]).max() <= 1e-5
    assert np.abs(F.interpolate(T("up_x"), scale_factor=2, mode="bilinear", align_corners=True).numpy() - d["up_y"]).max() <= 1e-6
    assert np.array_equal(F.max_pool2d(T("mp_x"), 2).numpy(), d["mp2_y"])
    assert np.array_equal(F.max_pool2d(T("mp_x"), 3, stride=2, padding=1).numpy(), d["mp3_y"])
    ys = F.relu(_bn(F.conv2d(T("stem_x"), T("stem_w"), stride=2, padding=3), d["stem_bn"]))
    assert np.abs(ys.numpy() - d["stem_y"]).max() <= 1e-5
    assert np.abs(torch.sigmoid(T("edge_logits")).numpy() - d["edge_sigmoid"]).max() <= 1e-7
    # the reference mask rule is sigmoid(l) > 0.5 in f32: true from ~9e-8 upwards, false at 0
    assert d["edge_mask"].tolist() == [0, 255, 0, 255, 0, 255, 0, 255, 0, 255, 0]


def test_unet_vectors():
    d = np.load(G / "unet.npz")
    idx = torch.from_numpy(d["sample_idx"])
    x = synth.unet_input(seed=3, batch=2)
    for tag, bilinear in (("convT", False), ("bilinear", True)):
        with torch.no_grad():
            y = synth.make_unet(1, bilinear)(x).reshape(2, -1)
        assert np.abs(y[:, idx].numpy() - d[f"{tag}_samples"]).max() <= 2e-4
        assert np.abs(y.double().sum(1).numpy() - d[f"{tag}_sum"]).max() <= 0.5       # f64 checksum over 65536 logits
        assert np.abs((torch.sigmoid(y) > 0.5).sum(1).numpy() - d[f"{tag}_mask_count"]).max() <= 2


def test_resnet_vectors():
    d = np.load(G / "resnet18.npz")
    with torch.no_grad():
        logits = synth.make_resnet(2)(synth.squares_input(seed=4, n=128))
    assert np.abs(logits.numpy() - d["logits"]).max() <= 1e-4
    assert np.abs(torch.softmax(logits, 1).numpy() - d["probs"]).max() <= 1e-5
