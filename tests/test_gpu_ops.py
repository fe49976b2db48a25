"""GPU parity, op by op: every HIP kernel class on the hot path against the torch-CPU fp32 arithmetic the
reference executes (SURVEY.md section 2.2 op list), called through the C ABI single-layer entry points.

Tolerances (north_star: 1e-3 against the fp32 CPU path):
  * f32 engine: max-abs <= 1e-4 on O(1) outputs (f32 MFMA = exact f32 products, order-of-summation noise only)
  * f16 engine: inputs/weights are rounded to f16 once (rel 2^-11), accumulation is f32; the bound used is
    4e-3 * max|ref| -- the per-layer rounding floor, stated where asserted.
  * f16x3 engine (split-f16: hi + lo f16 per value, 3 MFMAs per k-block): held to the f32 bound.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import prng

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-4, "f16": 4e-3, "f16x3": 1e-4}
PRECS = ["f32", "f16", "f16x3"]


def _t(seed, name, shape, std=1.0):
    return torch.from_numpy(prng.normal(seed, name, shape, 0.0, std))


def _err(got: torch.Tensor, ref: torch.Tensor):
    got = got.detach().cpu()
    d = (got - ref).abs()
    return float(d.max()), float(ref.abs().max())


def _assert_close(got, ref, prec, what):
    err, scale = _err(got, ref)
    bound = TOL[prec] * max(1.0, scale)
    if not err <= bound:
        d = (got.detach().cpu() - ref).abs()
        bad = (d > bound).nonzero()
        ch = sorted(set(int(b[1]) for b in bad[:2000]))[:16]
        raise AssertionError(f"{what} [{prec}]: max-abs err {err:.3e} > {bound:.3e} (|ref|max {scale:.3f}); "
                             f"{len(bad)} bad of {d.numel()}; first bad idx {bad[:4].tolist()}; bad channels {ch}")


def test_mfma_lane_maps(engines):
    e16, e32 = engines["f16"].selftest_mfma()
    assert e16 == 0.0 and e32 == 0.0, (e16, e32)


CONV_CASES = [
    # name, n, cin, h, w, cout, k, stride, relu, residual
    ("inc0_like_cin3", 2, 3, 32, 32, 64, 3, 1, True, False),
    ("c64_c64", 2, 64, 32, 32, 64, 3, 1, True, False),
    ("c64_c128_tailM", 1, 64, 24, 20, 128, 3, 1, True, False),
    ("c128_c256_small", 3, 128, 8, 8, 256, 3, 1, False, False),
    ("c256_c512_deepK", 2, 256, 16, 16, 512, 3, 1, True, False),
    ("stride2_resnet", 4, 64, 16, 16, 128, 3, 2, True, False),
    ("down1x1_s2", 4, 64, 16, 16, 128, 1, 2, False, False),
    ("residual_relu", 4, 128, 8, 8, 128, 3, 1, True, True),
    ("tiny_2x2", 8, 512, 2, 2, 512, 3, 1, True, True),
    ("cout_16_cin_8", 1, 8, 16, 16, 16, 3, 1, False, False),
    ("many_pixels_64x256cfg", 8, 64, 64, 64, 64, 3, 1, True, False),
    ("many_pixels_128x256cfg", 8, 64, 64, 64, 128, 3, 1, True, False),
    ("tile_256x256cfg", 8, 64, 64, 64, 512, 3, 1, True, False),
    ("tile_256x256_residual_tail", 9, 128, 60, 60, 256, 3, 1, True, True),
    ("resnet_layer1_residual_halo", 128, 64, 16, 16, 64, 3, 1, True, True),
    ("cin128_c64_halo_4blocks", 8, 128, 64, 64, 64, 3, 1, True, False),
    ("layer2_like_8x8_packed_images_tail", 514, 128, 8, 8, 128, 3, 1, True, True),
    ("layer2_like_8x8_packed_c256", 516, 128, 8, 8, 256, 3, 1, False, False),
    # >= 4 tiles per CU and K <= 72 stages: the persistent variants of the 8-wave halo tiles (with a ragged last image group)
    ("persistent_halo128_residual", 64, 64, 64, 64, 128, 3, 1, True, True),
    ("persistent_packed_8x8_tail", 4098, 128, 8, 8, 128, 3, 1, True, True),
    # round 4, split-K launches (few output tiles, long K): the halo kernel dealt by channel blocks (maps of 16 x 16 and up) ...
    ("splitk_halo_unet_down4_like", 1, 512, 16, 16, 256, 3, 1, True, False),
    ("splitk_halo_32x32_residual", 1, 256, 32, 32, 128, 3, 1, True, True),
    ("splitk_halo_uneven_blocks_cin160", 1, 160, 16, 16, 64, 3, 1, False, False),
    # ... and the generic kernel dealt by K stages (small maps, strided, 1x1)
    ("splitk_generic_resnet_layer4_like", 64, 512, 2, 2, 512, 3, 1, True, True),
    ("splitk_generic_stride2", 16, 256, 8, 8, 512, 3, 2, True, False),
    ("splitk_generic_4x4", 64, 256, 4, 4, 256, 3, 1, True, True),
]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_bn_relu(engines, prec, case):
    name, n, cin, h, w, cout, k, stride, relu, use_res = case
    eng = engines[prec]
    x = _t(11, name + "x", (n, cin, h, w))
    wt = _t(12, name + "w", (cout, cin, k, k), std=(2.0 / (cin * k * k)) ** 0.5)
    scale = torch.from_numpy(prng.uniform(13, name + "s", (cout,), 0.5, 1.5))
    shift = _t(14, name + "b", (cout,), std=0.1)
    ref = F.conv2d(x, wt, stride=stride, padding=(k - 1) // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = None
    if use_res:
        res = _t(15, name + "r", tuple(ref.shape))
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    got = eng.op_conv2d(x, wt, stride=stride, scale=scale, shift=shift, residual=res, relu=relu)
    _assert_close(got, ref, prec, f"conv {name}")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("shape", [(2, 64, 8, 8, 32), (1, 128, 16, 16, 64), (2, 1024, 4, 4, 512), (16, 128, 64, 64, 64),
                                   (1, 1024, 16, 16, 512)])       # the last one: UNet up1.up at one board (a split-K launch)
def test_conv_transpose_k2s2(engines, prec, shape):
    n, cin, h, w, cout = shape
    x = _t(21, "ctx", (n, cin, h, w))
    wt = _t(22, "ctw", (cin, cout, 2, 2), std=(1.0 / cin) ** 0.5)
    b = _t(23, "ctb", (cout,), std=0.1)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    got = engines[prec].op_conv_transpose2x2(x, wt, b)
    _assert_close(got, ref, prec, f"convT {shape}")


@pytest.mark.parametrize("prec", PRECS)
def test_maxpool2x2(engines, prec):
    x = _t(31, "mp", (2, 64, 16, 24))
    xq = x.half().float() if prec == "f16" else x
    got = engines[prec].op_maxpool2x2(x)
    if prec == "f16x3":
        assert float((got.cpu() - F.max_pool2d(x, 2)).abs().max()) <= 1e-6
    else:
        assert torch.equal(got.cpu(), F.max_pool2d(xq, 2))       # max is exact on the stored values


@pytest.mark.parametrize("prec", PRECS)
def test_maxpool3x3s2(engines, prec):
    x = _t(32, "mp3", (3, 64, 32, 32))
    xq = x.half().float() if prec == "f16" else x
    got = engines[prec].op_maxpool3x3s2(x)
    if prec == "f16x3":
        assert float((got.cpu() - F.max_pool2d(x, 3, stride=2, padding=1)).abs().max()) <= 1e-6
    else:
        assert torch.equal(got.cpu(), F.max_pool2d(xq, 3, stride=2, padding=1))


@pytest.mark.parametrize("prec", PRECS)
def test_upsample_bilinear_align_corners(engines, prec):
    x = _t(33, "up", (2, 32, 16, 16))
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    got = engines[prec].op_upsample_bilinear2x(x)
    _assert_close(got, ref, prec, "bilinear x2 align_corners")


def test_softmax13(engines):
    l = _t(34, "sm", (1000, 13), std=3.0)
    got = engines["f32"].softmax13(l)
    assert float((got.cpu() - torch.softmax(l, 1)).abs().max()) <= 1e-6
