"""CPU: cross-check the torch-CPU ops the oracle is composed of against the independent plain-C restatement
(oracle/c_ref/ops_ref.c, double accumulation).  Two implementations that share no code must agree."""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import prng

CREF = Path(os.environ.get("CV_ORACLE_CREF_DIR") or Path(__file__).resolve().parent.parent / "oracle" / "c_ref")   # (sanitizer builds: test_oracle_sanitizers.py)
_fp = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def cref():
    if (CREF / "Makefile").exists():
        subprocess.run(["make"], cwd=CREF, check=True, stdout=subprocess.DEVNULL)
    return ctypes.CDLL(str(CREF / "libops_ref.so"))


def _p(a):
    return a.ctypes.data_as(_fp)


def _rand(name, shape, std=1.0):
    return np.ascontiguousarray(prng.normal(7, name, shape, 0.0, std))


@pytest.mark.parametrize("cin,cout,k,stride,pad,h,w", [(3, 8, 3, 1, 1, 9, 11), (8, 16, 3, 2, 1, 8, 8), (16, 8, 1, 2, 0, 8, 8),
                                                     (1, 4, 7, 2, 3, 16, 16), (8, 1, 1, 1, 0, 5, 5)])
def test_conv2d(cref, cin, cout, k, stride, pad, h, w):
    x, wt, b = _rand("x", (2, cin, h, w)), _rand("w", (cout, cin, k, k), 0.3), _rand("b", (cout,))
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), stride=stride, padding=pad).numpy()
    out = np.empty_like(ref)
    cref.ref_conv2d(_p(x), _p(wt), _p(b), _p(out), 2, cin, h, w, cout, k, stride, pad)
    assert np.abs(out - ref).max() <= 2e-5


def test_conv_transpose_k2s2(cref):
    x, wt, b = _rand("x", (2, 6, 5, 7)), _rand("w", (6, 4, 2, 2), 0.3), _rand("b", (4,))
    ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), stride=2).numpy()
    out = np.empty_like(ref)
    cref.ref_conv_transpose2x2(_p(x), _p(wt), _p(b), _p(out), 2, 6, 5, 7, 4)
    assert np.abs(out - ref).max() <= 1e-5


def test_batchnorm_eval_relu(cref):
    x = _rand("x", (2, 5, 4, 4))
    g, b, m = _rand("g", (5,)), _rand("b", (5,)), _rand("m", (5,))
    v = np.ascontiguousarray(prng.uniform(7, "v", (5,), 0.5, 1.5))
    ref = F.relu(F.batch_norm(torch.from_numpy(x), torch.from_numpy(m), torch.from_numpy(v), torch.from_numpy(g),
                              torch.from_numpy(b), training=False, eps=1e-5)).numpy()
    tmp, out = np.empty_like(x), np.empty_like(x)
    cref.ref_batchnorm_eval(_p(x), _p(g), _p(b), _p(m), _p(v), ctypes.c_float(1e-5), _p(tmp), 2, 5, 16)
    cref.ref_relu(_p(tmp), _p(out), ctypes.c_size_t(x.size))
    assert np.abs(out - ref).max() <= 1e-5


@pytest.mark.parametrize("k,stride,pad", [(2, 2, 0), (3, 2, 1)])
def test_maxpool(cref, k, stride, pad):
    x = _rand("x", (2, 3, 10, 12))
    ref = F.max_pool2d(torch.from_numpy(x), k, stride=stride, padding=pad).numpy()
    out = np.empty_like(ref)
    cref.ref_maxpool2d(_p(x), _p(out), 2, 3, 10, 12, k, stride, pad)
    assert np.array_equal(out, ref)


def test_upsample_bilinear_align_corners(cref):
    x = _rand("x", (1, 3, 6, 5))
    ref = F.interpolate(torch.from_numpy(x), scale_factor=2, mode="bilinear", align_corners=True).numpy()
    out = np.empty_like(ref)
    cref.ref_upsample_bilinear2x(_p(x), _p(out), 1, 3, 6, 5)
    assert np.abs(out - ref).max() <= 1e-5


def test_head_ops(cref):
    x = _rand("x", (4, 16, 2, 2))
    wt, b = _rand("w", (13, 16), 0.3), _rand("b", (13,))
    pooled = F.adaptive_avg_pool2d(torch.from_numpy(x), 1).flatten(1)
    logits = F.linear(pooled, torch.from_numpy(wt), torch.from_numpy(b))
    probs = torch.softmax(logits, dim=1).numpy()
    p_c, l_c, s_c = np.empty((4, 16), np.float32), np.empty((4, 13), np.float32), np.empty((4, 13), np.float32)
    cref.ref_global_avgpool(_p(x), _p(p_c), 4, 16, 4)
    cref.ref_linear(_p(p_c), _p(wt), _p(b), _p(l_c), 4, 16, 13)
    cref.ref_softmax_rows(_p(l_c), _p(s_c), 4, 13)
    assert np.abs(s_c - probs).max() <= 1e-6
    z = _rand("z", (100,), 4.0)
    sg = np.empty_like(z)
    cref.ref_sigmoid(_p(z), _p(sg), ctypes.c_size_t(100))
    assert np.abs(sg - torch.sigmoid(torch.from_numpy(z)).numpy()).max() <= 1e-6
