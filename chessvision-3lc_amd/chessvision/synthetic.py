"""Deterministic random-init state dicts and inputs (no trained weights ship with the reference:
``weights/`` is git-ignored, reference ``README.md:39``; ``chessvision/constants.py:46-50``).

Product-side and self-contained (does not import the test oracle).  The shape tables below ARE the
checkpoint contract of the two models -- the same key names / shapes that
``scripts/train/train_unet.py:31-40`` and ``scripts/train/train_classifier.py:114-125`` write and that
``cv_load_unet`` / ``cv_load_resnet18`` validate.  ``tests/test_synthetic.py`` checks them against the oracle's
module trees and checks that both generators agree bit for bit.

Distribution: conv / linear weights He-normal (fan-in), BN gamma ~ U(0.5,1.5),
beta ~ N(0,0.1), running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M1, _M2, _GOLD = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB), np.uint64(0x9E3779B97F4A7C15)


def _hash_name(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for byte in name.encode():
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _finalise(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _words(seed: int, name: str, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        key = _finalise(np.uint64(seed) * _GOLD + _hash_name(name))
        return _finalise((np.arange(n, dtype=np.uint64) + np.uint64(1)) * _GOLD ^ key)


def _uniform(seed, name, shape, lo, hi):
    n = int(np.prod(shape))
    u = (_words(seed, name, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def _normal(seed, name, shape, std):
    n = int(np.prod(shape))
    w = _words(seed, name, n)
    u1 = ((w >> np.uint64(40)).astype(np.float64) + 1.0) * (1.0 / (1 << 24))
    u2 = (w & np.uint64(0xFFFFFF)).astype(np.float64) * (1.0 / (1 << 24))
    return (std * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32).reshape(shape)


def random_u8(seed: int, name: str, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return _words(seed, name, (n + 7) // 8).view(np.uint8)[:n].reshape(shape).copy()


# ---- state-dict shape tables -----------------------------------------------------------------------------
def _bn(prefix: str, c: int):
    return [(f"{prefix}.weight", (c,), "gamma"), (f"{prefix}.bias", (c,), "beta"),
            (f"{prefix}.running_mean", (c,), "mean"), (f"{prefix}.running_var", (c,), "var")]


def _double_conv(prefix: str, cin: int, cout: int, cmid: int | None = None):
    cmid = cmid or cout
    return ([(f"{prefix}.0.weight", (cmid, cin, 3, 3), "conv")] + _bn(f"{prefix}.1", cmid) +
            [(f"{prefix}.3.weight", (cout, cmid, 3, 3), "conv")] + _bn(f"{prefix}.4", cout))


def unet_spec(bilinear: bool = False):
    """[(key, shape, kind)] of UNet(3, 1, bilinear) in state-dict order."""
    f = 2 if bilinear else 1
    spec = _double_conv("inc.double_conv", 3, 64)
    enc = [64, 128, 256, 512, 1024 // f]
    for i in range(4):
        spec += _double_conv(f"down{i + 1}.maxpool_conv.1.double_conv", enc[i], enc[i + 1])
    ups = [(1024, 512 // f), (512, 256 // f), (256, 128 // f), (128, 64)]
    for i, (cin, cout) in enumerate(ups):
        if bilinear:
            spec += _double_conv(f"up{i + 1}.conv.double_conv", cin, cout, cin // 2)
        else:
            spec += [(f"up{i + 1}.up.weight", (cin, cin // 2, 2, 2), "convT"), (f"up{i + 1}.up.bias", (cin // 2,), "beta")]
            spec += _double_conv(f"up{i + 1}.conv.double_conv", cin, cout)
    spec += [("outc.conv.weight", (1, 64, 1, 1), "conv"), ("outc.conv.bias", (1,), "beta")]
    return spec


def resnet18_spec(num_classes: int = 13, in_chans: int = 1):
    """[(key, shape, kind)] of timm resnet18(num_classes, in_chans)."""
    spec = [("conv1.weight", (64, in_chans, 7, 7), "conv")] + _bn("bn1", 64)
    cin = 64
    for layer, width in enumerate([64, 128, 256, 512], start=1):
        for block in range(2):
            p = f"layer{layer}.{block}"
            stride = 2 if (block == 0 and layer > 1) else 1
            spec += [(f"{p}.conv1.weight", (width, cin, 3, 3), "conv")] + _bn(f"{p}.bn1", width)
            spec += [(f"{p}.conv2.weight", (width, width, 3, 3), "conv")] + _bn(f"{p}.bn2", width)
            if stride != 1 or cin != width:
                spec += [(f"{p}.downsample.0.weight", (width, cin, 1, 1), "conv")] + _bn(f"{p}.downsample.1", width)
            cin = width
    spec += [("fc.weight", (num_classes, 512), "linear"), ("fc.bias", (num_classes,), "beta")]
    return spec


def _fill(spec, seed: int, residual_gamma: float | None = None) -> "OrderedDict[str, np.ndarray]":
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape, kind in spec:
        if kind == "conv":
            arr = _normal(seed, key, shape, math.sqrt(2.0 / (shape[1] * shape[2] * shape[3])))
        elif kind == "convT":
            arr = _normal(seed, key, shape, math.sqrt(2.0 / shape[0]))
        elif kind == "linear":
            arr = _normal(seed, key, shape, math.sqrt(1.0 / shape[1]))
        elif kind == "gamma":
            arr = _uniform(seed, key, shape, 0.5, 1.5)
            if residual_gamma is not None and ".bn2." in key:
                arr = arr * np.float32(residual_gamma)
        elif kind == "var":
            arr = _uniform(seed, key, shape, 0.5, 1.5)
        else:                                   # beta / mean / bias
            arr = _normal(seed, key, shape, 0.1)
        out[key] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def unet_state_dict(seed: int = 1, bilinear: bool = False, segmenting: bool = False):
    """Random-init UNet state dict.  ``segmenting=True`` rewires channel 0 of the full-resolution path so that the
    network actually segments ``board_photo`` images (see ``make_segmenting``); every other weight stays random."""
    sd = _fill(unet_spec(bilinear), seed)
    return make_segmenting(sd) if segmenting else sd


def make_segmenting(sd: "OrderedDict[str, np.ndarray]", gain: float = 24.0, offset: float = -12.0, noise: float = 0.25):
    """Turn random UNet weights into a brightness segmenter without touching the architecture: channel 0 of
    inc -> (skip) -> up4.conv carries the mean of the three input channels (centre taps, identity BatchNorm), and OutConv
    reads it with weight ``gain`` and bias ``offset``; the other 63 OutConv weights are scaled by ``noise`` so the rest
    of the (random) network still perturbs the logits by a few tenths.  Bright quadrilaterals on a dark background come
    out as masks with real, slightly ragged contours -- what the contour / warp stages of the pipeline need to be
    exercised end to end when no trained checkpoint exists (the reference ships none, README.md:39)."""
    def identity_bn(prefix):
        sd[prefix + ".weight"][0] = 1.0
        sd[prefix + ".bias"][0] = 0.0
        sd[prefix + ".running_mean"][0] = 0.0
        sd[prefix + ".running_var"][0] = 1.0

    w = sd["inc.double_conv.0.weight"]
    w[0] = 0.0
    w[0, :, 1, 1] = 1.0 / 3.0
    identity_bn("inc.double_conv.1")
    for conv, bn in (("inc.double_conv.3", "inc.double_conv.4"), ("up4.conv.double_conv.0", "up4.conv.double_conv.1"),
                     ("up4.conv.double_conv.3", "up4.conv.double_conv.4")):
        w = sd[conv + ".weight"]
        w[0] = 0.0
        w[0, 0, 1, 1] = 1.0                       # input channel 0 = the skip half of cat([skip, up]) for up4.conv
        identity_bn(bn)
    sd["outc.conv.weight"] *= np.float32(noise)
    sd["outc.conv.weight"][0, 0, 0, 0] = gain
    sd["outc.conv.bias"][0] = offset
    return sd


def board_photo(seed: int, size: int = 512) -> np.ndarray:
    """(size,size,3) uint8 BGR "photo": a bright, slightly skewed convex quadrilateral with an 8x8 checker texture and a
    few piece-like blobs on dark noise."""
    rng = np.random.default_rng(seed)
    s = size / 512.0
    img = rng.integers(0, 36, (size, size, 3), dtype=np.uint8)
    j = rng.uniform(-1.0, 1.0, 8)
    corners = np.array([[95 + 30 * j[0], 75 + 25 * j[1]], [425 + 30 * j[2], 70 + 25 * j[3]],
                        [435 + 30 * j[4], 430 + 25 * j[5]], [85 + 30 * j[6], 440 + 25 * j[7]]]) * s   # TL, TR, BR, BL
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    inside = np.ones((size, size), bool)
    for k in range(4):
        (x0, y0), (x1, y1) = corners[k], corners[(k + 1) % 4]
        inside &= (x1 - x0) * (yy - y0) - (y1 - y0) * (xx - x0) >= 0
    # bilinear board coordinates (u, v) in [0,1]^2 from the corner frame, good enough for a texture
    u = (xx - corners[0, 0]) / max(1.0, corners[1, 0] - corners[0, 0])
    v = (yy - corners[0, 1]) / max(1.0, corners[3, 1] - corners[0, 1])
    checker = ((np.floor(u * 8) + np.floor(v * 8)) % 2).astype(np.uint8)
    base = (165 + 70 * checker).astype(np.uint8)
    img[inside] = base[inside][:, None]
    for _ in range(12):                                            # piece-like blobs, darker or lighter than the squares
        cu, cv_ = rng.integers(0, 8, 2)
        cx = corners[0, 0] + (cu + 0.5) / 8 * (corners[1, 0] - corners[0, 0])
        cy = corners[0, 1] + (cv_ + 0.5) / 8 * (corners[3, 1] - corners[0, 1])
        blob = ((xx - cx) ** 2 + (yy - cy) ** 2 <= (13 * s) ** 2) & inside
        img[blob] = rng.integers(150, 256)
    img[inside] = np.clip(img[inside].astype(np.int16) + rng.integers(-6, 7, (int(inside.sum()), 3)), 0, 255).astype(np.uint8)
    return img


def resnet18_state_dict(seed: int = 2):
    # bn2 gamma halved so eight stacked residual adds keep activations O(1)
    return _fill(resnet18_spec(), seed, residual_gamma=0.5)


def save_checkpoints(directory, seed_unet: int = 1, seed_resnet: int = 2, bilinear: bool = False, segmenting: bool = False):
    """Write random-init checkpoints in the reference's formats (train_unet.py:31-40, train_classifier.py:114-125)."""
    import torch
    from pathlib import Path

    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    meta = {"synthetic": True}
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in unet_state_dict(seed_unet, bilinear, segmenting).items()},
                "metadata": dict(meta, seed=seed_unet)}, d / "best_extractor.pth")
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in resnet18_state_dict(seed_resnet).items()},
                "optimizer_state_dict": {}, "metadata": dict(meta, seed=seed_resnet)}, d / "best_classifier.pth")
    return d / "best_extractor.pth", d / "best_classifier.pth"
