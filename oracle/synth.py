"""Deterministic synthetic weights / inputs for the oracle, the tests and bench.py.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  No trained weights ship with the reference
(``weights/`` is git-ignored, ``README.md:39``), so every measurement uses random-init
parameters drawn with ``oracle.prng`` following SURVEY.md section 8(d) config 2:
conv/linear weights He-normal (fan-in), BN gamma ~ U(0.5,1.5), beta ~ N(0,0.1),
running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5); inputs are uint8 U{0..255} / 255.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import prng
from .resnet_ref import ResNet18
from .unet_ref import UNet


def synth_state_dict(model: torch.nn.Module, seed: int, residual_gamma: float | None = None) -> dict:
    """Fill every parameter/buffer of ``model`` from the counter PRNG; returns the state dict.

    ``residual_gamma``: if set, the gamma of each ``bn2`` (last BN of a residual block) is scaled
    by it so eight stacked residual adds keep activations O(1) (keeps fp16 well inside range).
    """
    sd = model.state_dict()
    out = {}
    for key, ref in sd.items():
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.tensor(1000, dtype=torch.long)
            continue
        leaf = key.rsplit(".", 1)[-1]
        if ref.dim() == 4:                       # conv / conv-transpose weight
            if "up.weight" in key and ref.dim() == 4 and key.startswith("up"):
                fan_in = shape[0]                # ConvTranspose2d k2 s2: one tap per output pixel
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            arr = prng.normal(seed, key, shape, 0.0, math.sqrt(2.0 / fan_in))
        elif ref.dim() == 2:                     # linear
            arr = prng.normal(seed, key, shape, 0.0, math.sqrt(1.0 / shape[1]))
        elif leaf == "running_var":
            arr = prng.uniform(seed, key, shape, 0.5, 1.5)
        elif leaf == "running_mean":
            arr = prng.normal(seed, key, shape, 0.0, 0.1)
        elif leaf == "weight":                   # BN gamma
            arr = prng.uniform(seed, key, shape, 0.5, 1.5)
            if residual_gamma is not None and ".bn2." in key:
                arr = arr * np.float32(residual_gamma)
        elif leaf == "bias":                     # BN beta / conv / linear bias
            arr = prng.normal(seed, key, shape, 0.0, 0.1)
        else:
            raise KeyError(f"unexpected state-dict entry {key}")
        out[key] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    return out


def make_unet(seed: int = 1, bilinear: bool = False) -> UNet:
    net = UNet(3, 1, bilinear)
    net.load_state_dict(synth_state_dict(net, seed))
    return net.eval()


def make_resnet(seed: int = 2) -> ResNet18:
    net = ResNet18()
    net.load_state_dict(synth_state_dict(net, seed, residual_gamma=0.5))
    return net.eval()


def unet_input(seed: int, batch: int, hw: int = 256) -> torch.Tensor:
    """(B,3,hw,hw) float32 in [0,1]: u8 / 255 exactly as ``core.py:215-216`` produces it."""
    u8 = prng.bytes_u8(seed, "unet_in", (batch, hw, hw, 3))
    return (torch.from_numpy(u8).to(torch.float32) / 255).permute(0, 3, 1, 2).contiguous()


def squares_input(seed: int, n: int, hw: int = 64) -> torch.Tensor:
    """(N,1,hw,hw) float32 in [0,1]: u8 then ``/= 255.0`` as ``core.py:236-237``."""
    u8 = prng.bytes_u8(seed, "squares_in", (n, hw, hw, 1))
    t = torch.from_numpy(u8).to(torch.float32).permute(0, 3, 1, 2).contiguous()
    t /= 255.0
    return t


# ---- range-stress variants (tests of the f16-based engines outside the benign O(1) regime) -------------------------------
# Both rewrites are mathematically the identity on the network function, so the stressed oracle must agree with the plain
# one (checked in tests/test_oracle_structure.py) while its weights, BatchNorm statistics and activations span decades:
#   rescale_conv_bn: conv weight * a, BN running_mean * a, running_var -> a^2 (var + eps) - eps    (BN output unchanged);
#                    a in [1e-2, 1e2] puts running_var anywhere in ~[1e-4, 1e4] and weights in ~[1e-4, 1e1]
#   push_activation: BN gamma, beta * g (post-ReLU activations * g: ReLU is positively homogeneous) and the weights of the
#                    consuming conv / g -- the tensor in between carries magnitudes of 1e3..1e5 (> 65504 = f16 max).
_EPS = 1e-5


def rescale_conv_bn(sd: dict, conv_key: str, bn_key: str, a: float) -> None:
    sd[conv_key + ".weight"] = sd[conv_key + ".weight"] * a
    sd[bn_key + ".running_mean"] = sd[bn_key + ".running_mean"] * a
    sd[bn_key + ".running_var"] = (sd[bn_key + ".running_var"] + _EPS) * (a * a) - _EPS


def push_activation(sd: dict, bn_key: str, consumers: list, g: float) -> None:
    """consumers: [(conv_key, first input channel, number of input channels)] reading the scaled tensor."""
    sd[bn_key + ".weight"] = sd[bn_key + ".weight"] * g
    sd[bn_key + ".bias"] = sd[bn_key + ".bias"] * g
    for conv_key, c0, n in consumers:
        w = sd[conv_key + ".weight"].clone()
        w[:, c0:c0 + n] = w[:, c0:c0 + n] / g
        sd[conv_key + ".weight"] = w


def _log_uniform(seed: int, name: str, lo: float, hi: float) -> float:
    u = float(prng.uniform(seed, name, (1,), 0.0, 1.0)[0])
    return float(lo * (hi / lo) ** u)


def stress_unet_state_dict(seed: int = 1, bilinear: bool = False, push: float = 3.0e4) -> dict:
    net = UNet(3, 1, bilinear)
    sd = synth_state_dict(net, seed)
    pairs = [("inc.double_conv.0", "inc.double_conv.1"), ("inc.double_conv.3", "inc.double_conv.4")]
    for i in range(1, 5):
        p = f"down{i}.maxpool_conv.1.double_conv."
        pairs += [(p + "0", p + "1"), (p + "3", p + "4")]
        p = f"up{i}.conv.double_conv."
        pairs += [(p + "0", p + "1"), (p + "3", p + "4")]
    for conv, bn in pairs:
        rescale_conv_bn(sd, conv, bn, _log_uniform(seed + 77, conv, 1e-2, 1e2))
    # activations of 1e3..1e5 inside three DoubleConvs (first BN -> second conv) at different depths
    push_activation(sd, "inc.double_conv.1", [("inc.double_conv.3", 0, 64)], push / 10)
    push_activation(sd, "down3.maxpool_conv.1.double_conv.1", [("down3.maxpool_conv.1.double_conv.3", 0, 512)], push)
    mid = sd["up2.conv.double_conv.3.weight"].shape[1]
    push_activation(sd, "up2.conv.double_conv.1", [("up2.conv.double_conv.3", 0, mid)], push / 3)
    # ... and one SKIP tensor: down2's output feeds the pool of down3 and the first 256 input channels of up2's first conv, whose
    # other half (the up-sampled tensor) stays O(1) -- the two halves of one concatenated buffer then differ by ~2^15
    push_activation(sd, "down2.maxpool_conv.1.double_conv.4",
                    [("down3.maxpool_conv.1.double_conv.0", 0, 256), ("up2.conv.double_conv.0", 0, 256)], push)
    return sd


def stress_resnet_state_dict(seed: int = 2, push: float = 3.0e4) -> dict:
    net = ResNet18()
    sd = synth_state_dict(net, seed, residual_gamma=0.5)
    for layer in range(1, 5):
        for block in range(2):
            p = f"layer{layer}.{block}"
            rescale_conv_bn(sd, p + ".conv1", p + ".bn1", _log_uniform(seed + 77, p + ".conv1", 1e-2, 1e2))
            rescale_conv_bn(sd, p + ".conv2", p + ".bn2", _log_uniform(seed + 77, p + ".conv2", 1e-2, 1e2))
            if block == 0 and layer > 1:
                rescale_conv_bn(sd, p + ".downsample.0", p + ".downsample.1", _log_uniform(seed + 77, p + ".ds", 1e-2, 1e2))
    rescale_conv_bn(sd, "conv1", "bn1", 20.0)
    push_activation(sd, "layer1.1.bn1", [("layer1.1.conv2", 0, 64)], push / 10)
    push_activation(sd, "layer3.0.bn1", [("layer3.0.conv2", 0, 256)], push)
    return sd


def load(net: torch.nn.Module, sd: dict) -> torch.nn.Module:
    net.load_state_dict(sd)
    return net.eval()
