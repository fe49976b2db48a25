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
