"""Counter-based PRNG used for every synthetic tensor (oracle / tests / bench).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  It is deliberately independent of
``torch.manual_seed`` / ``numpy.random`` so the container that generates the golden fixtures
and the GPU box agree bit-for-bit whatever library versions they run.

Element ``i`` of stream ``name`` under ``seed`` is ``mix64(seed, fnv1a(name), i)`` -- a
splitmix64 finaliser -- so any slice of any tensor can be regenerated without state.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def bits64(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """``n`` 64-bit words of stream (seed, name) starting at element ``offset``."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        key = _mix(np.uint64(seed) * _GOLD + _fnv1a(name))
        return _mix((idx + np.uint64(1)) * _GOLD ^ key)


def uniform(seed: int, name: str, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """float32 U[lo, hi) with 24 random mantissa bits."""
    n = int(np.prod(shape))
    u = (bits64(seed, name, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """float32 N(mean, std) by Box-Muller on two decorrelated halves of one 64-bit word."""
    n = int(np.prod(shape))
    w = bits64(seed, name, n)
    u1 = ((w >> np.uint64(40)).astype(np.float64) + 1.0) * (1.0 / (1 << 24))      # (0, 1]
    u2 = ((w & np.uint64(0xFFFFFF)).astype(np.float64)) * (1.0 / (1 << 24))       # [0, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def bytes_u8(seed: int, name: str, shape) -> np.ndarray:
    """uint8 U{0..255}."""
    n = int(np.prod(shape))
    nw = (n + 7) // 8
    return bits64(seed, name, nw).view(np.uint8)[:n].reshape(shape).copy()
