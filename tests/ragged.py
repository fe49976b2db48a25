"""Ragged binary masks for the contour-parity tests: the reference's 631 label masks (tests/golden/masks_all.npz) with the kinds of
damage a real UNet output shows -- a noisy edge band, salt-and-pepper specks, eroded / dilated rims, pin-holes inside the board."""
from __future__ import annotations

from pathlib import Path

import numpy as np
from scipy import ndimage

G = Path(__file__).resolve().parent / "golden"


def label_masks() -> np.ndarray:
    z = np.load(G / "masks_all.npz")
    return (np.unpackbits(z["bits"], axis=-1)[..., :256] * 255).astype(np.uint8)


def ragged(mask: np.ndarray, kind: int, rng: np.random.Generator) -> np.ndarray:
    m = mask > 0
    if kind == 0:                                              # coin-flip band of +-1 px around the edge
        band = ndimage.binary_dilation(m, iterations=1) & ~ndimage.binary_erosion(m, iterations=1)
        m = np.where(band, rng.random(m.shape) < 0.5, m)
    elif kind == 1:                                            # salt and pepper everywhere
        m = m ^ (rng.random(m.shape) < 0.004)
    elif kind == 2:                                            # wider band, biased towards keeping the board, plus a few specks
        band = ndimage.binary_dilation(m, iterations=3) & ~ndimage.binary_erosion(m, iterations=2)
        m = np.where(band, rng.random(m.shape) < 0.75, m) ^ (rng.random(m.shape) < 0.0005)
    else:                                                      # pin-holes and small holes inside, blobs outside
        m = m.copy()
        for _ in range(int(rng.integers(1, 6))):
            y, x = rng.integers(4, 250, 2)
            r = int(rng.integers(1, 5))
            m[y:y + r, x:x + r] = ~m[y, x]
    return (m * 255).astype(np.uint8)


def ragged_set(count: int, seed: int = 2025):
    """``count`` (mask, source index, kind) triples, deterministic."""
    masks = label_masks()
    rng = np.random.default_rng(seed)
    for t in range(count):
        i = (t * 37) % len(masks)
        yield ragged(masks[i], t % 4, rng), i, t % 4
