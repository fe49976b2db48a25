"""HBM roofline of the byte kernels of SURVEY section 8f (resize_area_u8, extract_squares_u8): event-timed on 256 boards."""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from chessvision import synthetic  # noqa: E402
from chessvision.hip_backend import HipEngine, board_homographies  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main(n=256):
    eng = HipEngine(precision="f16")
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(np.stack([synthetic.board_photo(s) for s in range(16)] * (n // 16))).cuda()
    quads = np.stack([np.array([[430, 40], [60, 55], [45, 440], [470, 450]], np.float32) + rng.uniform(-25, 25, (4, 2)).astype(np.float32) for _ in range(n)])
    inv = torch.from_numpy(board_homographies(quads).reshape(n, 9)).pin_memory()
    out = {}
    ms = timed(lambda: eng.resize_area_u8(imgs, (256, 256)))
    nbytes = n * (512 * 512 * 3 + 256 * 256 * 3)
    out["resize_area_u8"] = {"ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 3)}
    ms = timed(lambda: eng.extract_squares_u8(imgs, inv, want_boards=True))
    nbytes = n * (512 * 512 * 3 + 2 * 512 * 512)
    out["extract_squares_u8(+boards)"] = {"ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 3)}
    ms = timed(lambda: eng.extract_squares_u8(imgs, inv, want_boards=False))
    nbytes = n * (512 * 512 * 3 + 512 * 512)
    out["extract_squares_u8"] = {"ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 3)}
    ms = timed(lambda: imgs.clone())
    out["copy_reference(torch clone of the images)"] = {"ms": round(ms, 4), "GBps": round(2 * imgs.numel() / ms / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
