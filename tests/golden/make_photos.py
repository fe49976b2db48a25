#!/usr/bin/env python3
"""Commit eight of the reference's real test photos as DATA.  Run in the build container only (the one place
/root/reference and a JPEG decoder exist):

    python tests/golden/make_photos.py

Source: /root/reference/data/test/*/raw/*.JPG (38 photos, 512x512 RGB) with data/test/*/ground_truth/*.txt (the piece
placement the reference's evaluation script scores against, scripts/eval/evaluate.py:264-330).  Four photos of each of the
two folders are decoded and stored in the channel order cv2.imread hands the reference (BGR), together with their file
names and ground-truth placements:

    photos8.npz   bgr (8,512,512,3) uint8 | names (8,) str | fen (8,) str

These are inputs (and labels), not outputs of anything: the parity tests push real image statistics -- instead of uniform
noise and synthetic boards -- through the load-time range calibration of the f16-based engines, the two CNNs and
ChessVision.process_images, and compare against the CPU oracle on the same arrays (tests/test_gpu_real_photos.py).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
from PIL import Image

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/data/test")


def main():
    picks = []
    for folder in sorted(p for p in REF.iterdir() if (p / "raw").is_dir()):
        raws = sorted((folder / "raw").glob("*.JPG"))
        picks += raws[:: max(1, len(raws) // 4)][:4]
    bgr, names, fens = [], [], []
    for path in picks[:8]:
        rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
        assert rgb.shape == (512, 512, 3), (path, rgb.shape)
        bgr.append(rgb[:, :, ::-1].copy())
        names.append(f"{path.parent.parent.name}/{path.name}")
        fens.append((path.parent.parent / "ground_truth" / (path.stem + ".txt")).read_text().strip())
    np.savez_compressed(HERE / "photos8.npz", bgr=np.stack(bgr), names=np.array(names), fen=np.array(fens))
    print("wrote", HERE / "photos8.npz", (HERE / "photos8.npz").stat().st_size, "bytes;", names)


if __name__ == "__main__":
    main()
