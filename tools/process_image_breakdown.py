"""Where a `process_image` call spends its time (developer tool): the native single-image path re-enacted stage by stage with a
device synchronisation after each stage (so the stage times do NOT overlap the way they do in the product; the sum is an upper bound)."""
from __future__ import annotations

import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from chessvision import ChessVision, constants, synthetic  # noqa: E402
from chessvision.hip_backend import board_homographies, decode_positions, find_quadrangle  # noqa: E402


def main(iters=100):
    with tempfile.TemporaryDirectory() as d:
        pe, pc = synthetic.save_checkpoints(d, segmenting=True)
        cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
        im = synthetic.board_photo(3)
        for _ in range(5):
            cv.process_image(im)
        eng = cv.board_extractor.engine
        st = cv._staging(tuple(im.shape))
        staged = st["images"][tuple(im.shape)]
        acc = {}

        def lap(name, t0):
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

        for _ in range(iters):
            t0 = time.perf_counter(); np.copyto(staged.numpy(), im); lap("host copy into pinned staging", t0)
            t0 = time.perf_counter(); img_dev = staged.to(cv.device, non_blocking=True)[None]; lap("H2D image", t0)
            t0 = time.perf_counter(); small = eng.resize_area_u8(img_dev, (256, 256)); lap("resize kernel", t0)
            t0 = time.perf_counter(); lg, mk = eng.unet_forward_u8(small, 0.5, True); lap("UNet B=1", t0)
            t0 = time.perf_counter(); st["mask"].copy_(mk[0], non_blocking=True); lap("D2H mask", t0)
            t0 = time.perf_counter(); st["logits"].copy_(lg[0, 0], non_blocking=True); lap("D2H logits", t0)
            t0 = time.perf_counter(); m = st["mask"].numpy().copy(); q = find_quadrangle(m); lap("mask copy + C++ contours", t0)
            t0 = time.perf_counter(); sc = cv._scale_quadrangle(q, im.shape[:2]); st["inv"].numpy()[:] = board_homographies(sc.reshape(1, 4, 2)).reshape(1, 9); lap("homography", t0)
            t0 = time.perf_counter(); sq, bd = eng.extract_squares_u8(img_dev, st["inv"]); lap("warp kernel (+72 B H2D)", t0)
            t0 = time.perf_counter(); st["board"].copy_(bd[0], non_blocking=True); lap("D2H board", t0)
            t0 = time.perf_counter(); eng.check_numerics(); lap("numeric guard read", t0)
            t0 = time.perf_counter(); pr = eng.resnet18_forward_u8(sq); lap("ResNet-18 B=64", t0)
            t0 = time.perf_counter(); st["probs"].copy_(pr, non_blocking=True); lap("D2H probs", t0)
            t0 = time.perf_counter(); p = st["probs"].numpy().copy(); decode_positions(p[None], False); lap("decode (C++)", t0)
            t0 = time.perf_counter(); b = st["board"].numpy().copy(); l = st["logits"].numpy().copy(); cv.extract_squares(b); lap("result copies", t0)
        total = 0.0
        for k, v in acc.items():
            print(f"{k:36s} {v / iters:8.4f} ms")
            total += v / iters
        print(f"{'sum (stages serialised)':36s} {total:8.4f} ms")
        t0 = time.perf_counter()
        for _ in range(iters):
            cv.process_image(im)
        print(f"{'process_image (as shipped)':36s} {(time.perf_counter() - t0) / iters * 1e3:8.4f} ms")


if __name__ == "__main__":
    main()
