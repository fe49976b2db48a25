"""Developer tool: stage breakdown of ChessVision.process_images (BASELINE configs[3]) for a few pipeline shapes.

usage (GPU box): python tools/e2e_profile.py [boards]"""
from __future__ import annotations

import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

from chessvision import ChessVision, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with tempfile.TemporaryDirectory() as d:
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    images = [synthetic.board_photo(s) for s in range(n)]
    cv.process_images(images[:96], fallback_quad=True, return_crops=False)
    for first, chunk, last in ((16, 64, 0), (16, 64, 16), (16, 64, 8), (0, 64, 0), (16, 64, 0), (16, 64, 16), (32, 64, 16), (16, 128, 16)):
        best = None
        for _ in range(4):
            tm = {}
            t0 = time.perf_counter()
            cv.process_images(images, fallback_quad=True, timings=tm, first_job=first, last_job=last, pipeline_chunk=chunk, return_crops=False)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, tm)
        dt, tm = best
        gpu = sum(v for k, v in tm.items() if k.endswith("_ms"))
        print(f"first={first:3d} chunk={chunk:3d} last={last:3d}: {n / dt:7.1f} boards/s  total {dt * 1e3:6.1f} ms  gpu {gpu:6.1f} ms  " +
              " ".join(f"{k}={v * 1e3:.1f}" for k, v in tm.items() if k.endswith("_s") and k != "total_s") +
              "  " + " ".join(f"{k}={v:.1f}" for k, v in tm.items() if k.endswith("_ms")))
