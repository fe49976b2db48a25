"""Where the host-in / FEN-out pipeline (ChessVision.process_images) spends its time: cProfile over one batch.

usage: python3 tools/e2e_profile.py [boards] [precision]"""
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import numpy as np  # noqa: E402

from chessvision import ChessVision, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
with tempfile.TemporaryDirectory() as d:
    pe, pc = synthetic.save_checkpoints(d)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), precision=prec)
    rng = np.random.default_rng(0)
    images = [rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(n)]
    cv.process_images(images[:8], fallback_quad=True)
    for rep in range(2):
        t0 = time.perf_counter()
        cv.process_images(images, fallback_quad=True)
        dt = time.perf_counter() - t0
        print(f"{n} boards: {dt * 1e3:.1f} ms, {n / dt:.1f} boards/s")
    pr = cProfile.Profile()
    pr.enable()
    cv.process_images(images, fallback_quad=True)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
