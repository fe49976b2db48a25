"""Latency of the reference's per-image entry point (`ChessVision.process_image`, what Flask and scripts/eval call) on a 512x512
photo: median / p10 / p90 wall milliseconds over warm calls, plus a host-side stage breakdown.

usage: python tools/process_image_latency.py [--prec f16x3] [--iters 200]"""
from __future__ import annotations

import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from chessvision import ChessVision, synthetic  # noqa: E402


def measure(prec: str, iters: int) -> dict:
    with tempfile.TemporaryDirectory() as d:
        pe, pc = synthetic.save_checkpoints(d, segmenting=True)
        cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), precision=prec)
        images = [synthetic.board_photo(s) for s in range(8)]
        for im in images:
            cv.process_image(im)                                   # lazy init, workspace, pinned staging
        torch.cuda.synchronize()
        times, found = [], 0
        for k in range(iters):
            im = images[k % len(images)]
            t0 = time.perf_counter()
            r = cv.process_image(im)
            times.append((time.perf_counter() - t0) * 1e3)
            found += int(r.position is not None)
        t_ext, t_cls = [], []
        for k in range(min(iters, 100)):
            im = images[k % len(images)]
            t0 = time.perf_counter()
            e = cv.extract_board(im)
            t1 = time.perf_counter()
            if e.board_image is not None:
                cv.classify_position(e.board_image)
                t_cls.append((time.perf_counter() - t1) * 1e3)
            t_ext.append((t1 - t0) * 1e3)
        a = np.array(times)
        return {"precision": prec, "image": "512x512x3 synthetic board photo", "iters": iters, "boards_found": found,
                "process_image_ms": {"median": round(float(np.median(a)), 3), "p10": round(float(np.percentile(a, 10)), 3),
                                     "p90": round(float(np.percentile(a, 90)), 3), "min": round(float(a.min()), 3)},
                "extract_board_ms_median": round(float(np.median(t_ext)), 3),
                "classify_position_ms_median": round(float(np.median(t_cls)), 3) if t_cls else None}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="f16x3")
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    for prec in args.prec.split(","):
        print(json.dumps(measure(prec, args.iters)))
