"""(test infrastructure, not collected by pytest) One pass over the serve paths of one precision, meant to be run with
`CV_GUARD_ALLOC=1` (every engine buffer ends at an unmapped page) or `=2` (starts at one): an out-of-bounds access of any kernel is
a device fault here instead of a silent read of the neighbouring allocation.   usage: python tests/dev/guard_probe.py <precision> [n]"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

from chessvision import ChessVision, synthetic  # noqa: E402


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    with tempfile.TemporaryDirectory() as d:
        pe, pc = synthetic.save_checkpoints(d, segmenting=True)
        cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), precision=precision)
        print(precision, "loaded", flush=True)
        images = [synthetic.board_photo(1000 + s) for s in range(8)]
        for k in range(3):
            r = cv.process_image(images[k])
            print(precision, "process_image", k, r.position.fen if r.position else None, flush=True)
        for m in (1, 3, 17, n):
            res = cv.process_images([images[i % 8] for i in range(m)], fallback_quad=True)
            print(precision, "process_images", m, sum(r.position is not None for r in res), flush=True)
        cv.close()
    print(precision, "done", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
