"""Where the Python layer of `process_image` spends its time (developer tool): the raw ctypes call into cv_process_image with
preallocated outputs, hip_backend.process_image_native, ChessVision.process_image -- medians over warm calls."""
import ctypes, statistics, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import numpy as np
from chessvision import ChessVision, synthetic, hip_backend as hb

with tempfile.TemporaryDirectory() as d:
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    img = synthetic.board_photo(3)
    for _ in range(10): cv.process_image(img)
    eng = cv.board_extractor.engine
    lib = hb.load_library()
    logits = np.empty((256, 256), np.float32); mask = np.empty((256, 256), np.uint8); board = np.empty((512, 512), np.uint8)
    probs = np.empty((64, 13), np.float32); squares = np.empty((64, 64, 64, 1), np.uint8)
    res = hb._ImageResult()
    res.logits, res.mask, res.board = logits.ctypes.data, mask.ctypes.data, board.ctypes.data
    res.probabilities, res.squares = probs.ctypes.data, squares.ctypes.data
    ip = img.ctypes.data; sp = hb._stream_ptr(eng.device)
    def med(f, n=300):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts)
    for rep in range(3):
        a = med(lambda: lib.cv_process_image(eng._h, eng._h, ip, 512, 512, 0.5, 0, 0, ctypes.byref(res), sp))
        b = med(lambda: hb.process_image_native(eng, eng, img, 0.5, False, False))
        c = med(lambda: cv.process_image(img))
        print(f"raw C call {a:.4f} ms | process_image_native {b:.4f} ms | ChessVision.process_image {c:.4f} ms")
