"""Developer probe: is the UNet slower inside process_images than alone?  Per-job event times of the pipeline's UNet launches
against the same job sizes run back to back on an otherwise idle GPU.   usage (GPU box): python tools/pipeline_unet_probe.py"""
import sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import numpy as np, torch
from chessvision import ChessVision, synthetic

with tempfile.TemporaryDirectory() as d:
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    images = [synthetic.board_photo(s) for s in range(256)]
    cv.process_images(images[:96], fallback_quad=True, return_crops=False)
    eng = cv._get_engine("unet")
    # patch the engine call to record one event pair per job
    recs = []
    orig = eng.unet_forward_u8
    def wrapped(x, **kw):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = orig(x, **kw); b.record(); recs.append((x.shape[0], a, b)); return out
    eng.unet_forward_u8 = wrapped
    for rep in range(3):
        recs.clear()
        t0 = time.perf_counter()
        cv.process_images(images, fallback_quad=True, return_crops=False)
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"pipeline call {rep}: {256 / dt:.0f} boards/s; UNet per job:", " ".join(f"{n}:{a.elapsed_time(b):.2f}" for n, a, b in recs),
              f"sum {sum(a.elapsed_time(b) for _, a, b in recs):.2f} ms", flush=True)
    eng.unet_forward_u8 = orig
    small = torch.from_numpy(np.stack([np.ascontiguousarray(im[::2, ::2]) for im in images[:64]])).cuda()
    for n in (16, 48, 64):
        x = small[:n]
        for _ in range(3): eng.unet_forward_u8(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): eng.unet_forward_u8(x)
        torch.cuda.synchronize()
        print(f"alone, {n} boards: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms", flush=True)
    # contention test: the same 64-board UNet while a side stream moves pinned memory both ways (what the pipeline's copy streams do)
    side = torch.cuda.Stream()
    hbuf = torch.empty((64, 512, 512, 3), dtype=torch.uint8, pin_memory=True)
    dbuf = torch.empty((64, 512, 512, 3), dtype=torch.uint8, device="cuda")
    hout = torch.empty((64, 512, 512), dtype=torch.uint8, pin_memory=True)
    dout = torch.empty((64, 512, 512), dtype=torch.uint8, device="cuda")
    x = small[:64]
    for mode in ("none", "h2d", "d2h", "both"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            with torch.cuda.stream(side):
                if mode in ("h2d", "both"): dbuf.copy_(hbuf, non_blocking=True)
                if mode in ("d2h", "both"): hout.copy_(dout, non_blocking=True)
            eng.unet_forward_u8(x)
        torch.cuda.synchronize()
        print(f"alone, 64 boards, side-stream copies = {mode}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms", flush=True)
