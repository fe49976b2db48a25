"""(test infrastructure, not collected by pytest) `ChessVision.process_image` from T request threads of one instance on T request
slots: aggregate requests/s for T = 1, 2, 4, 8 (CHESSVISION_REQUEST_SLOTS decides how many slots exist).  Under
`rocprofv3 --kernel-trace` (argument `trace <T>`) it runs T threads only, so that tests/dev/concurrent_overlap.py can reduce the
kernel trace to busy / overlapped time.

usage: python tests/dev/concurrent_probe.py [sweep | trace <threads>] [calls per thread]"""
from __future__ import annotations

import sys
import tempfile
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

from chessvision import ChessVision, synthetic  # noqa: E402


def run(cv, images, threads, calls):
    def worker(t):
        for k in range(calls):
            cv.process_image(images[(t + k) % len(images)])
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return threads * calls / (time.perf_counter() - t0)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "sweep"
    if "rccl" in sys.argv[2:]:                          # what else bench.py's process holds: a one-rank RCCL communicator ...
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        from chessvision import distributed as cvd
        _, _, device = cvd.init_process_group()
        print("rccl ranks", cvd.count_ranks(device), flush=True)
    extra = []
    if "engines" in sys.argv[2:]:                       # ... and more engines (each with its capture / side streams)
        import torch
        from chessvision.hip_backend import HipEngine
        for prec in ("f32", "f16", "f16r", "f16x3", "f16x3"):
            e = HipEngine(torch.device("cuda", 0), precision=prec)
            e.load_unet(synthetic.unet_state_dict(1, False)); e.load_resnet18(synthetic.resnet18_state_dict(2))
            extra.append(e)
        extra.append([torch.cuda.Stream() for _ in range(24)])
        print("extra engines", len(extra) - 1, flush=True)
    with tempfile.TemporaryDirectory() as d:
        pe, pc = synthetic.save_checkpoints(d, segmenting=True)
        cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
        images = [synthetic.board_photo(700 + k) for k in range(8)]
        for im in images:
            cv.process_image(im)
        if mode == "trace":
            threads = int(sys.argv[2])
            calls = int(sys.argv[3]) if len(sys.argv) > 3 else 100
            cv.warm_request_slots(threads)
            run(cv, images, threads, 20)
            print(f"threads {threads}: {run(cv, images, threads, calls):.0f} requests/s", flush=True)
            return
        calls = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 200
        print(f"slots {cv.warm_request_slots()}")
        if "serialfirst" in sys.argv[2:] or "parallelfirst" in sys.argv[2:]:
            # Does it matter whether the slots' streams are used for the first time one after the other or all at once?
            import numpy as np
            from chessvision.hip_backend import process_image_native
            if "serialfirst" in sys.argv[2:]:
                for slot in cv._slots:
                    for _ in range(2):
                        process_image_native(slot.extractor_engine, slot.classifier_engine, images[0], 0.5, False, True, stream=slot.stream.cuda_stream)
            run(cv, images, 4, 20)
            print(f"threads 4 ({'serial' if 'serialfirst' in sys.argv[2:] else 'parallel'} first use): {run(cv, images, 4, calls):.0f} requests/s", flush=True)
            return
        for variant in ("batch1", "batchother", "torchstreams", "bigws", "bigws1"):      # which part of a batch call is it? (one per process)
            if variant not in sys.argv[2:]:
                continue
            import torch
            run(cv, images, 4, 20)
            print(f"threads 4 before {variant}: {run(cv, images, 4, calls):.0f} requests/s", flush=True)
            if variant == "batch1":
                cv.process_images(images, fallback_quad=True, return_crops=False)
            elif variant == "batchother":
                other = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
                other.process_images([synthetic.board_photo(k) for k in range(128)], fallback_quad=True, return_crops=False)
                keep_other = other
            elif variant == "torchstreams":
                streams = [torch.cuda.Stream() for _ in range(3)]
                host = torch.empty((64, 512, 512, 3), dtype=torch.uint8, pin_memory=True)
                for st in streams:
                    with torch.cuda.stream(st):
                        dev = host.to("cuda", non_blocking=True)
                        back = torch.empty_like(host, pin_memory=True)
                        back.copy_(dev, non_blocking=True)
                torch.cuda.synchronize()
            else:
                nb = 64 if variant == "bigws" else 1
                eng = cv._get_engine("unet")
                small = torch.zeros((nb, 256, 256, 3), dtype=torch.uint8, device="cuda")
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):                       # primary engines on a third stream, as the pipeline's compute stream is
                    eng.unet_forward_u8(small, threshold=0.5, want_mask=True)
                torch.cuda.synchronize()
            run(cv, images, 4, 20)
            print(f"threads 4 after {variant}: {run(cv, images, 4, calls):.0f} requests/s", flush=True)
            return
        if "heat" in sys.argv[2:]:                      # ... and a device that has just run flat out for a while (the bench's other legs)
            print(f"threads 4 before the load: {run(cv, images, 4, calls):.0f} requests/s", flush=True)
            photos = [synthetic.board_photo(k) for k in range(256)]
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 20.0:
                cv.process_images(photos, fallback_quad=True, return_crops=False)
            print("20 s of process_images(256 photos) done", flush=True)
            for wait in (0, 2, 5, 10):
                time.sleep(wait)
                print(f"threads 4, {wait} s later: {run(cv, images, 4, calls):.0f} requests/s", flush=True)
        for threads in (1, 2, 3, 4, 8):
            run(cv, images, threads, 20)
            print(f"threads {threads}: {run(cv, images, threads, calls):.0f} requests/s", flush=True)


if __name__ == "__main__":
    main()
