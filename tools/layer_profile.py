"""Per-layer event timing of one forward (developer tool; writes a table to stdout / gpurun_out).

usage: python tools/layer_profile.py [--prec f16|f32] [--unet-batch 32] [--squares 4096] [--chunk 16] [--sq-chunk 4096]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import torch  # noqa: E402

from chessvision.hip_backend import HipEngine  # noqa: E402
from chessvision import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="f16")
    ap.add_argument("--unet-batch", type=int, default=128)
    ap.add_argument("--squares", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--sq-chunk", type=int, default=16384)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    eng = HipEngine(precision=args.prec, unet_chunk=args.chunk, resnet_chunk=args.sq_chunk)
    eng.load_unet(synthetic.unet_state_dict(1))
    eng.load_resnet18(synthetic.resnet18_state_dict(2))
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.randint(0, 256, (max(args.unet_batch, 1), 3, 256, 256), dtype=torch.uint8, device="cuda", generator=g).float().div_(255)[: args.unet_batch]
    sq = torch.randint(0, 256, (max(args.squares, 1), 1, 64, 64), dtype=torch.uint8, device="cuda", generator=g).float().div_(255)[: args.squares]
    for model, inp, units in (("unet", x, args.unet_batch), ("resnet18", sq, args.squares)):
        fwd = eng.unet_forward if model == "unet" else eng.resnet18_forward
        for _ in range(2):
            fwd(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fwd(inp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        macs = eng.model_macs(model)
        print(f"== {model} [{args.prec}] batch {units}: {dt * 1e3:.3f} ms/forward, {units / dt:.1f} units/s, "
              f"{2 * macs * units / dt / 1e12:.2f} TFLOP/s end-to-end")
        conv_ms, launches, all_ms, entries = eng.profile(model, inp, iters=1)
        agg = {}
        for e in entries:
            a = agg.setdefault(e["name"], [0.0, 0.0, 0, e["conv"]])
            a[0] += e["ms"]; a[1] += e["macs"]; a[2] += 1
        print(f"   event-timed: conv {conv_ms:.3f} ms over {launches} launches, all kernels {all_ms:.3f} ms")
        for name, (ms, m, cnt, conv) in agg.items():
            tf = 2 * m / (ms * 1e-3) / 1e12 if ms > 0 and m > 0 else 0.0
            print(f"   {name:42s} {ms:9.3f} ms  x{cnt:<3d} {tf:8.1f} TFLOP/s {'conv' if conv else ''}")


if __name__ == "__main__":
    main()
