"""A/B of the chained ResNet-18 layer1 launch (f16r engine; developer tool): parity of both builds of the schedule against the oracle
on the same squares, per-layer profile, timing.  usage: python tests/dev/chain_ab.py [--squares 16384]
Run once with CV_RESNET_CHAIN=0 and once without (the knob is read once per process)."""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import torch  # noqa: E402

from chessvision.hip_backend import HipEngine  # noqa: E402
from oracle import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--squares", type=int, default=16384)
    ap.add_argument("--check", type=int, default=512)
    ap.add_argument("--prec", default="f16r")
    args = ap.parse_args()
    net = synth.make_resnet(seed=2)
    eng = HipEngine(precision=args.prec, resnet_chunk=16384)
    eng.load_resnet18(net.state_dict())
    x = synth.squares_input(seed=4, n=args.check)
    with torch.no_grad():
        ref = net(x)
    out = eng.resnet18_forward(x.cuda()).cpu()
    p_err = float((torch.softmax(ref, 1) - torch.softmax(out, 1)).abs().max())
    print(f"chain={os.environ.get('CV_RESNET_CHAIN', '1')} form={os.environ.get('CV_CHAIN_WG', '2')} prec={args.prec}: logits err {float((out - ref).abs().max()):.3e} (max {float(ref.abs().max()):.2f}) "
          f"softmax err {p_err:.3e} argmax agree {float((ref.argmax(1) == out.argmax(1)).float().mean()):.4f}")
    for name in ("layer1.0", "layer1", "layer2"):
        try:
            a = torch.from_numpy(eng.activation("resnet18", name))
            print(f"   tap {name}: shape {tuple(a.shape)} absmax {float(a.abs().max()):.3f}")
        except Exception as ex:  # noqa: BLE001
            print(f"   tap {name}: {ex}")
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    sq = torch.randint(0, 256, (args.squares, 1, 64, 64), dtype=torch.uint8, device="cuda", generator=g).float().div_(255)
    for _ in range(3):
        eng.resnet18_forward(sq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.resnet18_forward(sq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    macs = eng.model_macs("resnet18")
    print(f"   {args.squares} squares: {dt * 1e3:.3f} ms/forward = {2 * macs * args.squares / dt / 1e12:.1f} TFLOP/s")
    conv_ms, launches, all_ms, entries = eng.profile("resnet18", sq, iters=1)
    for e in entries:
        tf = 2 * e["macs"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else 0
        gbs = e["bytes"] / (e["ms"] * 1e-3) / 1e9 if e["ms"] > 0 else 0
        print(f"   {e['name']:34s} {e['ms']:7.3f} ms {tf:8.1f} TF {gbs:8.0f} GB/s  {e['kernel']}")
    eng.close()


if __name__ == "__main__":
    main()
