"""(test infrastructure, not collected by pytest) Developer probe for the classifier's fp16 mode (precision "f16r"): parity against the CPU oracle at BASELINE configs[2]'s
batch (4096 squares), on the He-normal and on the stressed weights, per-layer error growth, and the forward time at the bench's
chunk (16384 squares) next to the f16 and f16x3 engines.  Writes JSON lines to stdout.

usage: python tests/dev/probe_f16r.py [--squares 4096] [--time-squares 16384]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import torch  # noqa: E402

from chessvision.hip_backend import HipEngine  # noqa: E402
from oracle import synth  # noqa: E402  (developer tool: the oracle is the checker here)

TAPS = ["maxpool", "layer1.0", "layer1", "layer2.0", "layer2", "layer3", "layer4"]


def oracle_taps(net, x):
    feats = {}
    hooks = []
    for name in TAPS:
        mod = dict(net.named_modules())[name]
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: feats.__setitem__(name, o.detach())))
    with torch.no_grad():
        out = net(x)
    for h in hooks:
        h.remove()
    return out, feats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--squares", type=int, default=4096)
    ap.add_argument("--time-squares", type=int, default=16384)
    ap.add_argument("--precs", default="f16,f16r,f16x3")
    args = ap.parse_args()
    precs = args.precs.split(",")
    sq = synth.squares_input(seed=42, n=args.squares)
    for wname in ("he_normal", "stress"):
        net = synth.make_resnet(seed=2)
        if wname == "stress":
            synth.load(net, synth.stress_resnet_state_dict(2))
        ref, feats = oracle_taps(net, sq[:256])
        with torch.no_grad():
            ref_all = net(sq)
        p_ref = torch.softmax(ref_all, 1)
        for prec in precs:
            eng = HipEngine(precision=prec, resnet_chunk=args.squares)
            eng.load_resnet18(net.state_dict())
            out = eng.resnet18_forward(sq).cpu()
            p = torch.softmax(out, 1)
            row = {"weights": wname, "prec": prec, "squares": args.squares,
                   "logit_err": float((out - ref_all).abs().max()), "logit_max": float(ref_all.abs().max()),
                   "prob_err": float((p - p_ref).abs().max()),
                   "argmax_agreement": float((p.argmax(1) == p_ref.argmax(1)).float().mean())}
            eng.resnet18_forward(sq[:256])
            layers = {}
            for name in TAPS:
                try:
                    got = torch.from_numpy(eng.activation("resnet18", name))
                except Exception:                      # f16r: layer1.0's output lives inside the chained layer1 launch (round 5)
                    continue
                layers[name] = [float((got - feats[name]).abs().max()), float(feats[name].abs().max())]
            row["layers"] = layers
            print(json.dumps(row), flush=True)
            eng.close()
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    big = torch.randint(0, 256, (args.time_squares, 1, 64, 64), dtype=torch.uint8, device="cuda", generator=g).float().div_(255)
    net = synth.make_resnet(seed=2)
    for prec in precs:
        eng = HipEngine(precision=prec, resnet_chunk=args.time_squares)
        eng.load_resnet18(net.state_dict())
        for _ in range(3):
            eng.resnet18_forward(big, check=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.resnet18_forward(big, check=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        conv_ms, launches, all_ms, entries = eng.profile("resnet18", big, iters=1)
        print(json.dumps({"prec": prec, "squares": args.time_squares, "ms_per_forward": round(ms, 3), "conv_ms": round(conv_ms, 3),
                          "all_ms": round(all_ms, 3), "workspace_gb": round(eng.workspace_bytes() / 1e9, 2),
                          "layers_ms": {e["name"]: round(e["ms"], 3) for e in entries}}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
