"""(test infrastructure, not collected by pytest) CPU emulation behind section 2 of profiles/r06_tuning.md: where the fp16 classifier's
error comes from, and what the load-time rounding-bias correction of the f16 weights removes.

    python tests/dev/emulate_rounding_bias.py attribute [seed=5] [squares=2048]   # one rounding source at a time
    python tests/dev/emulate_rounding_bias.py correct   [seed=5] [squares=2048]   # the correction, calibration set x test distribution

An f16 MFMA with f32 accumulation is emulated by rounding both operands to f16 and convolving in fp32 (emulate_fp16_classifier.py).
`correct` adds, per f16 layer and output channel, the mean over a calibration batch of conv(x_f16, w - w_f16) before the BatchNorm --
what `ConvLayer::fold_rounding_bias` folds into the epilogue shift on the device ("chan"); "pos" keeps one value per output position
instead (an upper bound on what a per-position shift could add)."""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
sys.path.insert(0, str(Path(__file__).resolve().parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from emulate_fp16_classifier import EXACT, F16, forward  # noqa: E402
from oracle import synth  # noqa: E402


def h(t):
    return t.half().float()


def bn(y, b):
    return F.batch_norm(y, b.running_mean, b.running_var, b.weight, b.bias, False, 0.0, b.eps)


class Emu:
    """`f16r` arithmetic (f32 trunk, exact stem and shortcuts) with an optional per-layer correction c[layer], added before the BN"""

    def __init__(self, net, mode):
        self.net, self.mode, self.c, self.learn = net, mode, {}, False

    def conv(self, name, x, conv, b, exact=False):
        if exact:
            return bn(F.conv2d(x, conv.weight, None, conv.stride, conv.padding), b)
        xq, wq = h(x), h(conv.weight)
        y = F.conv2d(xq, wq, None, conv.stride, conv.padding)
        if self.mode != "none":
            if self.learn:
                e = F.conv2d(xq, conv.weight - wq, None, conv.stride, conv.padding)
                self.c[name] = e.mean(dim=(0, 2, 3), keepdim=True) if self.mode == "chan" else e.mean(dim=0, keepdim=True)
            y = y + self.c[name]
        return bn(y, b)

    def forward(self, x):
        n = self.net
        y = n.maxpool(F.relu(self.conv("conv1", x, n.conv1, n.bn1, exact=True)))
        for li in range(1, 5):
            for bi in range(2):
                blk = getattr(n, f"layer{li}")[bi]
                nm = f"layer{li}.{bi}"
                sc = y if blk.downsample is None else self.conv(nm + ".down", y, blk.downsample[0], blk.downsample[1], exact=True)
                m = F.relu(self.conv(nm + ".conv1", y, blk.conv1, blk.bn1))
                y = F.relu(self.conv(nm + ".conv2", m, blk.conv2, blk.bn2) + sc)
        return n.fc(n.global_pool(y))


def engine_calibration_set():
    """the 128 squares of resnet_load's range calibration (noise, flat levels, ramps, checkers)"""
    host = np.zeros((128, 64, 64), np.float32)
    rs = 0x2545F491

    def rnd():
        nonlocal rs
        rs = (rs * 1664525 + 1013904223) & 0xffffffff
        return (rs >> 24) & 0xff
    ys, xs = np.mgrid[0:64, 0:64]
    for q in range(128):
        if q < 48:
            host[q] = np.array([[rnd() for _ in range(64)] for _ in range(64)])
        elif q < 80:
            host[q] = (q - 48) * 8 + 3
        elif q < 104:
            host[q] = ((xs * (q - 79)) + ys * 3) & 255
        else:
            host[q] = np.where((((xs >> (q & 3)) + (ys >> ((q >> 2) & 3))) & 1) == 1, 235.0, 20 + (q - 104) * 6)
    return torch.from_numpy(host / 255.0)[:, None]


def smooth(n, gen, cells=8):
    base = torch.rand((n, 1, cells, cells), generator=gen)
    return F.interpolate(base, size=64, mode="bilinear", align_corners=False).mul(255).round().div(255)


def report(name, out, ref, p_ref):
    p = torch.softmax(out, 1)
    print(f"{name:60s} soft-max {float((p - p_ref).abs().max()):.2e}  logit max {float((out - ref).abs().max()):.2e}  "
          f"logit rms {float((out - ref).pow(2).mean().sqrt()):.2e}", flush=True)


def attribute(seed, n):
    net = synth.make_resnet(seed=seed)
    sq = synth.squares_input(seed=1000 + seed, n=n)

    def base(nm):
        return EXACT if nm == "conv1" or nm.endswith("down") else F16
    variants = {"f16r": base}
    for layer in ("layer1", "layer2", "layer3", "layer4"):
        variants[f"f16r + {layer} exact"] = lambda nm, layer=layer: EXACT if nm.startswith(layer) else base(nm)
    variants["f16r + every weight exact"] = lambda nm: base(nm) if base(nm) == EXACT else (True, False)
    variants["f16r + every conv input exact"] = lambda nm: base(nm) if base(nm) == EXACT else (False, True)
    variants["f16r + conv2 inputs (mid tensors) exact"] = lambda nm: (False, True) if nm.endswith("conv2") else base(nm)
    variants["f16r + conv1 inputs (f16 copy of the trunk) exact"] = lambda nm: (False, True) if nm.endswith(".conv1") else base(nm)
    with torch.no_grad():
        ref = net(sq)
        p_ref = torch.softmax(ref, 1)
        for name, q in variants.items():
            report(name, torch.cat([forward(net, sq[i:i + 512], True, q) for i in range(0, n, 512)]), ref, p_ref)


def correct(seed, n):
    net = synth.make_resnet(seed=seed)
    g = torch.Generator().manual_seed(7)
    tests = {"noise bytes": synth.squares_input(seed=1000 + seed, n=n), "smooth, 8 cells": smooth(n, g), "smooth, 16 cells": smooth(n, g, 16),
             "near-flat": (smooth(n, g, 2) * 0.3 + 0.5).mul(255).round().div(255)}
    eng = engine_calibration_set()
    cals = {"engine's 128": eng, "noise 128": synth.squares_input(seed=77, n=128),
            "engine's 128 + noise 64 + smooth 64 (shipped)": torch.cat([eng, synth.squares_input(seed=77, n=64), smooth(64, torch.Generator().manual_seed(9), 4)])}
    with torch.no_grad():
        for tname, sq in tests.items():
            ref = net(sq)
            p_ref = torch.softmax(ref, 1)
            for mode in ("none", "chan", "pos"):
                for cname, cal in (cals.items() if mode != "none" else [("-", None)]):
                    em = Emu(net, mode)
                    if cal is not None:
                        em.learn = True
                        em.forward(cal)
                        em.learn = False
                    report(f"test={tname} corr={mode} calib={cname}", torch.cat([em.forward(sq[i:i + 512]) for i in range(0, n, 512)]), ref, p_ref)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "attribute"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    (attribute if what == "attribute" else correct)(seed, n)
