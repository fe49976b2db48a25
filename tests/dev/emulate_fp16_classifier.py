"""(test infrastructure, not collected by pytest) CPU emulation of the classifier's fp16 arithmetic, one rounding source at a time.

    python tests/dev/emulate_fp16_classifier.py [squares=4096] [he|stress] [input seed=42]

An f16 MFMA with f32 accumulation is emulated exactly (up to summation order) by rounding both operands to f16 and convolving in
fp32 on the CPU.  The script runs the oracle's ResNet-18 with selectable rounding of (i) convolution inputs, (ii) weights, (iii) the
stored residual trunk, per layer, and prints the worst soft-max / logit error against the fp32 oracle -- the table in DESIGN.md
section 2 that motivated precision "f16r" (f32 trunk + exact shortcut convolutions) comes from here; the GPU numbers it predicts
(f16: 1.41e-3 vs 1.38e-3 measured; f16r: 0.65e-3 vs 0.64e-3) are in profiles/r03_f16r_probe.jsonl.  The stressed weights overflow
here (no range scaling in the emulation); use the GPU probe for those.
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import synth  # noqa: E402


def h(t):
    return t.half().float()


def conv_bn(x, conv, bn, xq, wq):
    w = h(conv.weight) if wq else conv.weight
    x = h(x) if xq else x
    y = F.conv2d(x, w, None, conv.stride, conv.padding)
    return F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


def forward(net, x, trunk32, q):
    """q(layer name) -> (round the conv input, round the weights); trunk32: keep block outputs / shortcuts / stem output unrounded."""
    y = net.maxpool(F.relu(conv_bn(x, net.conv1, net.bn1, *q("conv1"))))
    if not trunk32:
        y = h(y)
    for li in range(1, 5):
        for bi in range(2):
            blk = getattr(net, f"layer{li}")[bi]
            name = f"layer{li}.{bi}"
            if blk.downsample is None:
                sc = y
            else:
                sc = conv_bn(y, blk.downsample[0], blk.downsample[1], *q(name + ".down"))
                if not trunk32:
                    sc = h(sc)
            m = F.relu(conv_bn(y, blk.conv1, blk.bn1, *q(name + ".conv1")))
            y = F.relu(conv_bn(m, blk.conv2, blk.bn2, *q(name + ".conv2")) + sc)
            if not trunk32:
                y = h(y)
    return net.fc(net.global_pool(y))


EXACT, F16 = (False, False), (True, True)
VARIANTS = {
    "f16 (everything rounded)": (False, lambda n: F16),
    "f32 trunk": (True, lambda n: EXACT if n == "conv1" else F16),
    "f32 trunk + exact shortcut convs  (= f16r)": (True, lambda n: EXACT if n == "conv1" or n.endswith("down") else F16),
    "f16r + exact weights everywhere (2 products)": (True, lambda n: EXACT if n == "conv1" or n.endswith("down") else (True, False)),
    "f16r + exact conv inputs everywhere": (True, lambda n: EXACT if n == "conv1" or n.endswith("down") else (False, True)),
    "f32 trunk, f16 shortcuts, stem + layer1 + layer2 exact": (True, lambda n: EXACT if n == "conv1" or n[:6] in ("layer1", "layer2") else F16),
    "f32 trunk, every conv1 + shortcut exact": (True, lambda n: EXACT if n == "conv1" or n.endswith(("conv1", "down")) else F16),
}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    wsel = sys.argv[2] if len(sys.argv) > 2 else "he"
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
    net = synth.make_resnet(seed=2)
    if wsel == "stress":
        synth.load(net, synth.stress_resnet_state_dict(2))
    sq = synth.squares_input(seed=seed, n=n)
    with torch.no_grad():
        ref = net(sq)
        p_ref = torch.softmax(ref, 1)
        for name, (trunk32, q) in VARIANTS.items():
            t0 = time.time()
            out = torch.cat([forward(net, sq[i:i + 512], trunk32, q) for i in range(0, n, 512)])
            p = torch.softmax(out, 1)
            print(json.dumps({"variant": name, "weights": wsel, "squares": n, "prob_err": float((p - p_ref).abs().max()),
                              "logit_err": float((out - ref).abs().max()), "logit_rms": float((out - ref).pow(2).mean().sqrt()),
                              "s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
