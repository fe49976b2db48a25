"""Repeat one ResNet-18 forward and report how many outputs differ between repeats (developer tool)."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth

prec = sys.argv[1] if len(sys.argv) > 1 else "f16r"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 128
net = synth.make_resnet(seed=2)
eng = HipEngine(precision=prec, resnet_chunk=chunk)
eng.load_resnet18(net.state_dict())
x = synth.squares_input(seed=4, n=n).cuda()
NAMES = ("maxpool", "layer1", "layer2.0.downsample", "layer2", "layer3", "layer4.0.downsample", "layer4")
outs, taps = [], []
for _ in range(4):
    outs.append(eng.resnet18_forward(x).cpu())
    taps.append({name: torch.from_numpy(eng.activation("resnet18", name)) for name in NAMES})
for i in range(1, 4):
    print("   taps differing from repeat 0:", {k: int((taps[i][k] != taps[0][k]).sum()) for k in NAMES})
for i in range(1, 4):
    d = (outs[i] - outs[0]).abs()
    print(f"chain={os.environ.get('CV_RESNET_CHAIN','1')} fast_sc={os.environ.get('CV_SHORTCUT_FAST','1')} n={n} chunk={chunk} repeat {i}: {int((d > 0).sum())} of {d.numel()} logits differ, max {float(d.max()):.3e}, rows {sorted(set((d > 0).nonzero()[:, 0].tolist()))[:8]}")
eng.close()
