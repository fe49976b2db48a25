"""Minimal workload for rocprofv3 --pmc passes: N forwards of one model at one precision (no event timing).

usage: python3 tools/pmc_run.py <f32|f16|f16x3> <unet|resnet18> [batch] [iters]
"""
import os
import sys
from pathlib import Path

# counter passes average over every dispatch of a kernel name: keep the load-time range calibration (small launches of the same
# kernels) out of them.  Exponents do not change what the kernels do per byte or per MFMA.
os.environ.setdefault("CV_CALIBRATE", "0")

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch  # noqa: E402
from chessvision import synthetic  # noqa: E402
from chessvision.hip_backend import HipEngine  # noqa: E402

prec, model = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else (64 if model == "unet" else 16384)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
eng = HipEngine(precision=prec, unet_chunk=64, resnet_chunk=16384)
g = torch.Generator(device="cuda"); g.manual_seed(7)
if model == "unet":
    eng.load_unet(synthetic.unet_state_dict(1))
    x = torch.randint(0, 256, (batch, 3, 256, 256), dtype=torch.uint8, device="cuda", generator=g).float() / 255
    for _ in range(iters):
        eng.unet_forward(x)
else:
    eng.load_resnet18(synthetic.resnet18_state_dict(2))
    x = torch.randint(0, 256, (batch, 1, 64, 64), dtype=torch.uint8, device="cuda", generator=g).float() / 255
    for _ in range(iters):
        eng.resnet18_forward(x)
torch.cuda.synchronize()
