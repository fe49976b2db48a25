"""Workload for `rocprofv3 --kernel-trace --stats`: the single-board shape of the reference's per-image path -- UNet B=1 +
ResNet-18 B=64 through the C ABI, f16x3, N iterations after warm-up (profiles/r04_b1_*)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
from chessvision import synthetic
from chessvision.hip_backend import HipEngine

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
eng = HipEngine(precision=prec)
eng.load_unet(synthetic.unet_state_dict(1)); eng.load_resnet18(synthetic.resnet18_state_dict(2))
x = torch.rand(1, 3, 256, 256, device="cuda"); sq = torch.rand(64, 1, 64, 64, device="cuda")
for _ in range(iters):
    eng.unet_forward(x, check=False); eng.resnet18_forward(sq, check=False)
torch.cuda.synchronize()
eng.check_numerics()
print("done")
