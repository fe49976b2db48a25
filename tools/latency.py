"""Single-board latency through the C ABI (the Flask use case): UNet B=1 + ResNet-18 B=64, per precision."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
from chessvision import synthetic
from chessvision.hip_backend import HipEngine

for prec in ("f16x3", "f32", "f16"):
    eng = HipEngine(precision=prec, unet_chunk=64, resnet_chunk=16384)
    eng.load_unet(synthetic.unet_state_dict(1)); eng.load_resnet18(synthetic.resnet18_state_dict(2))
    x = torch.rand(1, 3, 256, 256, device="cuda"); sq = torch.rand(64, 1, 64, 64, device="cuda")
    for _ in range(5):
        eng.unet_forward(x); eng.resnet18_forward(sq)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("unet_b1", lambda: eng.unet_forward(x)), ("resnet_b64", lambda: eng.resnet18_forward(sq))):
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 50 * 1e3
    print(f"{prec}: UNet B=1 {res['unet_b1']:.3f} ms, ResNet-18 B=64 {res['resnet_b64']:.3f} ms, board {sum(res.values()):.3f} ms")
    eng.close()
