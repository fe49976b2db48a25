"""The 2 x 2-block split-f16 up-sample kernel against the one-output-per-lane kernel (CV_UPSAMPLE_2X2=0, child process): outputs compared
bit for bit on several geometries, and against torch (developer tool)."""
import hashlib, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
import torch.nn.functional as F
from chessvision.hip_backend import HipEngine

def run():
    eng = HipEngine(precision="f16x3")
    out = {}
    g = torch.Generator().manual_seed(11)
    for shape in ((2, 32, 16, 16), (1, 512, 16, 16), (3, 64, 128, 128), (1, 8, 1, 1), (2, 16, 1, 7), (1, 24, 5, 3), (1, 256, 32, 32)):
        x = torch.randn(shape, generator=g) * 3
        y = eng.op_upsample_bilinear2x(x).cpu()
        ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        out[shape] = (hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16], float((y - ref).abs().max()))
    return out

if __name__ == "__main__":
    if os.environ.get("UPS_CHILD"):
        print(repr(run())); sys.exit(0)
    a = run()
    env = dict(os.environ, CV_UPSAMPLE_2X2="0", UPS_CHILD="1")
    b = eval(subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    for k in a:
        print(k, "same bits" if a[k][0] == b[k][0] else "DIFFERENT", f"err vs torch {a[k][1]:.2e} / {b[k][1]:.2e}")
