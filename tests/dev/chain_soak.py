"""Soak of the chained layer1 launch (f16r): many forwards at full occupancy and at ragged sizes, every output compared bit for bit
with the first one of its size and with the four-launch schedule's checksum (developer tool; prints a summary line)."""
import hashlib, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
net = synth.make_resnet(seed=2)
eng = HipEngine(precision="f16r", resnet_chunk=16384)
eng.load_resnet18(net.state_dict())
g = torch.Generator(device="cuda"); g.manual_seed(5)
big = torch.randint(0, 256, (16384, 1, 64, 64), dtype=torch.uint8, device="cuda", generator=g).float().div_(255)
bad = 0
first = {}
for it in range(iters):
    n = 16384 if it % 3 else [16384, 3000, 777, 64, 9000][(it // 3) % 5]
    out = eng.resnet18_forward(big[:n], check=False)
    key = n
    if key not in first:
        first[key] = out.clone()
    elif not torch.equal(first[key], out):
        bad += 1
eng.check_numerics()
sha = hashlib.sha256(first[16384].cpu().numpy().tobytes()).hexdigest()
print(f"chain={os.environ.get('CV_RESNET_CHAIN', '1')} form={os.environ.get('CV_CHAIN_WG', '2')}: {iters} forwards, {bad} differing from the first of their size, sha {sha[:16]}")
