"""A/B (one box): the 256-board step on ONE engine / one stream against TWO engines on two streams, each taking half of the boards
(the tails and launch gaps of one stream's kernels filled by the other's).  Developer experiment, r04_tuning.md."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
from chessvision import synthetic
from chessvision.hip_backend import HipEngine

B = 256
usd, rsd = synthetic.unet_state_dict(1), synthetic.resnet18_state_dict(2)
def make():
    e = HipEngine(precision="f16x3"); e.load_unet(usd); e.load_resnet18(rsd); return e
e0, e1 = make(), make()
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randint(0, 256, (B, 3, 256, 256), dtype=torch.uint8, device="cuda", generator=g).float() / 255
sq = torch.randint(0, 256, (B * 64, 1, 64, 64), dtype=torch.uint8, device="cuda", generator=g).float() / 255
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    return e0.unet_forward(x, check=False), e0.resnet18_forward(sq, check=False)

def two(order):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    h = B // 2
    if order == "same":            # both streams walk UNet then ResNet
        with torch.cuda.stream(sa):
            a = e0.unet_forward(x[:h], check=False); c = e0.resnet18_forward(sq[:h * 64], check=False)
        with torch.cuda.stream(sb):
            b = e1.unet_forward(x[h:], check=False); d = e1.resnet18_forward(sq[h * 64:], check=False)
    else:                           # staggered: B starts with its classifier
        with torch.cuda.stream(sa):
            a = e0.unet_forward(x[:h], check=False); c = e0.resnet18_forward(sq[:h * 64], check=False)
        with torch.cuda.stream(sb):
            d = e1.resnet18_forward(sq[h * 64:], check=False); b = e1.unet_forward(x[h:], check=False)
    cur.wait_stream(sa); cur.wait_stream(sb)
    return a, b, c, d

def timed(fn, steps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for rep in range(2):
    print(f"one engine, one stream : {timed(one):7.2f} ms per 256 boards")
    print(f"two engines, two streams: {timed(lambda: two('same')):7.2f} ms (same order)  {timed(lambda: two('staggered')):7.2f} ms (staggered)")
e0.check_numerics(); e1.check_numerics()
