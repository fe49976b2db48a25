"""GPU: the experiment switches of the conv path give the same network.

The library reads its knobs from the environment once per process, so every variant runs in a child process: a UNet forward at one
and at three boards and a ResNet-18 forward at 64 and 300 squares (f16x3), checked against the CPU oracle to north_star's 1e-3.
Covers the paths the default configuration never takes on the test box: split-K off / forced on every launch that can take it
(both kernels), hipGraph replay off, the 8 x 16 halo tile off, the persistent form of the production tile (`CV_HALO_PERSIST64=1`,
ADVICE r03) and the fused first two convolutions off."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r"""
import sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth

import os
boards = [int(v) for v in os.environ.get("KNOB_BOARDS", "1,3").split(",")]
squares = [int(v) for v in os.environ.get("KNOB_SQUARES", "64,300").split(",")]
unet, resnet = synth.make_unet(1), synth.make_resnet(2)
eng = HipEngine(precision="f16x3", unet_chunk=max(2, max(boards)), resnet_chunk=max(256, max(squares)))
eng.load_unet(unet.state_dict()); eng.load_resnet18(resnet.state_dict())
worst = 0.0
for b in boards:
    x = synth.unet_input(60 + b, b)
    with torch.no_grad():
        ref = unet(x)
    for _ in range(3):                                        # eager, capture, replay
        got = eng.unet_forward(x.cuda()).cpu()
    worst = max(worst, float((got - ref).abs().max()))
for n in squares:
    sq = synth.squares_input(70 + n, n)
    with torch.no_grad():
        ref = resnet(sq)
    for _ in range(3):
        got = eng.resnet18_forward(sq.cuda()).cpu()
    worst = max(worst, float((got - ref).abs().max()))
eng.check_numerics()
print("WORST", worst)
assert worst <= 1e-3, worst
print("KNOBS_OK")
"""

VARIANTS = [
    {"CV_SPLITK": "0"},
    {"CV_SPLITK_FORCE": "3"},
    {"CV_SPLITK_FORCE": "5", "CV_SPLITK_HALO": "0"},
    {"CV_GRAPH": "0", "CV_HALO_TH8": "0"},
    # >= 8 tiles per CU: up4.conv.0 at 8 boards (2048 tiles, 36 stages), layer1.x.conv1 at 2048 squares
    {"CV_HALO_PERSIST64": "1", "CV_HALO_PERSIST64_MAXK": "72", "KNOB_BOARDS": "8", "KNOB_SQUARES": "2048"},
    {"CV_FUSE_INC": "0", "CV_FUSE_POOL": "0"},
]


@pytest.mark.parametrize("knobs", VARIANTS, ids=[",".join(f"{k}={v}" for k, v in kv.items()) for kv in VARIANTS])
def test_forward_passes_under_experiment_switches(knobs):
    env = dict(os.environ)
    env.update(knobs)
    out = subprocess.run([sys.executable, "-c", SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "KNOBS_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
