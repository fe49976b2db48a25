"""The reference-side binding INTEGRATION.md shows (section B, ``chessvision/hip_models.py``) is executed verbatim against
the built library: a maintainer pasting it gets the same numbers as this repository's own shim."""
from __future__ import annotations

import re
from pathlib import Path

import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _stub_namespace():
    from chessvision.hip_backend import library_path

    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class _HipModel" in b)
    stub = stub.replace('ctypes.CDLL("libchessvision_hip.so")', f'ctypes.CDLL("{library_path()}")')
    ns: dict = {}
    exec(compile(stub, "INTEGRATION.md:hip_models.py", "exec"), ns)
    return ns


def test_documented_ctypes_stub_runs_both_models():
    from chessvision.hip_backend import HipEngine

    ns = _stub_namespace()
    unet, resnet = synth.make_unet(seed=1), synth.make_resnet(seed=2)
    x = synth.unet_input(seed=21, batch=2).cuda()
    sq = synth.squares_input(seed=22, n=64).cuda()
    m_u = ns["load_hip_unet"](unet.state_dict())
    m_r = ns["load_hip_resnet18"](resnet.state_dict())
    assert m_u.eval() is m_u and m_u.to("cuda") is m_u
    y_u, y_r = m_u(x), m_r(sq)
    assert tuple(y_u.shape) == (2, 1, 256, 256) and tuple(y_r.shape) == (64, 13)
    eng = HipEngine(precision="f16x3", unet_chunk=2, resnet_chunk=64)
    eng.load_unet(unet.state_dict())
    eng.load_resnet18(resnet.state_dict())
    torch.cuda.synchronize()
    assert torch.equal(y_u, eng.unet_forward(x))
    assert torch.equal(y_r, eng.resnet18_forward(sq))
    with torch.no_grad():
        assert float((y_u.cpu() - unet(x.cpu())).abs().max()) <= 1e-3
        assert float((y_r.cpu() - resnet(sq.cpu())).abs().max()) <= 1e-3
    eng.close()
