"""GPU: the HIP forward passes against the independent plain-C float64 composition of both networks (oracle/c_ref/nets_ref.c).

The other GPU parity tests compare with oracle/unet_ref.py / resnet_ref.py (torch nn.Modules on CPU).  This one takes the second,
code-independent implementation of the WHOLE networks as the judge (VERDICT r03 'next' 4): same state dict in, the HIP logits must
land within north_star's 1e-3 of the float64 function of the checkpoint -- at exactly the batch sizes the reference's per-image
path runs (UNet B = 1, ResNet-18 B = 64; reference core.py:215-220, 236-241), which are also the split-K launches of round 4."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from chessvision.hip_backend import HipEngine
from oracle import nets_c, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bilinear", [False, True])
def test_unet_b1_against_the_c_composition(bilinear):
    net = synth.make_unet(1, bilinear).eval()
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    x = synth.unet_input(5, 1)
    want = nets_c.unet_forward(sd, x.numpy())                   # float64, (1,1,256,256)
    scale = max(1.0, float(np.abs(want).max()))
    for prec, tol in (("f16x3", 1e-3), ("f32", 1e-3)):
        eng = HipEngine("cuda:0", precision=prec)
        eng.load_unet(net.state_dict())
        got = eng.unet_forward(x.cuda()).cpu().numpy().astype(np.float64)
        again = eng.unet_forward(x.cuda()).cpu().numpy().astype(np.float64)
        assert np.array_equal(got, again)                       # deterministic run to run (fixed split-K summation order)
        err = float(np.abs(got - want).max())
        assert err <= tol * scale, (prec, bilinear, err)
        assert ((got > 0) == (want > 0)).mean() >= 0.9999       # the mask the pipeline thresholds
        eng.close()


def test_resnet18_b64_against_the_c_composition():
    net = synth.make_resnet(2).eval()
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    sq = synth.squares_input(4, 64)
    want = nets_c.resnet18_forward(sd, sq.numpy())              # float64, (64,13)
    scale = max(1.0, float(np.abs(want).max()))
    for prec, tol in (("f16x3", 1e-3), ("f32", 1e-3), ("f16r", 5e-3)):
        eng = HipEngine("cuda:0", precision=prec)
        eng.load_resnet18(net.state_dict())
        got = eng.resnet18_forward(sq.cuda()).cpu().numpy().astype(np.float64)
        again = eng.resnet18_forward(sq.cuda()).cpu().numpy().astype(np.float64)
        assert np.array_equal(got, again)
        err = float(np.abs(got - want).max())
        assert err <= tol * scale, (prec, err)
        assert np.array_equal(got.argmax(1), want.argmax(1))
        if prec == "f16r":                                      # configs[2]'s bar: soft-max probabilities within 1e-3
            pg = torch.softmax(torch.from_numpy(got), 1).numpy()
            pw = torch.softmax(torch.from_numpy(want), 1).numpy()
            assert np.abs(pg - pw).max() <= 1e-3
        eng.close()
