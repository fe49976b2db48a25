"""CPU: the oracle's composed networks (torch nn.Modules, oracle/unet_ref.py / resnet_ref.py) against an independent plain-C,
float64 implementation of the WHOLE forward passes (oracle/c_ref/nets_ref.c), driven from the flat state dict.

The reference holds no tensor-level vector for either network and cannot be imported here, so parity
stays "unpinned" against the reference itself; what this removes is the single-implementation risk on the order of composition:
``cat([skip, up])`` channel order, where ``F.pad`` sits, which tensor the residual add takes, stem -> pool order, stride
placement.  Each of those, deliberately broken in the torch module, moves the output by orders of magnitude more than the
agreement asserted here (the mutation tests below show the C implementation would notice)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import nets_c, synth
from oracle.unet_ref import UNet, Up


def _np_sd(net):
    return {k: v.numpy() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("bilinear", [False, True])
def test_unet_module_matches_the_c_composition(bilinear):
    net = synth.make_unet(1, bilinear).eval()
    x = synth.unet_input(5, 2, hw=64).numpy()
    with torch.no_grad():
        want = net(torch.from_numpy(x)).numpy()
    got = nets_c.unet_forward(_np_sd(net), x)
    scale = max(1.0, float(np.abs(got).max()))
    assert np.abs(got - want).max() <= 2e-5 * scale, float(np.abs(got - want).max())
    assert np.abs(got).max() > 0.5                            # a real signal, not zeros agreeing with zeros


def test_unet_padding_path_on_a_size_that_is_not_a_multiple_of_16():
    """72 x 56: the deepest maps are 4 x 3 and every decoder stage must pad its up-sampled tensor (odd sizes on the way down);
    pins the F.pad placement [left, right, top, bottom] = [d // 2, d - d // 2] of Up.forward."""
    net = synth.make_unet(3, False).eval()
    x = synth.unet_input(6, 1, hw=72)[:, :, :, :56].contiguous().numpy()
    with torch.no_grad():
        want = net(torch.from_numpy(x)).numpy()
    got = nets_c.unet_forward(_np_sd(net), x)
    assert got.shape == want.shape == (1, 1, 72, 56)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, float(np.abs(got).max()))


def test_resnet18_module_matches_the_c_composition():
    net = synth.make_resnet(2).eval()
    x = synth.squares_input(4, 6).numpy()
    with torch.no_grad():
        want = net(torch.from_numpy(x)).numpy()
    got = nets_c.resnet18_forward(_np_sd(net), x)
    assert got.shape == (6, 13)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, float(np.abs(got).max())), float(np.abs(got - want).max())
    assert np.abs(got).max() > 0.1


def test_the_c_composition_notices_a_swapped_concatenation_and_a_misplaced_residual(monkeypatch):
    """Sensitivity: the agreement above is not vacuous.  (i) cat([up, skip]) instead of cat([skip, up]); (ii) the residual taken
    AFTER the block's last ReLU instead of before it: both leave every op and every shape intact and are invisible to the
    per-op cross-check, and both are far outside the tolerance of the composed check."""
    net = synth.make_unet(1, False).eval()
    x = synth.unet_input(5, 1, hw=64).numpy()
    ref = nets_c.unet_forward(_np_sd(net), x)

    def swapped(self, deep, skip):
        deep = self.up(deep)
        return self.conv(torch.cat([deep, skip], dim=1))

    monkeypatch.setattr(Up, "forward", swapped)
    with torch.no_grad():
        wrong = net(torch.from_numpy(x)).numpy()
    assert np.abs(wrong - ref).max() > 1e-2 * max(1.0, float(np.abs(ref).max()))

    from oracle import resnet_ref

    rnet = synth.make_resnet(2).eval()
    sq = synth.squares_input(4, 2).numpy()
    rref = nets_c.resnet18_forward(_np_sd(rnet), sq)
    block_cls = type(rnet.layer1[0])

    def late_residual(self, x):
        out = self.act1(self.bn1(self.conv1(x)))
        out = self.act2(self.bn2(self.conv2(out)))                      # ReLU BEFORE the add: wrong
        return out + (self.downsample(x) if self.downsample is not None else x)

    monkeypatch.setattr(block_cls, "forward", late_residual)
    with torch.no_grad():
        rwrong = rnet(torch.from_numpy(sq)).numpy()
    assert np.abs(rwrong - rref).max() > 1e-2 * max(1.0, float(np.abs(rref).max()))
    assert resnet_ref is not None


def test_key_and_shape_validation_of_the_c_walk():
    sd = _np_sd(synth.make_resnet(2))
    sd.pop("layer3.0.downsample.0.weight")
    with pytest.raises(ValueError, match="layer3.0.downsample.0.weight"):
        nets_c.resnet18_forward(sd, np.zeros((1, 1, 64, 64), np.float32))
    sd = _np_sd(synth.make_unet(1, False))
    sd["up2.up.weight"] = sd["up2.up.weight"][:, :64]
    with pytest.raises(ValueError, match="up2.up.weight"):
        nets_c.unet_forward(sd, np.zeros((1, 3, 32, 32), np.float32))
