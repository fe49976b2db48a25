"""CPU: the product-side checkpoint shape tables / generators agree with the oracle's module trees."""
from __future__ import annotations

import numpy as np
import pytest

from chessvision import synthetic
from oracle import resnet_ref, synth, unet_ref


@pytest.mark.parametrize("bilinear", [False, True])
def test_unet_spec_matches_oracle_state_dict(bilinear):
    sd = unet_ref.UNet(3, 1, bilinear).state_dict()
    spec = {k: tuple(s) for k, s, _ in synthetic.unet_spec(bilinear)}
    want = {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    assert spec == want
    assert [k for k, _, _ in synthetic.unet_spec(bilinear)] == [k for k in sd if not k.endswith("num_batches_tracked")]


def test_resnet_spec_matches_oracle_state_dict():
    sd = resnet_ref.ResNet18().state_dict()
    spec = {k: tuple(s) for k, s, _ in synthetic.resnet18_spec()}
    assert spec == {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("num_batches_tracked")}


def test_generators_agree_bit_for_bit():
    a = synthetic.resnet18_state_dict(2)
    b = synth.synth_state_dict(resnet_ref.ResNet18(), 2, residual_gamma=0.5)
    for k, v in a.items():
        assert np.array_equal(v, b[k].numpy()), k
    a = synthetic.unet_state_dict(1, bilinear=True)
    b = synth.synth_state_dict(unet_ref.UNet(3, 1, True), 1)
    for k, v in a.items():
        assert np.array_equal(v, b[k].numpy()), k


def test_checkpoints_written_in_reference_formats(tmp_path):
    import torch

    pe, pc = synthetic.save_checkpoints(tmp_path)
    ck = torch.load(pe, map_location="cpu", weights_only=False)
    assert set(ck) == {"model_state_dict", "metadata"}                     # scripts/train/train_unet.py:31-40
    ck = torch.load(pc, map_location="cpu", weights_only=False)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "metadata"}   # train_classifier.py:114-125
    unet_ref.UNet(3, 1).load_state_dict(torch.load(pe, map_location="cpu", weights_only=False)["model_state_dict"], strict=False)
