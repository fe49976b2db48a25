"""CPU: pin the oracle's module trees against everything the reference itself fixes about them
(SURVEY.md section 8c -- there are no golden tensors in the reference, so these structural pins, the C
cross-check in test_oracle_ops.py and the committed golden fixtures are what the oracle stands on)."""
from __future__ import annotations

import torch

from oracle import resnet_ref, unet_ref


def test_unet_module_tree_pins():
    for bilinear, params in ((False, 31_037_633), (True, 17_262_977)):
        net = unet_ref.UNet(3, 1, bilinear)
        names = [n for n, _ in net.named_modules()]
        assert len(names) == 95
        # embedding layer index used by the reference: scripts/train/train_unet.py:210,219
        assert names[52] == "down4.maxpool_conv.1.double_conv.5"
        assert isinstance(dict(net.named_modules())[names[52]], torch.nn.ReLU)
        assert sum(p.numel() for p in net.parameters()) == params
        assert net.n_channels == 3                      # attribute read at train_unet.py:296


def test_unet_state_dict_keys_follow_checkpoint_format():
    sd = unet_ref.UNet(3, 1, False).state_dict()
    for key, shape in {"inc.double_conv.0.weight": (64, 3, 3, 3), "inc.double_conv.4.running_var": (64,),
                       "down4.maxpool_conv.1.double_conv.3.weight": (1024, 1024, 3, 3),
                       "up1.up.weight": (1024, 512, 2, 2), "up1.up.bias": (512,),
                       "up4.conv.double_conv.0.weight": (64, 128, 3, 3), "outc.conv.weight": (1, 64, 1, 1),
                       "outc.conv.bias": (1,)}.items():
        assert tuple(sd[key].shape) == shape, key
    assert "up1.up.weight" not in unet_ref.UNet(3, 1, True).state_dict()


def test_unet_macs_match_survey():
    assert unet_ref.unet_macs(False) == 48_167_387_136      # 48.167 GMAC, SURVEY.md Appendix A
    assert unet_ref.unet_macs(True) == 39_980_105_728       # 39.980 GMAC


def test_unet_output_contract():
    net = unet_ref.UNet(3, 1).eval()
    with torch.no_grad():
        y = net(torch.zeros(1, 3, 64, 64))
    assert y.shape == (1, 1, 64, 64)                         # (B,1,H,W) logits consumed by core.py:220


# the exact listing printed by notebooks/model-summary.ipynb (cell 3) in the reference
_BLOCK = ["conv1", "bn1", "drop_block", "act1", "aa", "conv2", "bn2", "act2"]


def _expected_resnet_names():
    names = ["", "conv1", "bn1", "act1", "maxpool"]
    for layer in range(1, 5):
        names.append(f"layer{layer}")
        for block in range(2):
            names.append(f"layer{layer}.{block}")
            names += [f"layer{layer}.{block}.{m}" for m in _BLOCK]
            if block == 0 and layer > 1:
                names += [f"layer{layer}.0.downsample", f"layer{layer}.0.downsample.0", f"layer{layer}.0.downsample.1"]
    return names + ["global_pool", "global_pool.pool", "global_pool.flatten", "fc"]


def test_resnet18_module_tree_matches_reference_notebook():
    net = resnet_ref.ResNet18()
    names = [n for n, _ in net.named_modules()]
    assert names == _expected_resnet_names()
    assert len(names) == 94 and names[90] == "global_pool"   # scripts/train/train_classifier.py:32
    assert sum(p.numel() for p in net.parameters()) == 11_176_909   # model-summary.ipynb:233


def test_resnet18_layer_shapes_match_reference_notebook():
    net = resnet_ref.ResNet18().eval()
    shapes = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: shapes.__setitem__(n, tuple(o.shape)))
             for n, m in net.named_modules() if n in ("conv1", "maxpool", "layer1", "layer2", "layer3", "layer4", "fc")]
    with torch.no_grad():
        net(torch.zeros(1, 1, 64, 64))
    for h in hooks:
        h.remove()
    assert shapes == {"conv1": (1, 64, 32, 32), "maxpool": (1, 64, 16, 16), "layer1": (1, 64, 16, 16),
                      "layer2": (1, 128, 8, 8), "layer3": (1, 256, 4, 4), "layer4": (1, 512, 2, 2), "fc": (1, 13)}
    # 141.64 M mult-adds in the notebook (torchinfo rounds); exact count of the restated tree:
    assert resnet_ref.resnet18_macs() == 141_629_952


def test_range_stressed_networks_compute_the_same_function():
    """oracle/synth.py: stress_*_state_dict rescale weights / BatchNorm statistics by up to 1e+-2 per layer and push three
    activations to 1e3..1e5 while leaving the network function unchanged -- the premise of tests/test_gpu_numerics.py."""
    from oracle import synth

    x, sq = synth.unet_input(3, 1), synth.squares_input(4, 64)
    with torch.no_grad():
        a = synth.make_unet(1)(x)
        sd = synth.stress_unet_state_dict(1)
        b = synth.load(unet_ref.UNet(3, 1, False), sd)(x)
        c = synth.make_resnet(2)(sq)
        d = synth.load(resnet_ref.ResNet18(), synth.stress_resnet_state_dict(2))(sq)
    assert float((a - b).abs().max()) <= 2e-4 and float((c - d).abs().max()) <= 5e-5
    var = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("running_var")])
    assert float(var.min()) < 1e-3 and float(var.max()) > 1e3                   # BatchNorm scales from ~0.01x to ~100x
    seen = {}
    net = synth.load(unet_ref.UNet(3, 1, False), sd)
    hooks = [m.register_forward_hook(lambda m, i, o, n=n: seen.__setitem__(n, float(o.abs().max())))
             for n, m in net.named_modules() if isinstance(m, torch.nn.ReLU)]
    with torch.no_grad():
        net(x)
    for h in hooks:
        h.remove()
    assert seen["down3.maxpool_conv.1.double_conv.2"] > 65504 < seen["up2.conv.double_conv.2"]   # beyond the f16 range
    assert seen["down2.maxpool_conv.1.double_conv.5"] > 65504                                     # a skip tensor, too
