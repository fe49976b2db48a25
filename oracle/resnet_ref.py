"""CPU oracle: timm-style ResNet-18 (in_chans=1, num_classes=13) forward, torch fp32.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) -- never imported by the product path.

Restates what ``timm.create_model("resnet18", num_classes=13, in_chans=1)`` builds for the
reference (``chessvision/utils.py:32-39``; timm 1.0.15 per ``uv.lock:4079-4080``, not installed
here).  Pinned by the reference's own dump ``notebooks/model-summary.ipynb``: 94 modules in the
order listed there (``[90] == global_pool`` is what ``scripts/train/train_classifier.py:32``
hooks), 11,176,909 parameters, 141.64 M mult-adds at 1x1x64x64.

BasicBlock order (notebook lines 31-124): conv1 -> bn1 -> drop_block(Identity) -> act1 ->
aa(Identity) -> conv2 -> bn2 -> (+shortcut) -> act2; ``downsample`` = [conv1x1 stride 2, BN] on
the first block of layer2-4.
"""
from __future__ import annotations

import torch
import torch.nn as nn

BN_EPS = 1e-5
NUM_CLASSES = 13


class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout, eps=BN_EPS)
        self.drop_block = nn.Identity()
        self.act1 = nn.ReLU(inplace=True)
        self.aa = nn.Identity()
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout, eps=BN_EPS)
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout, eps=BN_EPS))

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.act1(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.act2(y + shortcut)


class _GlobalPool(nn.Module):
    """Mirrors timm's SelectAdaptivePool2d(avg, flatten=True): children ``pool`` and ``flatten``."""

    def __init__(self):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.flatten = nn.Flatten(1)

    def forward(self, x):
        return self.flatten(self.pool(x))


class ResNet18(nn.Module):
    def __init__(self, num_classes: int = NUM_CLASSES, in_chans: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=BN_EPS)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        widths = [64, 128, 256, 512]
        cin = 64
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(BasicBlock(cin, w, stride), BasicBlock(w, w, 1)))
            cin = w
        self.global_pool = _GlobalPool()
        self.fc = nn.Linear(512, num_classes)

    def forward_features(self, x):
        x = self.maxpool(self.act1(self.bn1(self.conv1(x))))
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return x

    def forward(self, x):
        return self.fc(self.global_pool(self.forward_features(x)))


def resnet18_macs(hw: int = 64) -> int:
    net = ResNet18().eval()
    total = 0

    def hook(mod, inp, out):
        nonlocal total
        if isinstance(mod, nn.Conv2d):
            total += out.numel() * mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1]
        else:
            total += out.numel() * mod.in_features

    hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
    with torch.no_grad():
        net(torch.zeros(1, 1, hw, hw))
    for h in hs:
        h.remove()
    return total
