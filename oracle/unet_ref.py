"""CPU oracle: UNet(n_channels=3, n_classes=1[, bilinear]) forward, torch fp32.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) -- never imported by the product path.

Restates the module tree the reference constructs at ``chessvision/core.py:88`` (source lives
in the empty submodule ``chessvision/pytorch_unet``; layout per SURVEY.md Appendix A).
Pinned structurally by the reference: 95 modules, ``named_modules()[52] ==
down4.maxpool_conv.1.double_conv.5`` (``scripts/train/train_unet.py:210,219``), 31,037,633
parameters (17,262,977 with ``bilinear=True``).

State-dict keys (the checkpoint format ``scripts/train/train_unet.py:31-40`` writes):
``inc.double_conv.{0,3}.weight``, ``inc.double_conv.{1,4}.{weight,bias,running_mean,running_var,
num_batches_tracked}``, ``down{1-4}.maxpool_conv.1.double_conv.*``, ``up{1-4}.up.{weight,bias}``
(transposed-conv variant only), ``up{1-4}.conv.double_conv.*``, ``outc.conv.{weight,bias}``.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-5


class DoubleConv(nn.Module):
    """[conv3x3(no bias) -> BN -> ReLU] x 2; attribute name ``double_conv`` is part of the key pin."""

    def __init__(self, cin: int, cout: int, cmid: int | None = None):
        super().__init__()
        cmid = cmid or cout
        self.double_conv = nn.Sequential(
            nn.Conv2d(cin, cmid, 3, padding=1, bias=False), nn.BatchNorm2d(cmid, eps=BN_EPS), nn.ReLU(inplace=True),
            nn.Conv2d(cmid, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout, eps=BN_EPS), nn.ReLU(inplace=True),
        )

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(cin, cout))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    def __init__(self, cin: int, cout: int, bilinear: bool):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(cin, cout, cin // 2)
        else:
            self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(cin, cout)

    def forward(self, deep, skip):
        deep = self.up(deep)
        dy = skip.shape[2] - deep.shape[2]
        dx = skip.shape[3] - deep.shape[3]
        if dy or dx:                                    # zero-size at 256x256
            deep = F.pad(deep, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([skip, deep], dim=1))   # skip first, then upsampled


class OutConv(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    def __init__(self, n_channels: int = 3, n_classes: int = 1, bilinear: bool = False):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        f = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 1024 // f)
        self.up1 = Up(1024, 512 // f, bilinear)
        self.up2 = Up(512, 256 // f, bilinear)
        self.up3 = Up(256, 128 // f, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)

    def forward(self, x):
        s1 = self.inc(x)
        s2 = self.down1(s1)
        s3 = self.down2(s2)
        s4 = self.down3(s3)
        y = self.down4(s4)
        y = self.up1(y, s4)
        y = self.up2(y, s3)
        y = self.up3(y, s2)
        y = self.up4(y, s1)
        return self.outc(y)


def unet_macs(bilinear: bool = False, hw: int = 256) -> int:
    """Multiply-accumulates of one forward at hw x hw (conv / conv-transpose only)."""
    net = UNet(3, 1, bilinear)
    total = 0
    hooks = []

    def hook(mod, inp, out):
        nonlocal total
        if isinstance(mod, nn.Conv2d):
            total += out.numel() * mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1]
        elif isinstance(mod, nn.ConvTranspose2d):
            total += inp[0].numel() * mod.out_channels * mod.kernel_size[0] * mod.kernel_size[1]

    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        net.eval()(torch.zeros(1, 3, hw, hw))
    for h in hooks:
        h.remove()
    return total
