#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run in the build container (the one place /root/reference exists):

    python tests/golden/make_golden.py

The reference package cannot be imported here (cv2, chess, timm and its UNet submodule are absent -- SURVEY.md
section 8c), so tensor-level vectors are produced by the CPU oracle (oracle/*_ref.py: the reference's module trees on
the torch-CPU kernels the reference itself runs).  What comes straight from the reference tree is DATA only:
three segmentation masks (data/board_extraction/masks/*.png) with their annotated corners (coordinates.json).

Outputs (all small):
  ops.npz          seeded inputs / parameters / expected outputs for every op-shape class on the hot path
  unet.npz         full UNet(3,1) forward, both variants, 2 seeded inputs: 512 sampled logits + f64 checksums
  resnet18.npz     full ResNet-18 forward: logits + softmax for 128 seeded squares
  masks/*.png, masks/corners.json   reference fixtures for the mask -> quadrangle chain
"""
from __future__ import annotations

import json
import re
import shutil
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import prng, synth  # noqa: E402

REF = Path("/root/reference")


def t(seed, name, shape, std=1.0):
    return torch.from_numpy(prng.normal(seed, name, shape, 0.0, std))


def bn_params(seed, name, c):
    return (torch.from_numpy(prng.uniform(seed, name + "g", (c,), 0.5, 1.5)), t(seed, name + "b", (c,), 0.1),
            t(seed, name + "m", (c,), 0.1), torch.from_numpy(prng.uniform(seed, name + "v", (c,), 0.5, 1.5)))


def make_ops():
    out = {}
    s = 101
    # conv3x3 (no bias) + BN(eval) + ReLU  -- UNet DoubleConv half
    x, w = t(s, "c3x", (2, 8, 12, 10)), t(s, "c3w", (16, 8, 3, 3), 0.2)
    g, b, m, v = bn_params(s, "c3", 16)
    y = F.relu(F.batch_norm(F.conv2d(x, w, padding=1), m, v, g, b, training=False, eps=1e-5))
    out.update(conv3_x=x, conv3_w=w, conv3_g=g, conv3_b=b, conv3_m=m, conv3_v=v, conv3_y=y)
    # conv3x3 stride 2 + BN + ReLU (ResNet layerN.0.conv1) and 1x1 stride 2 + BN (downsample)
    x, w = t(s, "s2x", (2, 16, 8, 8)), t(s, "s2w", (32, 16, 3, 3), 0.1)
    g, b, m, v = bn_params(s, "s2", 32)
    y = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=1), m, v, g, b, training=False, eps=1e-5))
    wd = t(s, "dsw", (32, 16, 1, 1), 0.25)
    gd, bd, md, vd = bn_params(s, "ds", 32)
    yd = F.batch_norm(F.conv2d(x, wd, stride=2), md, vd, gd, bd, training=False, eps=1e-5)
    # second conv of the block with the residual add + ReLU
    w2 = t(s, "s2w2", (32, 32, 3, 3), 0.08)
    g2, b2, m2, v2 = bn_params(s, "s22", 32)
    yb = F.relu(F.batch_norm(F.conv2d(y, w2, padding=1), m2, v2, g2, b2, training=False, eps=1e-5) + yd)
    out.update(blk_x=x, blk_w1=w, blk_bn1=torch.stack([g, b, m, v]), blk_y1=y, blk_wd=wd, blk_bnd=torch.stack([gd, bd, md, vd]),
               blk_yd=yd, blk_w2=w2, blk_bn2=torch.stack([g2, b2, m2, v2]), blk_y=yb)
    # ConvTranspose2d k2 s2 + bias (UNet Up.up)
    x, w, bb = t(s, "ctx", (1, 16, 5, 6)), t(s, "ctw", (16, 16, 2, 2), 0.25), t(s, "ctb", (16,), 0.1)
    out.update(convT_x=x, convT_w=w, convT_b=bb, convT_y=F.conv_transpose2d(x, w, bb, stride=2))
    # bilinear x2 align_corners=True (UNet Up.up, bilinear variant)
    x = t(s, "upx", (1, 8, 5, 7))
    out.update(up_x=x, up_y=F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))
    # max pools
    x = t(s, "mpx", (1, 8, 8, 12))
    out.update(mp_x=x, mp2_y=F.max_pool2d(x, 2), mp3_y=F.max_pool2d(x, 3, stride=2, padding=1))
    # 7x7 s2 p3 stem + BN + ReLU on a u8/255 input
    x = torch.from_numpy(prng.bytes_u8(s, "stx", (2, 1, 64, 64))).float() / 255
    w = t(s, "stw", (64, 1, 7, 7), 0.2)
    g, b, m, v = bn_params(s, "st", 64)
    y = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=3), m, v, g, b, training=False, eps=1e-5))
    out.update(stem_x=x, stem_w=w, stem_bn=torch.stack([g, b, m, v]), stem_y=y)
    # 1x1 conv 64 -> 1 + bias (OutConv), sigmoid + threshold incl. the edge logits of SURVEY.md section 7
    x, w, bb = t(s, "ocx", (1, 64, 6, 6)), t(s, "ocw", (1, 64, 1, 1), 0.1), t(s, "ocb", (1,), 0.1)
    out.update(outc_x=x, outc_w=w, outc_b=bb, outc_y=F.conv2d(x, w, bb))
    edge = torch.tensor([0.0, 1e-7, -1e-7, 9e-8, 5e-8, 1e-3, -1e-3, 20.0, -20.0, 88.0, -88.0], dtype=torch.float32)
    out.update(edge_logits=edge, edge_sigmoid=torch.sigmoid(edge),
               edge_mask=torch.where(torch.sigmoid(edge) > 0.5, 255, 0).to(torch.uint8))
    # global avg pool + fc + softmax (ResNet head)
    x, w, bb = F.relu(t(s, "hdx", (4, 512, 2, 2))), t(s, "hdw", (13, 512), 0.05), t(s, "hdb", (13,), 0.1)
    logits = F.linear(F.adaptive_avg_pool2d(x, 1).flatten(1), w, bb)
    out.update(head_x=x, head_w=w, head_b=bb, head_logits=logits, head_probs=torch.softmax(logits, 1))
    np.savez_compressed(HERE / "ops.npz", **{k: v.numpy() for k, v in out.items()})


def make_models():
    idx = prng.bits64(7, "sample_idx", 512) % np.uint64(65536)
    idx = idx.astype(np.int64)
    pack = {"sample_idx": idx}
    for bilinear in (False, True):
        net = synth.make_unet(seed=1, bilinear=bilinear)
        x = synth.unet_input(seed=3, batch=2)
        with torch.no_grad():
            y = net(x).double()
        tag = "bilinear" if bilinear else "convT"
        flat = y.reshape(2, -1)
        pack[f"{tag}_samples"] = flat[:, idx].float().numpy()
        pack[f"{tag}_sum"] = flat.sum(1).numpy()
        pack[f"{tag}_abs_sum"] = flat.abs().sum(1).numpy()
        pack[f"{tag}_mask_count"] = (torch.sigmoid(y.float()) > 0.5).reshape(2, -1).sum(1).numpy()
    np.savez_compressed(HERE / "unet.npz", **pack)
    net = synth.make_resnet(seed=2)
    x = synth.squares_input(seed=4, n=128)
    with torch.no_grad():
        logits = net(x)
    np.savez_compressed(HERE / "resnet18.npz", logits=logits.numpy(), probs=torch.softmax(logits, 1).numpy())


def copy_reference_masks():
    names = ["00159703-7d3c-43fa-9e8f-733416ab9ac3", "00a2be00-e7af-453b-a219-48db6e434dd8",
             "013d7cfa-f904-476c-9704-5ddfdc8f823c"]
    corners = {}
    for line in open(REF / "data/board_extraction/coordinates.json"):
        d = json.loads(line)
        key = re.findall(r"([0-9a-f]{8}(?:-[0-9a-f]{4}){3}-[0-9a-f]{12})\.JPG", d["content"], flags=re.I)[-1]
        if key in names and d["annotation"]:
            corners[key] = d["annotation"][0]["points"][:4]          # fractions of the 256-px image
    for n in names:
        shutil.copy(REF / "data/board_extraction/masks" / f"{n}.png", HERE / "masks" / f"{n}.png")
    json.dump(corners, open(HERE / "masks" / "corners.json", "w"), indent=1)


def pack_all_reference_masks():
    """All 631 label masks of the reference (data/board_extraction/masks, 0/255, 256x256) bit-packed with their annotated
    corners (coordinates.json): the full pin of the mask -> quadrangle stage.  ~100 KB compressed."""
    from PIL import Image
    ann = {}
    for line in open(REF / "data/board_extraction/coordinates.json"):
        d = json.loads(line)
        key = re.findall(r"([0-9a-f]{8}(?:-[0-9a-f]{4}){3}-[0-9a-f]{12})\.JPG", d["content"], flags=re.I)[-1]
        if d["annotation"]:
            ann[key] = d["annotation"][0]["points"][:4]
    names = sorted(p.stem for p in (REF / "data/board_extraction/masks").glob("*.png") if p.stem in ann)
    masks = np.stack([np.array(Image.open(REF / "data/board_extraction/masks" / f"{n}.png").convert("L")) for n in names])
    assert set(np.unique(masks)) <= {0, 255}
    np.savez_compressed(HERE / "masks_all.npz", names=np.array(names), bits=np.packbits(masks > 0, axis=-1),
                        corners=np.array([ann[n] for n in names], dtype=np.float64))


if __name__ == "__main__":
    make_ops()
    make_models()
    if REF.exists():
        copy_reference_masks()
        pack_all_reference_masks()
    print("golden fixtures written to", HERE)
