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


WARP_SCRIPT = r"""
import sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import numpy as np, torch
from chessvision import classical, utils
from chessvision.hip_backend import HipEngine, board_homographies
from oracle import classical_ref as cref

assert classical.warp_mode() == "float"
rng = np.random.default_rng(6)
imgs = rng.integers(0, 256, (3, 384, 512, 3), dtype=np.uint8)
quads = [np.array([[400, 60], [90, 40], [60, 330], [430, 350]], np.float32),
         np.array([[500, 10], [20, 5], [-30, 370], [530, 400]], np.float32),
         np.array([[255, 0], [0, 0], [0, 255], [255, 255]], np.float32) * np.float32(384 / 256.0)]
eng = HipEngine(precision="f32", unet_chunk=2, resnet_chunk=128)
inv = board_homographies(np.stack(quads), (512, 512))
squares, boards = eng.extract_squares_u8(torch.from_numpy(imgs), inv)
for k in range(3):
    host = classical.flip_horizontal(classical.bgr_to_gray(utils.extract_perspective(imgs[k], quads[k], (512, 512))))
    assert np.array_equal(boards[k].cpu().numpy(), host), k                                # device == host (float reading)
    fixed = classical.flip_horizontal(classical.bgr_to_gray(classical.warp_perspective(
        imgs[k], classical.get_perspective_transform(quads[k], np.array(((0, 0), (512, 0), (512, 512), (0, 512)), np.float32)), (512, 512), mode="fixed")))
    assert not np.array_equal(host, fixed)
rows = slice(200, 216)                                                                      # the scalar oracle on a band of each board
for k in range(3):
    dest = np.array(((0, 0), (512, 0), (512, 512), (0, 512)), np.float64)
    m = cref.perspective_matrix(quads[k].reshape(4, 2), dest)
    band = cref.warp_perspective_float(imgs[k], m, (512, 512), rows=range(rows.start, rows.stop))[rows]
    want = cref.flip_lr(cref.bgr_to_gray(band))
    assert np.array_equal(boards[k].cpu().numpy()[rows], want), k                           # device == independent oracle
eng.close()
print("WARP_FLOAT_OK")
"""


def test_float_reading_of_the_warp_device_equals_host_equals_oracle():
    """CV_WARP=float in a child process (the library reads it once): the device kernel, the host form and the scalar oracle agree byte
    for byte on interior, out-of-frame and whole-image-fallback quadrangles, and differ from the fixed-point reading."""
    env = dict(os.environ)
    env["CV_WARP"] = "float"
    out = subprocess.run([sys.executable, "-c", WARP_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0 and "WARP_FLOAT_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


CHAIN_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth

import os
net = synth.make_resnet(2)
chunk, count = int(os.environ.get("CHAIN_CHUNK", "256")), int(os.environ.get("CHAIN_N", "600"))
eng = HipEngine(precision="f16r", resnet_chunk=chunk)
eng.load_resnet18(net.state_dict())
x = synth.squares_input(91, count)                            # default: two full chunks and a ragged one of 88
with torch.no_grad():
    ref = net(x)
outs = [eng.resnet18_forward(x.cuda()).cpu() for _ in range(3)]
assert all(torch.equal(outs[0], o) for o in outs[1:]), "not deterministic"
p_err = float((torch.softmax(ref, 1) - torch.softmax(outs[0], 1)).abs().max())
assert p_err <= 1e-3 and bool((ref.argmax(1) == outs[0].argmax(1)).all()), p_err
eng.check_numerics()
print("SHA", hashlib.sha256(outs[0].numpy().tobytes()).hexdigest(), "PERR", p_err)
"""


def test_f16r_layer1_chain_forms_give_the_bits_of_the_four_launch_schedule():
    """Round 5: the fp16 classifier runs layer1 (four convolutions) as ONE launch with the image resident in LDS.  Every form of it --
    two workgroups per CU with the last convolution through the ordinary staged epilogue (default), one workgroup per CU with the
    f32 trunk in registers (CV_CHAIN_WG=1) -- produces the SAME BITS as the four separate launches (CV_RESNET_CHAIN=0), run to run,
    on full and ragged chunks; the dedicated shortcut kernel (default: split-f16 products instead of the f32-input MFMA; all of its forms give the same bits)
    stays inside the fp16 bar (soft-max within 1e-3 of the oracle, every arg-max equal)."""
    shas = {}
    for name, knobs in (("default", {"CV_SHORTCUT_FAST": "0"}), ("four_launches", {"CV_RESNET_CHAIN": "0", "CV_SHORTCUT_FAST": "0"}),
                        ("one_wg", {"CV_CHAIN_WG": "1", "CV_SHORTCUT_FAST": "0"}),
                        ("fast_shortcut", {}), ("fast_shortcut_first_form", {"CV_SHORTCUT_LDS": "0"}),
                        ("fast_shortcut_unstaged", {"CV_SHORTCUT_STAGE": "0"})):
        env = dict(os.environ)
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", CHAIN_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "SHA" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        shas[name] = out.stdout.split("SHA", 1)[1].split()[0]
    assert shas["default"] == shas["four_launches"] == shas["one_wg"], shas
    assert shas["fast_shortcut"] == shas["fast_shortcut_first_form"] == shas["fast_shortcut_unstaged"], shas   # every form of the dedicated
                                                                                  # shortcut kernel: same operation order per output


PAIR_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth
net = synth.make_resnet(2)
h = hashlib.sha256()
worst = 0.0
for prec in ("f16x3", "f32", "f16", "f16r"):
    eng = HipEngine(precision=prec)
    eng.load_resnet18(net.state_dict())
    for n in (64, 1, 128, 640):
        sq = synth.squares_input(70 + n, n)
        for _ in range(3):                                    # eager, capture, replay
            got = eng.resnet18_forward(sq.cuda()).cpu()
        h.update(got.numpy().tobytes())
        if prec in ("f16x3", "f32"):
            with torch.no_grad():
                worst = max(worst, float((got - net(sq)).abs().max()))
    eng.check_numerics()
assert worst <= 1e-3, worst
print("SHA", h.hexdigest())
"""


def test_two_layers_in_one_launch_give_the_bits_of_two_launches():
    """Round 5: at single-board sizes a ResNet-18 stage's shortcut convolution and its first 3x3 convolution go out as ONE launch
    (conv_igemm_pair_kernel, Engine::PendingConv).  Each half is the launch it replaces: the logits are bit-identical to CV_PAIR=0 at
    1, 64, 128 and 640 squares, every precision (the fp16 classifier pairs its f32 shortcut with an f16 convolution), eager and replayed."""
    shas = {}
    for name, knobs in (("paired", {}), ("two_launches", {"CV_PAIR": "0"}), ("paired_always", {"CV_PAIR_MAX_BLOCKS": "1000000"})):
        env = dict(os.environ)
        env["CV_SHORTCUT_FAST"] = "0"                        # the fp16 classifier's generic shortcut launches: the f32-beside-f16 pair
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", PAIR_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "SHA" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        shas[name] = out.stdout.split("SHA", 1)[1].split()[0]
    assert shas["paired"] == shas["two_launches"] == shas["paired_always"], shas


UPSAMPLE_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
import torch.nn.functional as F
from chessvision.hip_backend import HipEngine
eng = HipEngine(precision="f16x3")
g = torch.Generator().manual_seed(11)
h = hashlib.sha256()
worst = 0.0
for shape in ((2, 32, 16, 16), (1, 512, 16, 16), (3, 64, 128, 128), (1, 8, 1, 1), (2, 16, 1, 7), (1, 24, 5, 3), (1, 256, 32, 32), (1, 16, 9, 1)):
    x = torch.randn(shape, generator=g) * 3
    y = eng.op_upsample_bilinear2x(x).cpu()
    worst = max(worst, float((y - F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)).abs().max()))
    h.update(y.numpy().tobytes())
eng.check_numerics()
assert worst <= 1e-5, worst
print("SHA", h.hexdigest())
"""


def test_upsample_2x2_block_kernel_gives_the_bits_of_the_one_output_kernel():
    """Round 5: the split-f16 bilinear up-sample computes 2 x 2 output pixels per lane from ONE set of four source pixels (0.44 -> 0.62 of
    the HBM rate).  Same operations per output in the same order: bit-identical to the one-output-per-lane kernel (CV_UPSAMPLE_2X2=0) on
    square, 1-pixel, odd and single-column maps, and within 1e-5 of torch."""
    shas = {}
    for name, knobs in (("block", {}), ("single", {"CV_UPSAMPLE_2X2": "0"})):
        env = dict(os.environ)
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", UPSAMPLE_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "SHA" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        shas[name] = out.stdout.split("SHA", 1)[1].split()[0]
    assert shas["block"] == shas["single"], shas


POS_SCRIPT = r"""
import hashlib, os, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth
net = synth.make_resnet(2)
chunk = int(os.environ.get("POS_CHUNK", "1024"))
counts = [int(v) for v in os.environ.get("POS_N", "256,300,700").split(",")]
precs = os.environ.get("POS_PRECS", "f16x3,f32,f16,f16r").split(",")
want_pos = os.environ.get("CV_POS", "1") != "0"
h = hashlib.sha256()
tagged = set()
for prec in precs:
    eng = HipEngine(precision=prec, resnet_chunk=chunk)
    eng.load_resnet18(net.state_dict())
    for n in counts:
        sq = synth.squares_input(300 + n, n)
        got = eng.resnet18_forward(sq.cuda()).cpu()
        again = eng.resnet18_forward(sq.cuda()).cpu()
        assert torch.equal(got, again), "not deterministic"
        h.update((got + 0.0).numpy().tobytes())               # + 0.0: a skipped all-zero stage may flip the sign of an exact zero
        for tap in ("layer2.0.act1", "layer3.0.act1", "layer3", "layer4.0.act1", "layer4"):
            h.update((torch.from_numpy(eng.activation("resnet18", tap)) + 0.0).numpy().tobytes())
        if n <= 1024 and prec in ("f16x3", "f32"):
            with torch.no_grad():
                err = float((got - net(sq)).abs().max())
            assert err <= 1e-3, (prec, n, err)
    entries = eng.profile("resnet18", synth.squares_input(5, counts[-1]))[3]
    tagged |= {e["name"] for e in entries if ",POS" in e["kernel"]}
    eng.check_numerics()
    eng.close()
assert bool(tagged) == want_pos, tagged
print("POSLAYERS", ",".join(sorted(tagged)) or "-")
print("SHA", h.hexdigest())
"""


@pytest.mark.parametrize("shape", ["forced_small", "forced_small_128px_tiles", "production_chunk", "production_packing_mid_batch"])
def test_position_major_rows_skip_the_zero_border_and_keep_every_bit(shape):
    """Round 6: the generic kernel's 3x3 launches on the 2x2 / 4x4 / 8x8 maps of ResNet-18 order their rows [output position][image] and
    walk only the K stages whose tap reads a real pixel (ConvParams::ptab): 5 of 9 taps of every position of a 2x2 map gather the zero
    border.  A skipped stage would have added exact zeros, so logits and the layer2-4 activations are BIT-IDENTICAL to CV_POS=0 -- in all
    four precisions, on full and ragged image tiles (256 / 300 / 700 squares with the 256-pixel tiles forced), and at the production
    chunk of 16384 squares where the 256 x 256 tile takes layer3 / layer4 by itself."""
    base = {"forced_small": {"POS_CHUNK": "1024", "POS_N": "256,300,700", "CV_CONV_PT": "256", "CV_SPLITK": "0"},
            # the 4-wave 128 x 128 tile (what mid-size batches run: two workgroups per CU) takes position-major launches too
            "forced_small_128px_tiles": {"POS_CHUNK": "1024", "POS_N": "128,200,700", "CV_CONV_PT": "128", "CV_SPLITK": "0", "CV_CT256": "0"},
            "production_chunk": {"POS_CHUNK": "16384", "POS_N": "16384", "POS_PRECS": "f16x3,f16r"},
            # 4096 / 5000 squares on weights packed for the 16384-square chunk (configs[2]'s batch; a 64-board job of process_images):
            # layer3's 256-row launch would be one round deep, so its unequal positions take the 128-row packing (CV_POS_SMALL_CT)
            "production_packing_mid_batch": {"POS_CHUNK": "16384", "POS_N": "4096,5000"}}[shape]
    shas, layers = {}, {}
    for name, knobs in (("pos", {}), ("image_major", {"CV_POS": "0"})):
        env = dict(os.environ)
        env.update(base)
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", POS_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=1800)
        assert out.returncode == 0 and "SHA" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        shas[name] = out.stdout.split("SHA", 1)[1].split()[0]
        layers[name] = out.stdout.split("POSLAYERS", 1)[1].split()[0]
    assert shas["pos"] == shas["image_major"], shas
    for must in ("layer3.1.conv1", "layer4.0.conv1", "layer4.1.conv2"):
        assert must in layers["pos"], layers


BIAS_SCRIPT = r"""
import hashlib, json, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth
net = synth.make_resnet(5)                                     # the worst network of the seed search (tests/dev/f16r_seed_search.py)
sq = synth.squares_input(1005, 4096)
with torch.no_grad():
    ref = net(sq)
p_ref = torch.softmax(ref, 1)
out = {}
for prec, chunk in (("f16r", 4096), ("f16r", 96), ("f16", 4096), ("f16x3", 4096)):
    eng = HipEngine(precision=prec, resnet_chunk=chunk)
    eng.load_resnet18(net.state_dict())
    got = eng.resnet18_forward(sq).cpu()
    eng.check_numerics(); eng.close()
    out[f"{prec}/{chunk}"] = {"prob_err": float((torch.softmax(got, 1) - p_ref).abs().max()), "logit_rms": float((got - ref).pow(2).mean().sqrt()),
                             "sha": hashlib.sha256(got.numpy().tobytes()).hexdigest()}
print("BIAS", json.dumps(out))
"""


def test_rounding_bias_correction_cuts_the_fp16_classifier_error_and_leaves_f16x3_alone():
    """Round 6: the rounding errors of a layer's f16 weights are the same at every pixel of every image, so through post-ReLU inputs
    they add a constant offset per output channel -- 70 % of the fp16 classifier's logit-error variance (CPU emulation, DESIGN.md section
    2).  At load time the offset is measured on the calibration batch and folded into the f32 epilogue shift.  Against CV_BIAS_CORR=0 on
    the worst network of the seed search: the rms logit error drops by more than a quarter and the worst soft-max error falls from
    outside the 1e-3 bar to inside it; the plain f16 engine gains too; f16x3 (exact weights: nothing to correct) keeps its bits; and
    the correction does not depend on how the load-time passes are chunked (resnet_chunk 96 -> the 256-square pass runs in three pieces)."""
    import json
    res = {}
    for name, knobs in (("on", {}), ("off", {"CV_BIAS_CORR": "0"})):
        env = dict(os.environ)
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", BIAS_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0 and "BIAS" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        res[name] = json.loads(out.stdout.split("BIAS", 1)[1].strip().splitlines()[0])
    print({k: {kk: (vv["prob_err"], vv["logit_rms"]) for kk, vv in v.items()} for k, v in res.items()})
    on, off = res["on"], res["off"]
    assert on["f16r/4096"]["logit_rms"] <= 0.75 * off["f16r/4096"]["logit_rms"], (on, off)
    assert off["f16r/4096"]["prob_err"] > 1e-3 >= on["f16r/4096"]["prob_err"], (on, off)
    assert on["f16/4096"]["logit_rms"] <= 0.9 * off["f16/4096"]["logit_rms"], (on, off)
    assert on["f16x3/4096"]["sha"] == off["f16x3/4096"]["sha"]
    # forwards in chunks of 96 squares instead of one launch: the same constants (bit-identical logits need the same launch shapes,
    # so compare the error figures, which must agree to the last few per cent)
    assert abs(on["f16r/96"]["logit_rms"] - on["f16r/4096"]["logit_rms"]) <= 0.05 * on["f16r/4096"]["logit_rms"], on


SPLIT_SHORTCUT_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth
net = synth.make_resnet(2)
eng = HipEngine(precision="f16x3", resnet_chunk=2048)
eng.load_resnet18(net.state_dict())
h = hashlib.sha256()
kernels = set()
for n in (1024, 1500, 2048, 64):
    sq = synth.squares_input(400 + n, n)
    got = eng.resnet18_forward(sq.cuda()).cpu()
    with torch.no_grad():
        assert float((got - net(sq)).abs().max()) <= 1e-3
    h.update(got.numpy().tobytes())
    for tap in ("layer2.0.downsample", "layer3.0.downsample"):
        h.update(eng.activation("resnet18", tap).tobytes())
    kernels |= {e["kernel"] for e in eng.profile("resnet18", sq)[3] if "downsample" in e["name"]}
eng.check_numerics()
print("KERNELS", "|".join(sorted(kernels)))
print("SHA", h.hexdigest())
"""


def test_split_shortcut_kernel_gives_the_bits_of_the_generic_launch():
    """Round 6 (VERDICT r05 item 3 i): the dedicated shortcut kernel of round 5 (weights resident in LDS, persistent waves, full-line
    stores) serves the HEADLINE engine too, in a form that reads and writes split-f16 tensors.  It forms the products of the generic
    kernel in the generic kernel's order, so logits and the three shortcut tensors are bit-identical to CV_SHORTCUT_FAST=0 -- at 1024,
    1500 (ragged), 2048 squares, and at 64 where both runs take the paired generic launch anyway."""
    shas, kernels = {}, {}
    for name, knobs in (("dedicated", {}), ("generic", {"CV_SHORTCUT_FAST": "0"})):
        env = dict(os.environ)
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", SPLIT_SHORTCUT_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "SHA" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        shas[name] = out.stdout.split("SHA", 1)[1].split()[0]
        kernels[name] = out.stdout.split("KERNELS", 1)[1].split()[0]
    assert "shortcut1x1s2_kernel<64,split>" in kernels["dedicated"] and "shortcut1x1s2" not in kernels["generic"], kernels
    assert shas["dedicated"] == shas["generic"], shas


CONVT_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/chessvision-3lc_amd")
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth
net = synth.make_unet(1)
eng = HipEngine(precision="f16x3", unet_chunk=12)
eng.load_unet(net.state_dict())
h = hashlib.sha256()
kernels = set()
for b in (8, 11, 3):
    x = synth.unet_input(500 + b, b)
    got = eng.unet_forward(x.cuda()).cpu()
    with torch.no_grad():
        assert float((got - net(x)).abs().max()) <= 1e-3
    h.update(got.numpy().tobytes())
    for tap in ("up3.up", "up4.up"):
        h.update(eng.activation("unet", tap).tobytes())
    kernels |= {e["kernel"] for e in eng.profile("unet", x)[3] if e["name"] in ("up3.up", "up4.up")}
eng.check_numerics()
print("KERNELS", "|".join(sorted(kernels)))
print("SHA", h.hexdigest())
"""


def test_lds_resident_transposed_conv_gives_the_bits_of_the_generic_launch():
    """Round 6: UNet up3.up / up4.up (k2 s2 transposed convolutions with K = 256 / 128: the whole weight block fits in LDS) run on the
    persistent LDS-resident-weights kernel in its CONVT form from 8 boards up -- dense input pixels, pixel-shuffle stores into the
    channel slice of the concatenated tensor.  Same products in the same order as the generic 1-tap GEMM: logits and both up-sampled
    tensors are bit-identical to CV_CONVT_FAST=0 at 8 and 11 boards (and at 3, where both runs take the generic launch)."""
    shas, kernels = {}, {}
    for name, knobs in (("lds", {"CV_CONVT_FAST_256": "1"}), ("generic", {"CV_CONVT_FAST": "0"})):     # (up3.up takes the kernel on request only: slower there)
        env = dict(os.environ)
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", CONVT_SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "SHA" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        shas[name] = out.stdout.split("SHA", 1)[1].split()[0]
        kernels[name] = out.stdout.split("KERNELS", 1)[1].split()[0]
    assert "convt2x2_lds_kernel<128,split>" in kernels["lds"] and "convt2x2_lds_kernel<256,split>" in kernels["lds"], kernels
    assert "convt2x2" not in kernels["generic"], kernels
    assert shas["lds"] == shas["generic"], shas
