"""GPU parity of the two whole models against the CPU oracle (oracle/unet_ref.py, oracle/resnet_ref.py),
through the same seam the reference uses: ``model(tensor) -> tensor`` (core.py:220,241).

Bars (north_star: "within 1e-3 fp32"):
  * f32 engine and f16x3 engine (split-f16, 3 MFMAs per k-block): logits max-abs <= 1e-3 (measured ~1e-4 / ~1e-5),
    masks identical except at |logit| < 1e-4.
  * f16 engine (f16 storage, f32 accumulate) does NOT meet 1e-3 on logits and is not claimed to: every layer
    rounds activations and weights to 11 bits, and through 23 (UNet) / 20 (ResNet) conv layers the measured
    logit error is ~2.5e-3 * max|logit| on random-init weights.  Asserted here: the f16 rounding floor
    (<= 5e-3 * max|logit|), mask IoU >= 0.995, probability error <= 2e-2.  The numbers land in
    gpurun_out/parity_report.jsonl and DESIGN.md quotes them.
"""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu

OUT = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"


def _record(name, payload):
    try:
        OUT.mkdir(exist_ok=True)
        with open(OUT / "parity_report.jsonl", "a") as f:
            f.write(json.dumps({"test": name, **payload}) + "\n")
    except OSError:
        pass


def _hooks(net, names):
    got = {}
    hs = []
    mods = dict(net.named_modules())
    for n in names:
        hs.append(mods[n].register_forward_hook(lambda m, i, o, n=n: got.__setitem__(n, o.detach().clone())))
    return got, hs


UNET_TAPS = ["inc.double_conv.5", "down1.maxpool_conv.0", "down1.maxpool_conv.1.double_conv.5",
             "down2.maxpool_conv.1.double_conv.5", "down3.maxpool_conv.1.double_conv.5",
             "down4.maxpool_conv.1.double_conv.5", "up1.up", "up1.conv.double_conv.5", "up2.conv.double_conv.5",
             "up3.conv.double_conv.5", "up4.up"]   # up4.conv output is consumed inside the fused OutConv epilogue; inc.double_conv.2
                                                   # exists only inside inc.double_conv.3's kernel (f16x3 engine)


@pytest.mark.parametrize("bilinear", [False, True], ids=["convT", "bilinear"])
@pytest.mark.parametrize("prec", ["f32", "f16", "f16x3"])
def test_unet_forward_matches_oracle(prec, bilinear):
    from chessvision.hip_backend import HipEngine

    net = synth.make_unet(seed=1, bilinear=bilinear)
    x = synth.unet_input(seed=3, batch=3)                      # 3 images with chunk 2 -> exercises the chunk loop
    got_ref, hs = _hooks(net, UNET_TAPS)
    with torch.no_grad():
        ref = net(x)
    for h in hs:
        h.remove()
    eng = HipEngine(precision=prec, unet_chunk=2)
    eng.load_unet(net.state_dict())
    out = eng.unet_forward(x).cpu()
    assert out.shape == (3, 1, 256, 256)
    # per-layer diagnostics on the last chunk (image index 2)
    layer_err = {}
    for name in UNET_TAPS:
        a = torch.from_numpy(eng.activation("unet", name))
        r = got_ref[name][2:3]
        layer_err[name] = [float((a - r).abs().max()), float(r.abs().max())]
    err = float((out - ref).abs().max())
    scale = float(ref.abs().max())
    m_ref = torch.sigmoid(ref) > 0.5
    m_got = torch.sigmoid(out) > 0.5
    iou = float((m_ref & m_got).sum()) / max(1.0, float((m_ref | m_got).sum()))
    p_err = float((torch.sigmoid(out) - torch.sigmoid(ref)).abs().max())
    _record("unet", {"prec": prec, "bilinear": bilinear, "logit_max_abs_err": err, "logit_max": scale, "mask_iou": iou,
                     "prob_max_abs_err": p_err, "layers": layer_err})
    eng.close()
    if prec in ("f32", "f16x3"):
        assert err <= 1e-3, (err, layer_err)
        assert iou >= 0.9999
    else:
        assert err <= 5e-3 * max(1.0, scale), (err, scale, layer_err)
        assert iou >= 0.995, iou
        assert p_err <= 2e-2, p_err


RESNET_TAPS = ["act1", "maxpool", "layer1.0", "layer1", "layer2.0", "layer2", "layer3", "layer4"]


@pytest.mark.parametrize("prec", ["f32", "f16", "f16x3", "f16r"])
def test_resnet18_forward_matches_oracle(prec):
    from chessvision.hip_backend import HipEngine

    net = synth.make_resnet(seed=2)
    x = synth.squares_input(seed=4, n=200)                     # chunk 128 -> two chunks, ragged tail of 72
    names = {"act1": "act1", "maxpool": "maxpool", "layer1.0": "layer1.0", "layer1": "layer1", "layer2.0": "layer2.0",
             "layer2": "layer2", "layer3": "layer3", "layer4": "layer4"}
    got_ref, hs = _hooks(net, list(names.values()))
    with torch.no_grad():
        ref = net(x)
    for h in hs:
        h.remove()
    eng = HipEngine(precision=prec, resnet_chunk=128)
    eng.load_resnet18(net.state_dict())
    out = eng.resnet18_forward(x).cpu()
    assert out.shape == (200, 13)
    layer_err = {}
    for name in RESNET_TAPS:
        if name == "act1" and prec != "f32":
            continue                    # f16 / f16x3 engines fuse stem + max-pool: the 32x32 stem output never exists
        if name == "layer1.0" and prec == "f16r":
            continue                    # f16r runs layer1 as one chained launch: block 0's output stays in registers / LDS
        a = torch.from_numpy(eng.activation("resnet18", name))
        r = got_ref[name][128:200]
        layer_err[name] = [float((a - r).abs().max()), float(r.abs().max())]
    err = float((out - ref).abs().max())
    scale = float(ref.abs().max())
    p_ref, p_got = torch.softmax(ref, 1), torch.softmax(out, 1)
    p_err = float((p_ref - p_got).abs().max())
    agree = float((p_ref.argmax(1) == p_got.argmax(1)).float().mean())
    probs_dev = eng.softmax13(eng.resnet18_forward(x)).cpu()
    _record("resnet18", {"prec": prec, "logit_max_abs_err": err, "logit_max": scale, "prob_max_abs_err": p_err,
                         "argmax_agreement": agree, "layers": layer_err})
    eng.close()
    assert float((probs_dev - p_got).abs().max()) <= 1e-6
    if prec in ("f32", "f16x3"):
        assert err <= 1e-3, (err, layer_err)
        assert agree == 1.0
    else:
        assert err <= 5e-3 * max(1.0, scale), (err, scale, layer_err)
        assert p_err <= 1e-3, p_err          # fp16 storage / fp32 accumulate, probabilities within 1e-3
        assert agree >= 0.99


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_full_chunks_and_small_tail_launches(prec):
    """Chunks of 32 boards / 8192 squares: the full chunk runs the 256-row tiles and the halo kernel, the
    one-board / 64-square tail runs the same layers from the 128-row weight packing.  Both must match the oracle."""
    from chessvision.hip_backend import HipEngine

    unet, resnet = synth.make_unet(seed=1), synth.make_resnet(seed=2)
    x = synth.unet_input(seed=11, batch=33)
    sq = synth.squares_input(seed=12, n=8192 + 64)
    eng = HipEngine(precision=prec, unet_chunk=32, resnet_chunk=8192)
    eng.load_unet(unet.state_dict())
    eng.load_resnet18(resnet.state_dict())
    out_u = eng.unet_forward(x).cpu()
    out_r = eng.resnet18_forward(sq).cpu()
    one = eng.unet_forward(x[5:6]).cpu()                       # the same board alone: small-launch path end to end
    few = eng.resnet18_forward(sq[100:164]).cpu()
    eng.close()
    pick_u = [0, 5, 31, 32]
    pick_r = list(range(0, 64)) + list(range(8100, 8256))
    with torch.no_grad():
        ref_u = unet(x[pick_u])
        ref_r = resnet(sq[pick_r])
    err_u = float((out_u[pick_u] - ref_u).abs().max())
    err_r = float((out_r[pick_r] - ref_r).abs().max())
    err_one = float((one - ref_u[1:2]).abs().max())
    with torch.no_grad():
        err_few = float((few - resnet(sq[100:164])).abs().max())
    _record("chunks_and_tails", {"prec": prec, "unet": err_u, "resnet18": err_r, "unet_single": err_one, "resnet_64": err_few})
    assert max(err_u, err_r, err_one, err_few) <= 1e-3, (err_u, err_r, err_one, err_few)


def test_forward_passes_are_deterministic():
    """No atomics, fixed tile schedules: the same input gives bit-identical outputs run after run, and a board gives the
    same logits whether it travels alone or inside a batch of the same launch class (chunk boundaries do not leak)."""
    from chessvision.hip_backend import HipEngine

    eng = HipEngine(precision="f16x3", unet_chunk=4, resnet_chunk=256)
    eng.load_unet(synth.make_unet(seed=1).state_dict())
    eng.load_resnet18(synth.make_resnet(seed=2).state_dict())
    x = synth.unet_input(seed=31, batch=6).cuda()
    sq = synth.squares_input(seed=32, n=300).cuda()
    a, b = eng.unet_forward(x), eng.unet_forward(x)
    c, d = eng.resnet18_forward(sq), eng.resnet18_forward(sq)
    assert torch.equal(a, b) and torch.equal(c, d)
    assert torch.equal(eng.unet_forward(x[:4]), a[:4])          # first chunk of the batch == the same four boards alone
    assert torch.equal(eng.resnet18_forward(sq[:256]), c[:256])
    eng.close()


@pytest.mark.parametrize("prec", ["f16x3", "f16r", "f16", "f32"])
def test_every_mfma_kernel_repeats_its_bits_over_many_runs(prec):
    """the run-to-run differences round 5 chased came from inline asm next to independent MFMAs and were fixed in ONE
    kernel.  Every MFMA kernel of every engine is repeated here: UNet at 1 (split-K launches, 8 x 16 patches), 3 and 8 boards (fused
    first layer, halo tiles, transposed convolutions, fused pool / head), ResNet-18 at 64 (paired launches, split-K), 700 (ragged) and
    4096 squares (position-major tiles, packed 8 x 8 images, the chained layer1 and dedicated shortcuts under f16r) -- eight forwards
    each, alternating between shapes so that hipGraph replays and eager launches interleave; all bit-identical to the first."""
    from chessvision.hip_backend import HipEngine

    eng = HipEngine(precision=prec, unet_chunk=8, resnet_chunk=4096)
    eng.load_unet(synth.make_unet(seed=1).state_dict())
    eng.load_resnet18(synth.make_resnet(seed=2).state_dict())
    xs = {b: synth.unet_input(seed=90 + b, batch=b).cuda() for b in (1, 3, 8)}
    sqs = {n: synth.squares_input(seed=95 + n, n=n).cuda() for n in (64, 700, 4096)}
    first = {}
    for rep in range(8):
        for b, x in xs.items():
            out = eng.unet_forward(x, check=False)
            assert torch.equal(first.setdefault(("u", b), out.clone()), out), (prec, "unet", b, rep)
        for n, sq in sqs.items():
            out = eng.resnet18_forward(sq, check=False)
            assert torch.equal(first.setdefault(("r", n), out.clone()), out), (prec, "resnet18", n, rep)
    eng.check_numerics()
    eng.close()


def test_empty_batches_are_no_ops():
    """B = 0 is legal at the seam (an image without a detected board yields no squares): shapes only, no launches."""
    from chessvision.hip_backend import HipEngine

    eng = HipEngine(precision="f16x3")
    eng.load_unet(synth.make_unet(seed=1).state_dict())
    eng.load_resnet18(synth.make_resnet(seed=2).state_dict())
    assert tuple(eng.unet_forward(torch.empty(0, 3, 256, 256)).shape) == (0, 1, 256, 256)
    assert tuple(eng.resnet18_forward(torch.empty(0, 1, 64, 64)).shape) == (0, 13)
    logits, mask = eng.unet_forward_u8(torch.empty(0, 256, 256, 3, dtype=torch.uint8))
    assert tuple(logits.shape) == (0, 1, 256, 256) and tuple(mask.shape) == (0, 256, 256)
    assert tuple(eng.resnet18_forward_u8(torch.empty(0, 64, 64, dtype=torch.uint8)).shape) == (0, 13)
    assert tuple(eng.softmax13(torch.empty(0, 13)).shape) == (0, 13)
    eng.close()


def test_u8_entry_points_match_float_path():
    """cv_unet_forward_u8 / cv_resnet18_forward_u8 == float path fed with u8/255 (core.py:215,237)."""
    from chessvision.hip_backend import HipEngine
    from oracle import prng

    eng = HipEngine(precision="f32", unet_chunk=2, resnet_chunk=128)
    unet, rn = synth.make_unet(1), synth.make_resnet(2)
    eng.load_unet(unet.state_dict())
    eng.load_resnet18(rn.state_dict())
    img = torch.from_numpy(prng.bytes_u8(5, "img", (2, 256, 256, 3)))
    logits_u8, mask = eng.unet_forward_u8(img, threshold=0.5)
    logits_f = eng.unet_forward((img.float() / 255).permute(0, 3, 1, 2).contiguous())
    assert torch.equal(logits_u8.cpu(), logits_f.cpu())
    expect = torch.where(torch.sigmoid(logits_f.cpu())[:, 0] > 0.5, 255, 0).to(torch.uint8)
    assert float((mask.cpu() != expect).float().mean()) <= 1e-5
    sq = torch.from_numpy(prng.bytes_u8(6, "sq", (96, 64, 64)))
    probs = eng.resnet18_forward_u8(sq)
    f = sq.float().unsqueeze(1)
    f /= 255.0
    ref = torch.softmax(eng.resnet18_forward(f), 1)
    assert float((probs - ref).abs().max()) <= 1e-6
    eng.close()


def test_errors_are_exceptions_not_aborts():
    from chessvision.hip_backend import HipBackendError, HipEngine

    eng = HipEngine(precision="f16")
    with pytest.raises(HipBackendError, match="not loaded"):
        eng.unet_forward(torch.zeros(1, 3, 256, 256))
    sd = synth.make_unet(1).state_dict()
    sd.pop("up3.conv.double_conv.0.weight")
    with pytest.raises(HipBackendError, match="up3.conv.double_conv.0.weight"):
        eng.load_unet(sd)
    with pytest.raises(HipBackendError):
        eng.unet_forward(torch.zeros(1, 3, 128, 128))
    eng.close()


def test_baseline_config1_unet_fp32_batch32_mask_iou():
    """BASELINE.json configs[1]: UNet 256x256 fp32 forward, batch = 32, against the CPU oracle: logits within 1e-3, mask IoU."""
    from chessvision.hip_backend import HipEngine

    net = synth.make_unet(seed=1)
    x = synth.unet_input(seed=41, batch=32)
    with torch.no_grad():
        ref = net.to(memory_format=torch.channels_last)(x)
    eng = HipEngine(precision="f32", unet_chunk=32)
    eng.load_unet(net.state_dict())
    out = eng.unet_forward(x).cpu()
    eng.close()
    err = float((out - ref).abs().max())
    m_ref, m_got = ref > 0, out > 0
    iou = float((m_ref & m_got).sum()) / max(1.0, float((m_ref | m_got).sum()))
    _record("config1_unet_f32_b32", {"logit_max_abs_err": err, "mask_iou": iou})
    assert err <= 1e-3 and iou >= 0.9999, (err, iou)


def test_baseline_config2_resnet18_fp16_batch4096():
    """BASELINE.json configs[2]: ResNet-18 classifier, 64 x 64 squares = batch 4096, fp16 products with fp32 accumulate, against
    the fp32 CPU oracle.  The bar: soft-max probabilities within 1e-3, arg-max agreement.
      * "f16r" -- the classifier's fp16 mode: ONE f16 MFMA product per MAC, the residual trunk and the 1x1 shortcut convolutions
        in f32 -- must meet the bar, on the He-normal weights AND on the stressed ones (BatchNorm statistics over eight decades);
      * plain "f16" rounds the trunk to f16 after every block: 1.4e-3 here, reported, not inside the bar;
      * "f16x3" meets it with three orders of magnitude to spare."""
    from chessvision.hip_backend import HipEngine

    sq = synth.squares_input(seed=42, n=4096)
    res = {}
    for wname in ("he_normal", "stress"):
        net = synth.make_resnet(seed=2)
        if wname == "stress":
            synth.load(net, synth.stress_resnet_state_dict(2))
        with torch.no_grad():
            ref = net(sq)
        p_ref = torch.softmax(ref, 1)
        for prec in ("f16r", "f16", "f16x3"):
            eng = HipEngine(precision=prec, resnet_chunk=4096)
            eng.load_resnet18(net.state_dict())
            out = eng.resnet18_forward(sq).cpu()
            eng.close()
            p = torch.softmax(out, 1)
            res[f"{wname}/{prec}"] = (float((out - ref).abs().max()), float((p - p_ref).abs().max()),
                                     float((p.argmax(1) == p_ref.argmax(1)).float().mean()))
    _record("config2_resnet18_b4096", {k: {"logit_err": v[0], "prob_err": v[1], "argmax_agreement": v[2]} for k, v in res.items()})
    for wname in ("he_normal", "stress"):
        assert res[f"{wname}/f16r"][1] <= 1e-3 and res[f"{wname}/f16r"][2] >= 0.9995, res      # THE bar of configs[2]
        assert res[f"{wname}/f16"][1] <= 3e-3 and res[f"{wname}/f16"][2] >= 0.999, res          # rounding floor of the plain f16 engine
        assert res[f"{wname}/f16x3"][0] <= 1e-3 and res[f"{wname}/f16x3"][2] == 1.0, res


def fp16_classifier_worst_case(seeds=range(8), squares=4096, precision="f16r"):
    """Worst soft-max error of the fp16 classifier against the fp32 oracle over weight seeds x {He-normal, stressed} (shared with
    bench.py, which reports the same search in its line).  Returns (worst error, its case, per-case rows)."""
    from chessvision.hip_backend import HipEngine

    rows = []
    for seed in seeds:
        sq = synth.squares_input(seed=1000 + seed, n=squares)
        for wname in ("he_normal", "stress"):
            net = synth.make_resnet(seed=seed)
            if wname == "stress":
                synth.load(net, synth.stress_resnet_state_dict(seed))
            with torch.no_grad():
                ref = net(sq)
            p_ref = torch.softmax(ref, 1)
            eng = HipEngine(precision=precision, resnet_chunk=squares)
            eng.load_resnet18(net.state_dict())
            out = eng.resnet18_forward(sq).cpu()
            eng.close()
            p = torch.softmax(out, 1)
            flipped = p.argmax(1) != p_ref.argmax(1)
            top2 = p_ref.topk(2, dim=1).values
            rows.append({"seed": seed, "weights": wname, "prob_err": float((p - p_ref).abs().max()), "logit_err": float((out - ref).abs().max()),
                         "argmax_flips": int(flipped.sum()),
                         # an arg-max may only change where the oracle itself holds a tie inside the tolerance
                         "worst_flip_margin": float((top2[:, 0] - top2[:, 1])[flipped].max()) if bool(flipped.any()) else 0.0})
    worst = max(rows, key=lambda r: r["prob_err"])
    return worst["prob_err"], worst, rows


def test_fp16_classifier_bar_holds_on_the_worst_of_eight_seeds_and_both_weight_kinds():
    """VERDICT r05 'weak' 1: configs[2]'s bar was met at 8.6e-4 of 1e-3 on ONE batch of one weight set.  Here the fp16 classifier
    ("f16r") runs 4096 squares on 8 weight seeds x {He-normal, stressed BatchNorm statistics} = 16 networks and the WORST soft-max
    error must stay within 1e-3 of the fp32 oracle; an arg-max may differ only where the oracle's two best classes are themselves
    within 2e-3 (random-init networks do produce such ties).  Before round 6's rounding-bias correction of the f16 weights
    (ConvLayer::want_round_err) the worst of this grid was 1.27e-3 (seed 5); with it 8.0e-4."""
    worst, case, rows = fp16_classifier_worst_case()
    _record("fp16_classifier_seed_search", {"worst": case, "rows": rows})
    assert worst <= 1e-3, case
    assert all(r["worst_flip_margin"] <= 2e-3 for r in rows), [r for r in rows if r["worst_flip_margin"] > 2e-3]


def test_workspace_growth_is_transactional():
    """A batch whose workspace cannot be allocated (here: a chunk whose tensors would pass 4 GiB) must fail with an error and
    leave the engine exactly as it was -- the next small batch runs on the old buffers and gives the same logits as before."""
    from chessvision.hip_backend import HipBackendError, HipEngine

    net = synth.make_unet(seed=1)
    eng = HipEngine(precision="f16x3", unet_chunk=160)
    eng.load_unet(net.state_dict())
    x1 = synth.unet_input(seed=5, batch=1)
    before = eng.unet_forward(x1).cpu()
    ws = eng.workspace_bytes()
    big = torch.zeros((160, 3, 256, 256), device="cuda")
    with pytest.raises(HipBackendError, match="4 GiB"):
        eng.unet_forward(big)
    del big
    assert eng.workspace_bytes() == ws
    after = eng.unet_forward(x1).cpu()
    assert torch.equal(before, after)
    eng.close()


def test_graph_replay_of_small_forwards_is_transparent():
    """Round 4: a forward that fits one small chunk (UNet <= 8 images, ResNet-18 <= 512 squares) is captured into a hipGraph on its
    second call with the same pointers and replayed afterwards.  Eager call, capture call and replays give identical bits; a
    different output tensor, a workspace growth (which moves every activation buffer) and new input VALUES in the same tensor are
    all honoured (the captured launches read the caller's memory at replay time; stale captures are dropped by epoch)."""
    from chessvision.hip_backend import HipEngine

    unet, resnet = synth.make_unet(seed=1), synth.make_resnet(seed=2)
    eng = HipEngine(precision="f16x3")
    eng.load_unet(unet.state_dict())
    eng.load_resnet18(resnet.state_dict())
    x = synth.unet_input(seed=41, batch=1).cuda()
    sq = synth.squares_input(seed=42, n=64).cuda()
    with torch.no_grad():
        ref_u, ref_r = unet(x.cpu()), resnet(sq.cpu())
    lib, h = eng._lib, eng._h
    out_u = torch.empty((1, 1, 256, 256), device="cuda")
    out_r = torch.empty((64, 13), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    runs_u, runs_r = [], []
    for _ in range(4):                                       # eager, capture, replay, replay -- same pointers every time
        assert lib.cv_unet_forward(h, x.data_ptr(), 1, out_u.data_ptr(), stream) == 0
        assert lib.cv_resnet18_forward(h, sq.data_ptr(), 64, out_r.data_ptr(), stream) == 0
        torch.cuda.synchronize()
        runs_u.append(out_u.cpu().clone()); runs_r.append(out_r.cpu().clone())
    assert all(torch.equal(runs_u[0], r) for r in runs_u[1:]) and all(torch.equal(runs_r[0], r) for r in runs_r[1:])
    assert (runs_u[0] - ref_u).abs().max() <= 1e-3 and (runs_r[0] - ref_r).abs().max() <= 1e-3
    # new values in the same input tensor: the replay must read them
    x2, sq2 = synth.unet_input(seed=43, batch=1), synth.squares_input(seed=44, n=64)
    x.copy_(x2); sq.copy_(sq2)
    assert lib.cv_unet_forward(h, x.data_ptr(), 1, out_u.data_ptr(), stream) == 0
    assert lib.cv_resnet18_forward(h, sq.data_ptr(), 64, out_r.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    with torch.no_grad():
        assert (out_u.cpu() - unet(x2)).abs().max() <= 1e-3 and (out_r.cpu() - resnet(sq2)).abs().max() <= 1e-3
    after_new_values = out_u.cpu().clone()
    # grow the workspace (batch 3 re-allocates every UNet tensor), then the captured shape again: stale graph dropped, same bits
    eng.unet_forward(synth.unet_input(seed=45, batch=3).cuda())
    eng.resnet18_forward(synth.squares_input(seed=46, n=700).cuda())
    for _ in range(3):
        assert lib.cv_unet_forward(h, x.data_ptr(), 1, out_u.data_ptr(), stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(out_u.cpu(), after_new_values)
    eng.check_numerics()
    eng.close()


def test_activation_taps_report_the_batch_of_the_last_forward_also_after_a_graph_replay():
    """a hipGraph replay skips the host side of the chunk functions, where `last_n` used to be set -- after an eager forward
    of 2 images followed by a replayed forward of 1, the taps reported N = 2 over buffers that held 1 image."""
    from chessvision.hip_backend import HipEngine

    eng = HipEngine(precision="f16x3")
    eng.load_unet(synth.make_unet(seed=1).state_dict())
    eng.load_resnet18(synth.make_resnet(seed=2).state_dict())
    one, two = synth.unet_input(seed=51, batch=1).cuda(), synth.unet_input(seed=52, batch=2).cuda()
    s64, s128 = synth.squares_input(seed=53, n=64).cuda(), synth.squares_input(seed=54, n=128).cuda()
    for _ in range(3):                                       # eager, capture, replay of the small shapes
        eng.unet_forward(one); eng.resnet18_forward(s64)
    eng.unet_forward(two); eng.resnet18_forward(s128)        # another batch in between (eager)
    assert eng.activation("unet", "inc").shape[0] == 2 and eng.activation("resnet18", "layer4").shape[0] == 128
    eng.unet_forward(one); eng.resnet18_forward(s64)         # replayed
    assert eng.activation("unet", "inc").shape[0] == 1 and eng.activation("resnet18", "layer4").shape[0] == 64
    eng.close()


def test_rejected_calibration_import_leaves_the_engine_untouched():
    """an out-of-range exponent at tensor i used to return an error AFTER tensors 0..i-1 were overwritten."""
    from chessvision.hip_backend import HipBackendError, HipEngine

    net = synth.make_resnet(seed=2)
    eng = HipEngine(precision="f16x3", resnet_chunk=128)
    eng.load_resnet18(net.state_dict())
    x = synth.squares_input(seed=61, n=64).cuda()
    before = eng.resnet18_forward(x).cpu()
    cal = eng.export_calibration("resnet18")
    bad = cal.copy()
    bad[:-2] += 1                                            # legal changes in front ...
    bad[-2] = 99                                             # ... of an illegal last entry
    with pytest.raises(HipBackendError):
        eng.import_calibration("resnet18", bad)
    assert np.array_equal(eng.export_calibration("resnet18"), cal)
    assert torch.equal(eng.resnet18_forward(x).cpu(), before)
    eng.close()


@pytest.mark.gpu
def test_fp16_classifier_at_ragged_tiny_batches():
    """Round 5: the fp16 classifier's persistent kernels (chained layer1: one workgroup per square; shortcut kernel: groups of 16 output
    pixels per wave, partial last group) at 1 .. 65 squares: soft-max within configs[2]'s 1e-3 of the oracle, every arg-max equal."""
    from chessvision.hip_backend import HipEngine

    net = synth.make_resnet(2)
    eng = HipEngine(precision="f16r")
    eng.load_resnet18(net.state_dict())
    for n in (1, 2, 3, 5, 17, 63, 65):
        x = synth.squares_input(70 + n, n)
        with torch.no_grad():
            ref = net(x)
        got = eng.resnet18_forward(x.cuda()).cpu()
        assert float((torch.softmax(ref, 1) - torch.softmax(got, 1)).abs().max()) <= 1e-3, n
        assert bool((ref.argmax(1) == got.argmax(1)).all()), n
    eng.check_numerics()
