"""The C ABI from plain C: tests/c_abi/consumer.c is compiled with gcc against include/chessvision_hip.h and linked to the
library -- no Python, no torch, no ctypes in the call path.  What a cgo / JNI / C++ host of the reference's path would do.

CPU: the host-only entry points (mask -> quadrangle, probabilities -> FEN + pawn rule, status/last-error behaviour).
GPU: a ResNet-18 state dict from a flat file -> cv_load_resnet18 -> cv_resnet18_forward_u8 on device buffers the C program
allocates itself with the HIP runtime's C API -> probabilities equal to the Python binding's and within 1e-3 of the oracle."""
from __future__ import annotations

import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIBDIR = ROOT / "chessvision-3lc_amd" / "lib"
SRC = ROOT / "tests" / "c_abi" / "consumer.c"


def _build(out: Path, gpu: bool) -> Path:
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(SRC), "-o", str(out),
           f"-L{LIBDIR}", "-lchessvision_hip", f"-Wl,-rpath,{LIBDIR}"]
    if gpu:
        cmd += ["-DWITH_GPU", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_host_entry_points_from_plain_c(tmp_path):
    exe = _build(tmp_path / "consumer_host", gpu=False)
    out = subprocess.run([str(exe), "host"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict(ln.split(" ", 1) for ln in out.stdout.strip().splitlines())
    assert lines["abi"] == "6"
    assert float(lines["homography_err2"]) < 1e-16                        # cv_board_homographies from C: corners -> board corners, inv * fwd = I
    quad = [int(v) for v in lines["quad"].split()]
    assert quad[0] == 1                                                   # found; vertices = TR, TL, BL, BR of the drawn shape
    pts = np.array(quad[1:]).reshape(4, 2)
    assert np.abs(pts - np.array([[209, 60], [50, 60], [63, 199], [200, 199]])).max() <= 2
    assert lines["fen0"] == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"
    assert lines["orig1"] == "pnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"  # the forced pawn on a8 ...
    assert lines["fen1"] == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"   # ... becomes the runner-up rook
    assert lines["fixes"].split() == ["1", "1", "0", "9", "11"]             # one fix: board 1, square 0 (a8), 'p' -> 'r'
    assert lines["null_mask"].startswith("rc=1 msg=") and len(lines["null_mask"]) > len("rc=1 msg=")


@pytest.mark.gpu
def test_classifier_from_plain_c_matches_python_binding_and_oracle(tmp_path):
    import torch

    from chessvision import synthetic
    from chessvision.hip_backend import HipEngine
    from oracle import synth

    sd = synthetic.resnet18_state_dict(2)
    blob = tmp_path / "state.blob"
    with open(blob, "wb") as f:
        f.write(struct.pack("<i", len(sd)))
        for name, arr in sd.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", a.ndim))
            f.write(struct.pack("<4q", *(list(a.shape) + [0] * (4 - a.ndim))))
            f.write(a.tobytes())
    rng = np.random.default_rng(4)
    squares = rng.integers(0, 256, (96, 64, 64), dtype=np.uint8)
    sq_path = tmp_path / "squares.bin"
    sq_path.write_bytes(struct.pack("<i", len(squares)) + squares.tobytes())

    exe = _build(tmp_path / "consumer_gpu", gpu=True)
    out = subprocess.run([str(exe), "gpu", str(blob), str(sq_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [ln.split()[1:] for ln in out.stdout.splitlines() if ln.startswith("probs")]
    got = np.array(rows, dtype=np.float64).astype(np.float32)
    assert got.shape == (96, 13)
    assert any(ln.startswith("unet_not_loaded rc=3") for ln in out.stdout.splitlines())
    trimmed = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("trimmed ")][0]
    assert int(trimmed[1]) > 20 << 20 and int(trimmed[3]) == 0            # cv_trim_memory from C: the classifier's blocks, then nothing left

    eng = HipEngine(precision="f16x3")
    eng.load_resnet18(sd)
    py = eng.resnet18_forward_u8(torch.from_numpy(squares)).cpu().numpy()
    eng.close()
    assert np.array_equal(got, py)                                        # same library, same arithmetic: bit-equal
    net = synth.make_resnet(seed=2)
    with torch.no_grad():
        ref = torch.softmax(net(torch.from_numpy(squares).float().div(255)[:, None]), 1).numpy()
    assert np.abs(got - ref).max() <= 1e-3


def _write_blob(path, sd):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(sd)))
        for name, arr in sd.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", a.ndim))
            f.write(struct.pack("<4q", *(list(a.shape) + [0] * (4 - a.ndim))))
            f.write(a.tobytes())


@pytest.mark.gpu
def test_process_image_from_plain_c_equals_the_python_class(tmp_path):
    """`cv_process_image`: the reference's per-image entry point (core.py:152-195) as ONE call from a C host -- photo in, mask /
    quadrangle / board / probabilities / FEN out.  Same bits as `ChessVision.process_image` (which wraps the same call) and as
    the batched `process_images`."""
    from chessvision import ChessVision, synthetic

    usd, rsd = synthetic.unet_state_dict(1, segmenting=True), synthetic.resnet18_state_dict(2)
    _write_blob(tmp_path / "unet.blob", usd)
    _write_blob(tmp_path / "resnet.blob", rsd)
    image = synthetic.board_photo(77)
    (tmp_path / "image.bin").write_bytes(struct.pack("<2i", image.shape[0], image.shape[1]) + image.tobytes())
    exe = _build(tmp_path / "consumer_image", gpu=True)
    out = subprocess.run([str(exe), "image", str(tmp_path / "unet.blob"), str(tmp_path / "resnet.blob"), str(tmp_path / "image.bin")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = dict(ln.split(" ", 1) for ln in out.stdout.strip().splitlines() if " " in ln)
    pe, pc = synthetic.save_checkpoints(tmp_path, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    want = cv.process_images([image], fallback_quad=True)[0]
    assert lines["pi_found"] == "1" and want.position is not None
    assert lines["pi_fen"] == want.position.fen and lines["pi_orig"] == want.position.original_fen
    assert np.array_equal(np.array(lines["pi_quad"].split(), dtype=np.float64).astype(np.float32).reshape(4, 1, 2), want.board_extraction.quadrangle)
    assert int(lines["pi_mask_sum"]) == int(want.board_extraction.binary_mask.astype(np.uint64).sum())
    board = want.board_extraction.board_image.reshape(-1).astype(np.uint64)
    assert int(lines["pi_board_checksum"]) == int((board * (np.arange(board.size, dtype=np.uint64) % 251 + 1)).sum())
    crops = ChessVision.extract_squares(want.board_extraction.board_image).reshape(-1).astype(np.uint64)      # the reference's tiling (utils.py:115-132)
    assert int(lines["pi_squares_checksum"]) == int((crops * (np.arange(crops.size, dtype=np.uint64) % 253 + 1)).sum())
    probs = np.array(lines["pi_probs"].split(), dtype=np.float64).astype(np.float32).reshape(64, 13)
    assert np.array_equal(probs, want.position.model_probabilities)
    assert lines["pi_null_image"].startswith("rc=1 msg=")
    # ABI 6: a binary built against the ABI 3 / 4 struct (no `squares`) calls the old entry point -- the bytes behind its struct stay
    # untouched and the results are the same; a struct too short for even that layout is refused
    assert lines["pi_legacy_ok"] == "1 found 1 same_fen 1 same_probs 1", lines["pi_legacy_ok"]
    assert lines["pi_short_struct"] == "rc=1"
