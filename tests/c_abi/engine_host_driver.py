"""(test infrastructure; run by tests/test_engine_host_sanitizers.py in a child process) Drives the HOST side of the library -- built from
csrc/ with `--cuda-host-only`, AddressSanitizer + UBSan, and linked against tests/c_abi/hip_host_stub.cpp instead of the HIP runtime --
through its C ABI: four precisions, both UNet checkpoint variants, whole-model loads (weight packing, BN folding, offset / position
tables, the chain's and the LDS-resident kernels' weight forms, calibration bookkeeping, rounding-bias folding), forwards at several
batch sizes (launch plans and grids, split-K decisions, workspace growth, chunking), the single-layer entry points at the op tests'
shapes, the one-call `cv_process_image_v2`, profiling tables, calibration export / import, error paths, destroy / trim.  Kernels do not
run (the stand-in turns a launch into a checked no-op), so outputs are not looked at; what is looked at is that the sanitizers stay
silent, that every launch has a grid, block and LDS size the hardware would take, and that every call returns the status it should."""
from __future__ import annotations

import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))

from chessvision import hip_backend as hb  # noqa: E402
from chessvision import synthetic  # noqa: E402

OK, ERR_INVALID, ERR_STATE = 0, 1, 3
_vp = ctypes.c_void_p


def ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)


def check(lib, rc, what, want=OK):
    if rc != want:
        raise SystemExit(f"{what}: status {rc} (wanted {want}): {lib.cv_last_error().decode(errors='replace')}")


def main() -> int:
    lib = hb.load_library()
    n = ctypes.c_int(0)
    check(lib, lib.cv_device_count(ctypes.byref(n)), "cv_device_count")
    assert n.value == 1 and lib.cv_abi_version() == hb.ABI_VERSION
    rng = np.random.default_rng(5)
    rsd = synthetic.resnet18_state_dict(2)
    unets = {False: synthetic.unet_state_dict(1, False), True: synthetic.unet_state_dict(1, True)}
    photo = synthetic.board_photo(3)
    calls = 0
    precisions = sorted({v: k for k, v in sorted(hb._PRECISIONS.items(), reverse=True)}.items())      # one name per arithmetic (aliases dropped)
    precisions = [(name, value) for value, name in precisions]
    for prec_name, prec in precisions:
        for bilinear in ((False, True) if prec_name in ("f16x3", "f32") else (False,)):       # the second UNet variant: headline and exact engines
            h = _vp()
            check(lib, lib.cv_engine_create(0, prec, ctypes.byref(h)), f"create {prec_name}")
            # forwards before a load are refused, not crashed
            x1 = np.zeros((1, 3, 256, 256), np.float32)
            lg = np.zeros((1, 1, 256, 256), np.float32)
            check(lib, lib.cv_unet_forward(h, ptr(x1), 1, ptr(lg), None), "forward before load", ERR_STATE)
            check(lib, lib.cv_engine_set_chunk(h, 4, 512), "set_chunk")
            table, cnt, keep = hb._as_param_table(unets[bilinear])
            check(lib, lib.cv_load_unet(h, table, cnt), f"load_unet {prec_name} bilinear={bilinear}")
            table, cnt, keep2 = hb._as_param_table(rsd)
            check(lib, lib.cv_load_resnet18(h, table, cnt), f"load_resnet18 {prec_name}")
            # a state dict with a missing key / a wrong shape names the key instead of reading past a buffer
            broken = dict(rsd)
            broken.pop("layer3.0.downsample.0.weight")
            t2, c2, k2 = hb._as_param_table(broken)
            h2 = _vp()
            check(lib, lib.cv_engine_create(0, prec, ctypes.byref(h2)), "create second")
            assert lib.cv_load_resnet18(h2, t2, c2) != OK and b"layer3.0.downsample.0.weight" in lib.cv_last_error()
            broken = dict(rsd)
            broken["fc.weight"] = np.zeros((13, 511), np.float32)
            t2, c2, k2 = hb._as_param_table(broken)
            assert lib.cv_load_resnet18(h2, t2, c2) != OK and b"fc.weight" in lib.cv_last_error()
            check(lib, lib.cv_engine_destroy(h2), "destroy second")
            # forwards: batch sizes around the chunk (4 images / 512 squares) -- single board (split-K plans), part chunks, several chunks
            for b in (1, 2, 3, 4, 5, 9):
                x = rng.random((b, 3, 256, 256), dtype=np.float32)
                lg = np.zeros((b, 1, 256, 256), np.float32)
                check(lib, lib.cv_unet_forward(h, ptr(x), b, ptr(lg), None), f"unet_forward b={b}")
                xu = rng.integers(0, 256, (b, 256, 256, 3), dtype=np.uint8)
                mk = np.zeros((b, 256, 256), np.uint8)
                check(lib, lib.cv_unet_forward_u8(h, ptr(xu), b, ptr(lg), ptr(mk), 0.5, None), f"unet_forward_u8 b={b}")
                calls += 2
            for s in (1, 63, 64, 65, 200, 512, 513, 1100):
                sq = rng.random((s, 1, 64, 64), dtype=np.float32)
                out = np.zeros((s, 13), np.float32)
                check(lib, lib.cv_resnet18_forward(h, ptr(sq), s, ptr(out), None), f"resnet18_forward n={s}")
                squ = rng.integers(0, 256, (s, 64, 64), dtype=np.uint8)
                check(lib, lib.cv_resnet18_forward_u8(h, ptr(squ), s, ptr(out), None), f"resnet18_forward_u8 n={s}")
                check(lib, lib.cv_softmax13(h, ptr(out), s, ptr(out), None), "softmax13")
                calls += 3
            ws = ctypes.c_size_t(0)
            check(lib, lib.cv_engine_workspace_bytes(h, ctypes.byref(ws)), "workspace_bytes")
            assert ws.value > 0
            macs = ctypes.c_int64(0)
            check(lib, lib.cv_model_macs(h, b"unet", ctypes.byref(macs)), "model_macs")
            assert macs.value in (48_167_385_088, 39_980_105_728) or macs.value > 3e10, macs.value
            # the whole request in one call (host photo in, host results out), both struct sizes
            res = hb._ImageResult()
            crops = np.zeros((64, 64, 64, 1), np.uint8)
            lg1, mk1 = np.zeros((256, 256), np.float32), np.zeros((256, 256), np.uint8)
            bd, pr = np.zeros((512, 512), np.uint8), np.zeros((64, 13), np.float32)
            res.logits, res.mask, res.board, res.probabilities = ptr(lg1), ptr(mk1), ptr(bd), ptr(pr)
            res.squares = ptr(crops)
            for fallback in (0, 1):
                for flip in (0, 1):
                    check(lib, lib.cv_process_image_v2(h, h, ptr(photo), photo.shape[0], photo.shape[1], 0.5, flip, fallback, ctypes.byref(res),
                                                       ctypes.sizeof(res), None), "process_image_v2")
                    calls += 1
            small = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)          # an odd-sized photo: the staging block regrows
            check(lib, lib.cv_process_image_v2(h, h, ptr(small), 97, 131, 0.5, 0, 1, ctypes.byref(res), ctypes.sizeof(res), None), "odd photo")
            check(lib, lib.cv_process_image_v2(h, h, None, 512, 512, 0.5, 0, 0, ctypes.byref(res), ctypes.sizeof(res), None), "null photo", ERR_INVALID)
            check(lib, lib.cv_process_image_v2(h, h, ptr(photo), 512, 512, 0.5, 0, 0, ctypes.byref(res), 8, None), "short struct", ERR_INVALID)
            # profiling tables of both models (event-timed passes, names, bytes)
            ms, launches, allms = ctypes.c_float(0), ctypes.c_int(0), ctypes.c_float(0)
            x = rng.random((2, 3, 256, 256), dtype=np.float32)
            sq = rng.random((130, 1, 64, 64), dtype=np.float32)
            plg, p13 = np.zeros((2, 1, 256, 256), np.float32), np.zeros((130, 13), np.float32)
            check(lib, lib.cv_profile_convs(h, b"unet", ptr(x), 2, ptr(plg), 1, None, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(allms)), "profile unet")
            name = ctypes.create_string_buffer(256)
            e_ms, e_flop, e_n = ctypes.c_float(0), ctypes.c_double(0), ctypes.c_int(0)
            k = 0
            while lib.cv_profile_entry(h, k, name, 256, ctypes.byref(e_ms), ctypes.byref(e_flop), ctypes.byref(e_n)) == OK:
                nbytes = ctypes.c_double(0)
                check(lib, lib.cv_profile_entry_bytes(h, k, ctypes.byref(nbytes)), "entry_bytes")
                kn = ctypes.create_string_buffer(8)                              # a buffer far too short for a kernel name: truncated, not overrun
                lib.cv_profile_entry_kernel(h, k, kn, 8)
                k += 1
            assert k >= 20, k
            check(lib, lib.cv_profile_convs(h, b"resnet18", ptr(sq), 130, ptr(p13), 1, None, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(allms)), "profile resnet")
            # calibration tables travel between engines (rank 0 -> the others)
            for model in (b"unet", b"resnet18"):
                need = ctypes.c_int(0)
                lib.cv_engine_export_calibration(h, model, None, 0, ctypes.byref(need))
                exps = (ctypes.c_int32 * max(1, need.value))()
                got = ctypes.c_int(0)
                check(lib, lib.cv_engine_export_calibration(h, model, exps, need.value, ctypes.byref(got)), "export_calibration")
                changed = ctypes.c_int(0)
                check(lib, lib.cv_engine_import_calibration(h, model, exps, got.value, ctypes.byref(changed)), "import_calibration")
                if got.value > 1:
                    assert lib.cv_engine_import_calibration(h, model, exps, got.value - 1, ctypes.byref(changed)) != OK   # a table one short is refused
            lib.cv_engine_numeric_status(h, None)
            # activations by name: the documented tap set, and a name that does not exist
            cap = 64 * 128 * 128
            buf = np.zeros(cap, np.float32)
            shape = (ctypes.c_int64 * 4)()
            lib.cv_get_activation(h, b"unet", b"down1", buf.ctypes.data_as(hb._fp), cap, shape)
            assert lib.cv_get_activation(h, b"unet", b"no such tensor", buf.ctypes.data_as(hb._fp), cap, shape) != OK
            assert lib.cv_get_activation(h, b"resnet18", b"layer1", buf.ctypes.data_as(hb._fp), 16, shape) != OK       # capacity too small
            check(lib, lib.cv_engine_destroy(h), "destroy")
            print(f"{prec_name:6s} bilinear={int(bilinear)}: ok", flush=True)

    # single-layer entry points at the shapes of tests/test_gpu_ops.py (tile tails, deep K, strides, residuals, odd sizes)
    conv_cases = [(2, 3, 32, 32, 64, 3, 1), (2, 64, 32, 32, 64, 3, 1), (1, 64, 24, 20, 128, 3, 1), (3, 128, 8, 8, 256, 3, 1), (2, 256, 16, 16, 512, 3, 1),
                  (4, 64, 16, 16, 128, 3, 2), (4, 64, 16, 16, 128, 1, 2), (4, 128, 8, 8, 128, 3, 1), (1, 8, 5, 7, 16, 3, 1), (70, 64, 16, 16, 64, 3, 1),
                  (300, 256, 4, 4, 256, 3, 1), (300, 512, 2, 2, 512, 3, 1)]
    for prec_name, prec in precisions:
        h = _vp()
        check(lib, lib.cv_engine_create(0, prec, ctypes.byref(h)), "create for ops")
        for (nb, cin, hh, ww, cout, k, stride) in conv_cases:
            x = rng.standard_normal((nb, cin, hh, ww)).astype(np.float32)
            w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
            sc, sh = np.ones(cout, np.float32), np.zeros(cout, np.float32)
            ho, wo = (hh + 2 * (k // 2) - k) // stride + 1, (ww + 2 * (k // 2) - k) // stride + 1
            y = np.zeros((nb, cout, ho, wo), np.float32)
            res_in = rng.standard_normal(y.shape).astype(np.float32) if stride == 1 and cin == cout else None
            check(lib, lib.cv_op_conv2d(h, ptr(x), nb, cin, hh, ww, w.ctypes.data_as(hb._fp), cout, k, stride, sc.ctypes.data_as(hb._fp),
                                        sh.ctypes.data_as(hb._fp), ptr(res_in) if res_in is not None else None, 1, ptr(y), None), f"op_conv2d {prec_name} {cin}->{cout}")
            calls += 1
        for (nb, cin, hh, ww, cout) in ((2, 64, 8, 8, 32), (1, 128, 5, 3, 64), (3, 256, 16, 16, 128), (9, 128, 128, 128, 64)):
            x = rng.standard_normal((nb, cin, hh, ww)).astype(np.float32)
            w = rng.standard_normal((cin, cout, 2, 2)).astype(np.float32)
            b = np.zeros(cout, np.float32)
            y = np.zeros((nb, cout, 2 * hh, 2 * ww), np.float32)
            check(lib, lib.cv_op_conv_transpose2x2(h, ptr(x), nb, cin, hh, ww, w.ctypes.data_as(hb._fp), cout, b.ctypes.data_as(hb._fp), ptr(y), None), "op_convT")
        for (nb, c, hh, ww) in ((2, 64, 16, 16), (1, 3, 6, 10), (2, 13, 9, 9)):
            x = rng.standard_normal((nb, c, hh, ww)).astype(np.float32)
            y = np.zeros((nb, c, 2 * hh, 2 * ww), np.float32)
            check(lib, lib.cv_op_maxpool2x2(h, ptr(x), nb, c, hh - hh % 2, ww - ww % 2, ptr(y), None), "op_maxpool2x2")
            check(lib, lib.cv_op_maxpool3x3s2(h, ptr(x), nb, c, hh, ww, ptr(y), None), "op_maxpool3x3s2")
            check(lib, lib.cv_op_upsample_bilinear2x(h, ptr(x), nb, c, hh, ww, ptr(y), None), "op_upsample")
        x = rng.standard_normal((2, 64, 32, 32)).astype(np.float32)
        w, b = rng.standard_normal(64).astype(np.float32), np.zeros(1, np.float32)
        lg, mk = np.zeros((2, 1, 32, 32), np.float32), np.zeros((2, 32, 32), np.uint8)
        check(lib, lib.cv_op_outc_1x1(h, ptr(x), 2, 64, 32, 32, w.ctypes.data_as(hb._fp), b.ctypes.data_as(hb._fp), 0.5, ptr(lg), ptr(mk), None), "op_outc")
        assert lib.cv_op_conv2d(h, None, 1, 8, 4, 4, w.ctypes.data_as(hb._fp), 8, 3, 1, None, None, None, 0, ptr(lg), None) == ERR_INVALID
        # arguments no caller should send come back as a status -- never as a launch with an empty grid or a size computed from them
        xin, yout, w8 = np.zeros((1, 8, 8, 8), np.float32), np.zeros((1, 8, 16, 16), np.float32), np.zeros((16, 8, 3, 3), np.float32)
        for (kk, st) in ((2, 1), (5, 1), (3, 0), (0, 1), (-3, 1)):
            assert lib.cv_op_conv2d(h, ptr(xin), 1, 8, 8, 8, w8.ctypes.data_as(hb._fp), 16, kk, st, None, None, None, 0, ptr(yout), None) == ERR_INVALID, (kk, st)
        for (hh, ww) in ((0, 8), (8, 0), (-1, 8), (8, -2), (1, 1)):
            want = ERR_INVALID
            assert lib.cv_op_conv2d(h, ptr(xin), 1, 8, hh, ww, w8.ctypes.data_as(hb._fp), 16, 3, 1, None, None, None, 0, ptr(yout), None) == (OK if hh > 0 and ww > 0 else want)
            assert lib.cv_op_maxpool2x2(h, ptr(xin), 1, 8, hh, ww, ptr(yout), None) == want, (hh, ww)                 # 1 x 1: no output pixel
            assert lib.cv_op_maxpool3x3s2(h, ptr(xin), 1, 8, hh, ww, ptr(yout), None) == (OK if hh > 0 and ww > 0 else want), (hh, ww)
            assert lib.cv_op_upsample_bilinear2x(h, ptr(xin), 1, 8, hh, ww, ptr(yout), None) == (OK if hh > 0 and ww > 0 else want), (hh, ww)
        for b in (-1, -2 ** 31):
            assert lib.cv_softmax13(h, ptr(lg), b, ptr(lg), None) == ERR_INVALID
        for args in ((0, 16, 16, 3, 8, 8), (1, 0, 16, 3, 8, 8), (1, 16, 16, 3, 0, 8), (-1, 16, 16, 3, 8, 8)):
            assert lib.cv_resize_area_u8(h, ptr(xin), *args[:4], ptr(yout), args[4], args[5], None) == ERR_INVALID, args
        assert lib.cv_engine_create(7, prec, ctypes.byref(_vp())) == ERR_INVALID and lib.cv_engine_create(0, 99, ctypes.byref(_vp())) == ERR_INVALID
        assert lib.cv_engine_destroy(None) == OK
        # byte-path entry points: resize to odd targets, the warp from homographies
        img = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
        for (oh, ow) in ((256, 256), (100, 77), (480, 640), (1, 1)):
            dst = np.zeros((3, oh, ow, 3), np.uint8)
            check(lib, lib.cv_resize_area_u8(h, ptr(img), 3, 480, 640, 3, ptr(dst), oh, ow, None), f"resize_area {oh}x{ow}")
        inv = np.tile(np.eye(3, dtype=np.float64).reshape(1, 9), (3, 1))
        sqs, bds = np.zeros((3 * 64, 64, 64), np.uint8), np.zeros((3, 512, 512), np.uint8)
        check(lib, lib.cv_extract_squares_u8(h, ptr(img), 3, 480, 640, inv.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ptr(sqs), ptr(bds), None), "extract_squares")
        check(lib, lib.cv_engine_destroy(h), "destroy ops engine")
        print(f"{prec_name:6s} single-layer entry points: ok", flush=True)

    # the block cache: an engine like one that was just destroyed is served entirely from the blocks that one left behind
    lib.cv_stub_launches.restype = lib.cv_stub_bad_launches.restype = lib.cv_stub_allocations.restype = ctypes.c_long

    def load_and_drop():
        h = _vp()
        check(lib, lib.cv_engine_create(0, hb._PRECISIONS["f16x3"], ctypes.byref(h)), "create for the cache")
        table, cnt, keep = hb._as_param_table(rsd)
        check(lib, lib.cv_load_resnet18(h, table, cnt), "load for the cache")
        sq = np.zeros((70, 1, 64, 64), np.float32)
        out = np.zeros((70, 13), np.float32)
        check(lib, lib.cv_resnet18_forward(h, ptr(sq), 70, ptr(out), None), "forward for the cache")
        check(lib, lib.cv_engine_destroy(h), "destroy for the cache")

    load_and_drop()
    before = lib.cv_stub_allocations()
    load_and_drop()
    reused = lib.cv_stub_allocations() - before
    assert reused == 0, f"{reused} blocks came from the runtime although an identical engine had just been destroyed"
    freed = ctypes.c_size_t(0)
    check(lib, lib.cv_trim_memory(ctypes.byref(freed)), "trim")
    assert freed.value > 0
    again = ctypes.c_size_t(1)
    check(lib, lib.cv_trim_memory(ctypes.byref(again)), "second trim")
    assert again.value == 0
    before = lib.cv_stub_allocations()
    load_and_drop()
    assert lib.cv_stub_allocations() - before > 20                      # ... and after a trim everything comes from the runtime again
    check(lib, lib.cv_trim_memory(ctypes.byref(freed)), "last trim")
    print(f"engine host sanitizers: ok -- {calls} forward / op calls, {lib.cv_stub_launches()} launches checked "
          f"({lib.cv_stub_bad_launches()} refused), {lib.cv_stub_allocations()} allocations, {freed.value >> 20} MB trimmed", flush=True)
    return 1 if lib.cv_stub_bad_launches() else 0


if __name__ == "__main__":
    sys.exit(main())
