"""GPU: the reference's own integration tests (tests/test_chessvision.py:45-116) restated against the HIP-backed
ChessVision with random-init checkpoints in the reference's file formats, plus the batched API."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from chessvision import ChessVision, constants, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv_model(tmp_path_factory):
    d = tmp_path_factory.mktemp("weights")
    pe, pc = synthetic.save_checkpoints(d)
    return ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), lazy_load=True)


def _board_photo(seed=0):
    """512x512 BGR: a bright quadrilateral 'board' with an 8x8 checker texture on dark noise (BASELINE config 1)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 40, (512, 512, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:512, 0:512]
    inside = (xx > 90 + 0.05 * yy) & (xx < 430 - 0.04 * yy) & (yy > 70) & (yy < 440)
    checker = (((xx - 90) // 43 + (yy - 70) // 46) % 2).astype(np.uint8)
    img[inside] = (150 + 80 * checker[inside])[:, None]
    return img


def test_lazy_state_then_models_are_hip_objects(cv_model):
    assert cv_model._board_extractor is None and cv_model._classifier is None
    from chessvision.hip_backend import HipBoardExtractor, HipPieceClassifier

    assert isinstance(cv_model.board_extractor, HipBoardExtractor)
    assert isinstance(cv_model.classifier, HipPieceClassifier)
    assert cv_model._classifier_model_id == "resnet18"                 # fallback branch of core.py:121-130
    assert cv_model.board_extractor.metadata["synthetic"] is True      # checkpoint metadata is carried over


def test_process_image_result_contract(cv_model):
    result = cv_model.process_image(_board_photo())
    assert result.board_extraction is not None
    assert isinstance(result.board_extraction.binary_mask, np.ndarray)
    assert result.board_extraction.binary_mask.dtype == np.uint8
    assert result.board_extraction.probabilities.shape == (256, 256)
    if result.board_extraction.board_image is not None:
        assert result.board_extraction.board_image.shape == (512, 512)
        assert result.position is not None and result.position.fen.count("/") == 7
        assert result.position.model_probabilities.shape == (64, 13)
        assert (len(result.position.validation_fixes) > 0) == (result.position.original_fen != result.position.fen)
    else:
        assert result.position is None
    assert result.processing_time > 0


def test_classify_position_on_a_given_board(cv_model):
    board = np.random.default_rng(1).integers(0, 256, (512, 512), dtype=np.uint8)
    res = cv_model.classify_position(board)
    assert res.squares.shape == (64, 64, 64, 1) and res.model_probabilities.shape == (64, 13)
    assert np.allclose(res.model_probabilities.sum(axis=1), 1.0, atol=1e-5)
    assert res.square_names == constants.SQUARE_NAMES_NORMAL
    flipped = cv_model.classify_position(board, flip=True)
    assert flipped.square_names == constants.SQUARE_NAMES_FLIPPED
    assert np.allclose(flipped.model_probabilities, res.model_probabilities)
    for fix in res.validation_fixes:
        assert fix.square_name in res.square_names and fix.corrected_piece in constants.LABEL_NAMES


def test_batched_api_equals_per_image_api(cv_model):
    images = [_board_photo(s) for s in range(3)]
    single = [cv_model.process_image(im) for im in images]
    batched = cv_model.process_images(images)
    assert len(batched) == 3
    for a, b in zip(single, batched):
        assert np.abs(a.board_extraction.probabilities - b.board_extraction.probabilities).max() <= 1e-4
        assert float((a.board_extraction.binary_mask != b.board_extraction.binary_mask).mean()) <= 1e-4
        assert (a.position is None) == (b.position is None)
        if a.position is not None:
            assert np.array_equal(a.board_extraction.quadrangle, b.board_extraction.quadrangle)
            assert np.array_equal(a.board_extraction.board_image, b.board_extraction.board_image)   # same device warp, same matrix
            assert np.abs(a.position.model_probabilities - b.position.model_probabilities).max() <= 1e-4
            assert a.position.fen == b.position.fen and a.position.original_fen == b.position.original_fen
    assert cv_model.process_images([]) == []


def test_device_resize_matches_host_restatement(engines):
    import torch

    from chessvision import classical

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (3, 512, 512, 3), dtype=np.uint8)
    got = engines["f32"].resize_area_u8(torch.from_numpy(img), (256, 256)).cpu().numpy()
    want = np.stack([classical.resize_area(im, (256, 256)) for im in img])
    assert np.array_equal(got, want)                                       # integer factor: exact box mean
    from oracle import classical_ref as cref

    for shape in ((2, 300, 400, 3), (1, 384, 512, 3), (1, 257, 641, 3)):    # fractional shrink: OpenCV's float32 table form, bit-exact
        odd = rng.integers(0, 256, shape, dtype=np.uint8)
        got = engines["f32"].resize_area_u8(torch.from_numpy(odd), (256, 256)).cpu().numpy()
        want = np.stack([classical.resize_area(im, (256, 256)) for im in odd])
        assert np.array_equal(got, want), shape                              # device == host
        assert np.array_equal(got[0], cref.resize_area(odd[0], (256, 256)))  # ... == the independent oracle
    for _ in range(8):                                                       # random sizes, tables rebuilt per geometry
        h, w = int(rng.integers(256, 900)), int(rng.integers(256, 900))
        odd = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        got = engines["f32"].resize_area_u8(torch.from_numpy(odd), (256, 256)).cpu().numpy()
        assert np.array_equal(got, np.stack([classical.resize_area(im, (256, 256)) for im in odd])), (h, w)


def test_device_resize_enlarging_matches_host_and_oracle(engines):
    """INTER_AREA on photos smaller than 256 px (reference core.py:212): OpenCV's fixed-point bilinear path with the AREA coefficient
    rule.  device == host == independent oracle, byte for byte, on 8 random sizes below 256 px, mixed enlarge / shrink geometries and
    the whole path through ``process_images``."""
    import torch

    from chessvision import classical
    from oracle import classical_ref as cref

    rng = np.random.default_rng(15)
    sizes = [tuple(int(v) for v in rng.integers(24, 256, 2)) for _ in range(8)] + [(100, 400), (400, 100), (256, 128), (255, 257)]
    for h, w in sizes:
        img = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        got = engines["f32"].resize_area_u8(torch.from_numpy(img), (256, 256)).cpu().numpy()
        assert np.array_equal(got, np.stack([classical.resize_area(im, (256, 256)) for im in img])), (h, w)      # device == host
        assert np.array_equal(got[0], cref.resize_area_enlarge(img[0], (256, 256))), (h, w)                       # == oracle


def test_device_warp_gray_flip_split_matches_host_chain(engines):
    import torch

    from chessvision import classical, utils
    from chessvision.core import ChessVision

    rng = np.random.default_rng(6)
    imgs = rng.integers(0, 256, (2, 384, 512, 3), dtype=np.uint8)
    quads = [np.array([[400, 60], [90, 40], [60, 330], [430, 350]], np.float32),
             np.array([[500, 10], [20, 5], [-30, 370], [530, 400]], np.float32)]      # second one leaves the frame
    from chessvision.hip_backend import board_homographies
    from oracle import classical_ref as cref

    quads.append(np.array([[255, 0], [0, 0], [0, 255], [255, 255]], np.float32) * np.float32(384 / 256.0))   # whole-image fallback: dyadic ties
    imgs = np.concatenate([imgs, imgs[:1]])
    inv = board_homographies(np.stack(quads), (512, 512))
    squares, boards = engines["f32"].extract_squares_u8(torch.from_numpy(imgs), inv)
    for k in range(3):
        board = classical.flip_horizontal(classical.bgr_to_gray(utils.extract_perspective(imgs[k], quads[k], (512, 512))))
        assert np.array_equal(boards[k].cpu().numpy(), board)                  # device == host, byte for byte
        independent = cref.flip_lr(cref.bgr_to_gray(cref.extract_board(imgs[k], quads[k], (512, 512))))
        assert np.array_equal(boards[k].cpu().numpy(), independent)            # ... == the independent oracle
        want = ChessVision.extract_squares(boards[k].cpu().numpy())[..., 0]
        assert np.array_equal(squares[k * 64:(k + 1) * 64].cpu().numpy(), want)


def test_batched_api_with_mixed_sizes_and_missing_boards(tmp_path_factory):
    """One call with images of two sizes (jobs are formed per shape), a frame without a board (position None, no fallback)
    and a non-square photo: every result equals the per-image API's (device resize / warp vs the numpy restatements)."""
    d = tmp_path_factory.mktemp("weights_seg")
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    rng = np.random.default_rng(3)
    wide = np.zeros((384, 512, 3), np.uint8)
    wide[:, :384] = synthetic.board_photo(5, 384)
    images = [synthetic.board_photo(1), rng.integers(0, 50, (512, 512, 3), dtype=np.uint8), wide, synthetic.board_photo(2)]
    batched = cv.process_images(images, return_crops=True)
    assert [r.position is None for r in batched] == [False, True, False, False]
    for im, b in zip(images, batched):
        a = cv.process_image(im)
        assert (a.position is None) == (b.position is None)
        assert float((a.board_extraction.binary_mask != b.board_extraction.binary_mask).mean()) <= 1e-4
        if a.position is None:
            assert b.board_extraction.board_image is None and b.board_extraction.quadrangle is None
            continue
        assert np.array_equal(a.board_extraction.quadrangle, b.board_extraction.quadrangle)
        if np.array_equal(a.board_extraction.binary_mask, b.board_extraction.binary_mask):
            assert np.array_equal(a.board_extraction.board_image, b.board_extraction.board_image)
        assert np.abs(a.position.model_probabilities - b.position.model_probabilities).max() <= 1e-3
        assert b.position.squares.shape == (64, 64, 64, 1)


def test_concurrent_request_threads_share_one_lazy_instance(tmp_path_factory):
    """The reference's Flask app keeps ONE global ChessVision(lazy_load=...) and serves requests from several threads
    (app/computeroot/cv_endpoint.py:131-133) with no lock of its own.  Eight threads hit a still-lazy instance at once: the models
    are initialised exactly once, the forwards serialise on the engine's mutex, and every thread gets the result the sequential
    call gives."""
    import threading

    d = tmp_path_factory.mktemp("weights_threads")
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    images = [synthetic.board_photo(900 + k) for k in range(8)]
    ref_cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    want = [ref_cv.process_image(im) for im in images]

    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))      # lazy: nothing loaded yet
    assert cv._board_extractor is None and cv._classifier is None
    got, errors = [None] * 8, []
    start = threading.Barrier(8)

    def worker(k):
        try:
            start.wait()
            for _ in range(3):                                  # a few rounds so that forwards really interleave
                got[k] = cv.process_image(images[k])
        except Exception as exc:                                # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert len(cv._engines) == 1                                # one engine, created once
    for g, w in zip(got, want):
        assert np.array_equal(g.board_extraction.binary_mask, w.board_extraction.binary_mask)
        assert np.array_equal(g.board_extraction.probabilities, w.board_extraction.probabilities)
        assert (g.position is None) == (w.position is None)
        if w.position is not None:
            assert g.position.fen == w.position.fen
            assert np.array_equal(g.position.model_probabilities, w.position.model_probabilities)


def test_four_request_threads_overlap_on_the_device_and_return_the_serial_results(tmp_path_factory):
    """VERDICT r05 item 4.  One ChessVision instance, four request threads x 200 ``process_image`` calls (the reference's Flask app:
    a global instance behind a threaded server, app/computeroot/cv_endpoint.py:131-133,159).  Round 5 queued them behind ONE staging
    block (~1050 requests/s whatever the thread count, three quarters of the chip idle); now every in-flight request holds a request
    slot -- its own engines, page-locked block, stream and hipGraphs -- and the B=1 forwards run side by side.  Every one of the 800
    results is bit-identical to the serial call, and the aggregate rate beats one thread's by a clear factor."""
    import threading
    import time

    d = tmp_path_factory.mktemp("weights_slots")
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    images = [synthetic.board_photo(700 + k) for k in range(8)]
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    want = [cv.process_image(im) for im in images]
    assert all(w.position is not None for w in want)
    t0 = time.perf_counter()
    for k in range(200):
        cv.process_image(images[k % 8])
    serial = 200 / (time.perf_counter() - t0)
    assert len(cv._slots) == 1                                    # a single-threaded caller never pays for replicas
    assert cv.warm_request_slots(4) == 4
    errors, bad = [], []
    warm = [threading.Thread(target=lambda t=t: [cv.process_image(images[(t + k) % 8]) for k in range(20)]) for t in range(4)]
    for th in warm:                                            # a replica's first call grows its workspace, its second records the hipGraphs
        th.start()
    for th in warm:
        th.join(timeout=600)

    def worker(t):
        try:
            for k in range(200):
                r, w = cv.process_image(images[(t + k) % 8]), want[(t + k) % 8]
                same = (np.array_equal(r.board_extraction.binary_mask, w.board_extraction.binary_mask)
                        and np.array_equal(r.board_extraction.probabilities, w.board_extraction.probabilities)
                        and np.array_equal(r.board_extraction.board_image, w.board_extraction.board_image)
                        and r.position.fen == w.position.fen and np.array_equal(r.position.model_probabilities, w.position.model_probabilities)
                        and np.array_equal(r.position.squares, w.position.squares))
                if not same:
                    bad.append((t, k))
        except Exception as exc:                                # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    rate = 800 / (time.perf_counter() - t0)
    print(f"process_image: one thread {serial:.0f}/s, four threads on four slots {rate:.0f}/s")
    assert not errors and not bad, (errors[:3], bad[:5])
    assert len(cv._slots) == 4 and len(cv._engines) == 1
    assert rate >= 1.6 * serial, (rate, serial)


def test_classify_position_sees_in_place_edits_of_the_extracted_board(tmp_path):
    """`classify_position(result.board_image)` reuses the squares that are still on the device -- but only while the
    array still holds the pixels that were rectified.  A caller that edits the board in place (masks a square, draws on it) gets the
    edited board classified, as the reference does (it classifies the array it is handed, core.py:225-249)."""
    pe, pc = synthetic.save_checkpoints(tmp_path, segmenting=True)            # a UNet that finds the synthetic board
    cv_model = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), lazy_load=True)
    img = synthetic.board_photo(3)
    ext = cv_model.extract_board(img)
    assert ext.board_image is not None
    untouched = cv_model.classify_position(ext.board_image)                 # device squares reused: same pixels
    ext2 = cv_model.extract_board(img)
    ext2.board_image[:256, :256] = 255 - ext2.board_image[:256, :256]         # edit in place between the two calls
    edited = cv_model.classify_position(ext2.board_image)
    fresh = cv_model.classify_position(ext2.board_image.copy())              # a different array object: always staged from the host
    assert np.array_equal(edited.model_probabilities, fresh.model_probabilities)
    assert np.array_equal(edited.squares, fresh.squares)
    assert not np.array_equal(edited.model_probabilities, untouched.model_probabilities)


def test_repeated_process_image_calls_never_return_stale_staging(tmp_path):
    """Round 5: ``cv_process_image`` lets the UNet head write the mask and the classifier head write the probabilities straight into the
    engine's page-locked block, sends the rectified board home on a side stream beside the classifier and passes the warp matrix as a
    kernel argument.  300 calls alternating three photos (two with a board, one without): every result equals the first result of its
    photo bit for bit -- nothing of the previous call's staging is ever seen."""
    pe, pc = synthetic.save_checkpoints(tmp_path, segmenting=True)
    cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc))
    photos = [synthetic.board_photo(11, 512), np.random.default_rng(5).integers(0, 30, (512, 512, 3), dtype=np.uint8), synthetic.board_photo(12, 512)]
    first = [cv.process_image(p) for p in photos]
    assert first[0].position is not None and first[2].position is not None and first[1].position is None
    assert first[0].position.fen != first[2].position.fen or not np.array_equal(first[0].board_extraction.board_image, first[2].board_extraction.board_image)
    for it in range(300):
        k = (it * 7 + it // 5) % 3
        r, f = cv.process_image(photos[k]), first[k]
        assert np.array_equal(r.board_extraction.binary_mask, f.board_extraction.binary_mask), it
        assert np.array_equal(r.board_extraction.probabilities, f.board_extraction.probabilities), it
        assert (r.position is None) == (f.position is None), it
        if f.position is not None:
            assert np.array_equal(r.board_extraction.board_image, f.board_extraction.board_image), it
            assert np.array_equal(r.position.model_probabilities, f.position.model_probabilities) and r.position.fen == f.position.fen, it


def test_close_returns_the_device_memory_and_the_instance_reloads_lazily(tmp_path_factory):
    """`ChessVision.close()` (round 6: an instance may now hold replica engines for its request slots and an exact-f32 twin) releases
    engines, slots, staging buffers and streams at once; the instance is lazy again and the next call gives the same bits.  The
    device blocks go to the library's process-wide cache (`cv_trim_memory`, include/chessvision_hip.h): a reload takes them from
    there instead of from the driver, and `trim_memory()` hands them back."""
    from chessvision import hip_backend
    d = tmp_path_factory.mktemp("weights_close")
    pe, pc = synthetic.save_checkpoints(d, segmenting=True)
    image = synthetic.board_photo(321)
    torch.cuda.synchronize()
    hip_backend.trim_memory()                                       # blocks of instances closed by earlier tests
    free0 = torch.cuda.mem_get_info()[0]
    with ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc)) as cv:
        first = cv.process_image(image)
        assert cv.warm_request_slots(2) == 2
        cv.process_images([image] * 4, fallback_quad=True)
        held = free0 - torch.cuda.mem_get_info()[0]
        assert held > 500 << 20, held                               # two engine pairs + their workspaces
    assert cv._slots == [] and cv._engines == {} and cv._board_extractor is None
    torch.cuda.synchronize()
    again = cv.process_image(image)                                 # lazy reload ...
    assert cv.warm_request_slots(2) == 2
    cv.process_images([image] * 4, fallback_quad=True)
    regrown = free0 - torch.cuda.mem_get_info()[0]
    assert regrown < held + (64 << 20), (regrown, held)             # ... out of the cached blocks: nothing new from the driver
    assert again.position is not None and again.position.fen == first.position.fen
    assert np.array_equal(again.position.model_probabilities, first.position.model_probabilities)
    assert np.array_equal(again.board_extraction.probabilities, first.board_extraction.probabilities)
    cv.close()
    freed = hip_backend.trim_memory()
    assert freed > 500 << 20, freed
    assert hip_backend.trim_memory() == 0
    torch.cuda.synchronize()
    leaked = free0 - torch.cuda.mem_get_info()[0]
    assert leaked < 64 << 20, (leaked, held)                        # everything came back (torch's own caching aside)
    again = cv.process_image(image)                                 # and the instance still reloads after a trim
    assert again.position is not None and again.position.fen == first.position.fen
    cv.close(); cv.close()                                          # idempotent


def test_request_threads_beside_batches_and_instances_that_come_and_go():
    """The round-6 soak in small (tests/dev/slots_soak.py, in a process of its own: what it guards against is a device fault, which
    ends the process): four request threads on the slots of one instance, one thread running `process_images` batches on the same
    instance and one creating, using and closing OTHER instances.  Every single-image result equals the serial one bit for bit.  What
    it found (profiles/r06_tuning.md section 8): forwards of one engine on two streams overlapped (6567 of 12000 results wrong),
    and instances closed beside running forwards ended in "Memory access fault by GPU" in roughly a third of the runs."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "tests" / "dev" / "slots_soak.py"), "4", "250"], capture_output=True, text=True, timeout=600)
    tail = "\n".join((out.stdout + out.stderr).strip().splitlines()[-6:])
    assert out.returncode == 0, tail
    assert "1000 results" in out.stdout and " 0 differing from the serial result, 0 errors" in out.stdout, tail
    assert "instances created, used twice and closed meanwhile: 0" not in out.stdout, tail
