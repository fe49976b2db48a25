"""Host logic of the request slots (chessvision/core.py: _acquire_slot, _build_slot, _warm_slot) with the native calls replaced by
recorders -- no GPU.  What is pinned here is ORDER: a slot serves two blank-photo requests before anybody else can see it, slots of
the process are warmed one at a time (the HIP runtime binds a stream to a hardware queue at the stream's first use; slots first used
together shared queues for good: profiles/r06_tuning.md section 5), and a single-threaded caller never pays for replicas."""
from __future__ import annotations

import threading
import time

import numpy as np
import pytest

from chessvision import ChessVision, core, hip_backend


class _FakeStream:
    made = 0

    def __init__(self, device=None):
        type(self).made += 1
        self.cuda_stream = 1000 + type(self).made


class _FakeEngine:
    made = 0

    def __init__(self, device=None, precision="f16x3", **kw):
        type(self).made += 1
        self.ident, self.precision, self.loaded = type(self).made, precision, []

    def load_unet(self, state):
        self.loaded.append("unet")

    def load_resnet18(self, state):
        self.loaded.append("resnet18")

    def close(self):
        pass


@pytest.fixture()
def slots_env(monkeypatch):
    """A ChessVision whose models are `_HipModel`s around fake engines; every native request is recorded as
    (engine ident, stream, blank?, thread name) and takes `delay[0]` seconds."""
    calls, delay, in_warm, overlaps = [], [0.0], [0], [0]
    lock = threading.Lock()

    def fake_native(unet_engine, classifier_engine, image, threshold=0.5, flip=False, fallback_quad=False, stream=None):
        blank = not image.any()
        with lock:
            calls.append((unet_engine.ident, stream, blank, threading.current_thread().name))
            if blank:
                in_warm[0] += 1
                overlaps[0] = max(overlaps[0], in_warm[0])
        time.sleep(delay[0] if not blank else 0.01)
        if blank:
            with lock:
                in_warm[0] -= 1
        return {"found": False, "mask": np.zeros((256, 256), np.uint8), "logits": np.zeros((256, 256), np.float32)}

    _FakeStream.made = 0
    _FakeEngine.made = 0
    monkeypatch.setattr(hip_backend, "process_image_native", fake_native)
    monkeypatch.setattr(hip_backend, "HipEngine", _FakeEngine)
    monkeypatch.setattr(core.torch.cuda, "Stream", _FakeStream)
    monkeypatch.setattr(core.utils, "read_checkpoint", lambda path: ({}, None))
    cv = ChessVision(board_extractor_weights="unet.pth", classifier_weights="resnet.pth")
    primary = _FakeEngine()
    model = hip_backend._HipModel.__new__(hip_backend._HipModel)
    model.engine = primary
    cv._board_extractor = cv._classifier = model
    return cv, calls, delay, overlaps


def _photo(v=7):
    return np.full((32, 32, 3), v, np.uint8)


def test_first_request_warms_slot_zero_and_a_serial_caller_builds_no_replica(slots_env):
    cv, calls, _, _ = slots_env
    for _ in range(5):
        r = cv.process_image(_photo())
        assert r.position is None and r.board_extraction.binary_mask.shape == (256, 256)
    assert [c[2] for c in calls] == [True, True] + [False] * 5        # two blank requests, then the caller's
    assert len({c[0] for c in calls}) == 1 and len({c[1] for c in calls}) == 1   # all on the primary engines and slot 0's stream
    assert len(cv._slots) == 1 and _FakeEngine.made == 1                # nobody was ever kept waiting: no replica


def test_a_replica_is_warmed_before_it_is_published_and_warmups_never_overlap(slots_env):
    cv, calls, delay, overlaps = slots_env
    cv.process_image(_photo())                                          # slot 0 exists and is warm
    delay[0] = 0.05
    threads = [threading.Thread(target=lambda: [cv.process_image(_photo()) for _ in range(12)], name=f"req{t}") for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert len(cv._slots) == 4
    assert overlaps[0] == 1                                             # one slot of the process at a time
    for ident in {c[0] for c in calls}:
        mine = [c for c in calls if c[0] == ident]
        assert [c[2] for c in mine[:2]] == [True, True], ident          # the first two requests a slot ever serves are the blank ones ...
        assert not any(c[2] for c in mine[2:]), ident
        if ident != 1:
            assert all(c[3] == "chessvision-slot-builder" for c in mine[:2]), mine[:2]   # ... sent by its builder, before any request thread
        assert len({c[1] for c in mine}) == 1                           # one stream per slot
    assert len({c[1] for c in calls}) == 4


def test_warm_request_slots_is_a_start_up_hook_and_close_forgets_the_slots(slots_env):
    cv, calls, _, overlaps = slots_env
    assert cv.warm_request_slots(3) == 3
    assert len(calls) == 6 and all(c[2] for c in calls) and overlaps[0] == 1
    cv.process_image(_photo())
    assert sum(not c[2] for c in calls) == 1
    cv.close()
    assert cv._slots == []


def test_a_failing_warm_up_leaves_a_slot_that_serves(slots_env, monkeypatch):
    cv, calls, _, _ = slots_env
    real = hip_backend.process_image_native

    def flaky(unet_engine, classifier_engine, image, *a, **kw):
        if not image.any():
            raise hip_backend.HipBackendError("blank photo refused")
        return real(unet_engine, classifier_engine, image, *a, **kw)

    monkeypatch.setattr(hip_backend, "process_image_native", flaky)
    assert cv.process_image(_photo()).position is None
    assert len(cv._slots) == 1 and [c[2] for c in calls] == [False]
