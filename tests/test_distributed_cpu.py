"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py / the batched API minus the GPU --
one flat-buffer weight broadcast, strided board sharding, rank-major gather and re-interleave."""
from __future__ import annotations

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chessvision import distributed as cvd
from chessvision import synthetic


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    try:
        r, w, device = cvd.init_process_group(backend="gloo")
        assert (r, w) == (rank, world) and device.type == "cpu"
        spec = synthetic.resnet18_spec()
        state = synthetic.resnet18_state_dict(2) if rank == 0 else None
        got = cvd.broadcast_state_dict(state, spec, device)
        want = synthetic.resnet18_state_dict(2)
        same = all(np.array_equal(got[k], want[k]) for k, _, _ in spec)
        # 10 boards sharded r::world; every rank "classifies" its boards (here: tags them with their index)
        n_boards = 10
        mine = list(cvd.shard_indices(n_boards, rank, world))
        local = torch.tensor(mine, dtype=torch.float32).reshape(-1, 1).repeat(1, 13)
        gathered = cvd.all_gather_rows(local)
        ordered = cvd.interleave_shards(gathered, world)
        ok_order = ordered[:, 0].tolist() == list(range(n_boards))
        t = cvd.max_over_ranks(float(rank + 1), device)
        seen = cvd.count_ranks(device)                 # bench.py's `rccl_ranks_seen`: an all-reduce of ones

        class _Eng:                                    # the two calibration calls of HipEngine; rank 1 starts with other exponents
            def __init__(self):
                self.cal = {"unet": np.arange(6, dtype=np.int32) + rank, "resnet18": np.full(4, 3, np.int32)}

            def export_calibration(self, model):
                return self.cal[model]

            def import_calibration(self, model, e):
                changed = not np.array_equal(self.cal[model], e)
                self.cal[model] = np.array(e, np.int32)
                return changed

        eng = _Eng()
        rep = cvd.sync_calibration(eng, device)
        cal_ok = (np.array_equal(eng.cal["unet"], np.arange(6)) and rep["changed_here"] == (rank == 1)
                  and rep["identical_across_ranks"] is False and cvd.sync_calibration(eng, device)["identical_across_ranks"] is True)
        cvd.barrier(device)
        q.put((rank, same and cal_ok, mine, ok_order, t, seen))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_world2_broadcast_shard_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, same0, mine0, ord0, t0, seen0), (r1, same1, mine1, ord1, t1, seen1) = results
    assert seen0 == seen1 == 2
    assert same0 and same1, "broadcast state dict differs from the source"
    assert mine0 == [0, 2, 4, 6, 8] and mine1 == [1, 3, 5, 7, 9]
    assert ord0 and ord1
    assert t0 == t1 == 2.0


def test_single_process_paths_are_identity():
    spec = synthetic.resnet18_spec()
    sd = synthetic.resnet18_state_dict(2)
    out = cvd.broadcast_state_dict(sd, spec, torch.device("cpu"))
    assert all(np.array_equal(out[k], sd[k]) for k in sd)
    flat = cvd.flatten_state(sd, spec)
    assert flat.numel() == 11_176_909 + sum(int(np.prod(s)) for k, s, kind in spec if kind in ("mean", "var"))
    back = cvd.unflatten_state(flat, spec)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
    x = torch.arange(6.0).reshape(3, 2)
    assert torch.equal(cvd.all_gather_rows(x), x)
    assert list(cvd.shard_indices(7, 2, 4)) == [2, 6]
    assert cvd.count_ranks(torch.device("cpu")) == 1


class _FakeVision:
    """Stands in for ChessVision on a CPU-only box: 'classifies' a photo from the board index painted into its first pixel."""

    def process_images(self, images, threshold=0.5, flip=False, fallback_quad=False, timings=None, **kw):
        from chessvision.cv_types import BoardExtractionResult, ChessVisionResult, PositionResult

        out = []
        for im in images:
            idx = int(im[0, 0, 0])
            mask = np.full((256, 256), idx, np.uint8)
            if idx == 3:                                   # a photo without a board
                out.append(ChessVisionResult(BoardExtractionResult(np.zeros((256, 256), np.float32), mask, None, None), None, 0.0))
                continue
            probs = np.full((64, 13), 0.01, np.float32)
            probs[np.arange(64), (np.arange(64) + idx) % 13] = 0.88
            quad = np.arange(8, dtype=np.float32).reshape(4, 1, 2) + idx
            pos = PositionResult(fen=f"local-{idx}", original_fen=f"local-{idx}", model_probabilities=probs, squares=None,
                                 square_names=[], validation_fixes=[])
            out.append(ChessVisionResult(BoardExtractionResult(np.zeros((256, 256), np.float32), mask, quad,
                                                               np.zeros((512, 512), np.uint8)), pos, 0.0))
        return out


def _sharded_worker(rank: int, world: int, port: int, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        cvd.init_process_group(backend="gloo")
        n = 7                                              # odd: the last shard is padded for the collectives
        photos = [np.full((4, 4, 3), i, np.uint8) for i in range(n)]
        tm = {}
        res = cvd.process_images_sharded(_FakeVision(), photos, timings=tm)
        summary = []
        for i, r in enumerate(res):
            be, pos = r.board_extraction, r.position
            summary.append((int(be.binary_mask[0, 0]), None if be.quadrangle is None else float(be.quadrangle[0, 0, 0]),
                            None if pos is None else (int(pos.model_probabilities[0].argmax()), pos.fen.startswith("local"))))
        local_only = cvd.process_images_sharded(_FakeVision(), photos, gather=False)
        q.put((rank, summary, [r is not None for r in local_only], tm["shard_boards"], cvd.host_threads(), len(os.sched_getaffinity(0))))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_world2_process_images_sharded_returns_the_global_batch_in_order():
    """BASELINE configs[4] plumbing: rank r processes photos r::2, probabilities / quadrangles / masks are gathered rank-major
    and re-interleaved, remote FENs are re-derived by the native decoder -- every rank ends with all 7 results in order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    cpus_before = len(os.sched_getaffinity(0))
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, summary, local_mask, shard_boards, threads, cpus in got:
        assert [s[0] for s in summary] == list(range(7))                      # masks in board order
        assert [s[1] for s in summary] == [0.0, 1.0, 2.0, None, 4.0, 5.0, 6.0]  # quadrangles; board 3 has none
        for i, s in enumerate(summary):
            if i == 3:
                assert s[2] is None
            else:
                assert s[2][0] == i % 13                                      # probabilities of board i
                assert s[2][1] == (i % 2 == rank)                             # own boards keep the local object, others are rebuilt
        assert local_mask == [i % 2 == rank for i in range(7)]
        assert shard_boards == (4 if rank == 0 else 3)
        if cpus_before >= 2:                                                  # every rank was pinned to its half of the CPUs
            assert cpus == cpus_before // 2 and threads == min(32, cpus)


def test_host_threads_divide_the_cpus_among_local_ranks(monkeypatch):
    cpus = len(os.sched_getaffinity(0))
    monkeypatch.setattr(cvd, "_PINNED", False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert cvd.host_threads() == max(1, min(32, cpus // 8))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert cvd.host_threads(cap=4) == min(4, cpus)
    # a launcher that exports WORLD_SIZE only (multi-node): the share is sized by the GPUs of THIS host, never by the whole job;
    # without GPUs (this test box) the world size is the only bound left
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.setenv("WORLD_SIZE", "64")
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    assert cvd.local_world_size() == (min(64, n_dev) if n_dev else 64)
    # the "pinned" flag is module state: a child process (which inherits the environment, not the module) divides again
    assert "CV_RANK_CPUS_PINNED" not in os.environ
