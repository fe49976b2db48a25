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
        cvd.barrier(device)
        q.put((rank, same, mine, ok_order, t, seen))
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
