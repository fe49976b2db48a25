"""Multi-GPU plumbing: one process per GPU, boards sharded, weights replicated by one RCCL broadcast.

The reference has no distributed code at all.  The hot path shards naturally -- boards are
independent, BatchNorm is in eval mode -- so the only collective on the path is the one-time broadcast of the
flattened state dicts from rank 0 over xGMI (~170 MB); steady-state inference exchanges nothing.  Results can
optionally be all-gathered (68 KB / board).  Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo"
on CPU (used by the world_size-2 tests).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Mapping, Sequence

import numpy as np
import torch
import torch.distributed as dist


def env_world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: str | None = None) -> tuple[int, int, torch.device]:
    """Initialise torch.distributed from env when WORLD_SIZE > 1; returns (rank, world, device)."""
    rank, world, local = env_world()
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    force = os.environ.get("CV_FORCE_DIST", "0") == "1"          # exercise the RCCL path on one GPU (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kwargs = {}
        if use_gpu:
            kwargs["device_id"] = device
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_indices(n_items: int, rank: int, world: int) -> range:
    """Strided shard: rank r owns items r, r+world, r+2*world, ...."""
    return range(rank, n_items, world)


def flatten_state(state: Mapping[str, np.ndarray | torch.Tensor], spec: Sequence[tuple]) -> torch.Tensor:
    parts = []
    for key, shape, _ in spec:
        v = state[key]
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) if isinstance(v, np.ndarray) else v.detach().float().cpu()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {tuple(t.shape)} != spec {tuple(shape)}")
        parts.append(t.reshape(-1))
    return torch.cat(parts)


def unflatten_state(flat: torch.Tensor, spec: Sequence[tuple]) -> "OrderedDict[str, np.ndarray]":
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    flat = flat.detach().cpu().numpy()
    off = 0
    for key, shape, _ in spec:
        n = int(np.prod(shape))
        out[key] = flat[off:off + n].reshape(shape).copy()
        off += n
    if off != flat.size:
        raise ValueError("flat state size does not match the spec")
    return out


def broadcast_state_dict(state: Mapping[str, np.ndarray] | None, spec: Sequence[tuple], device: torch.device,
                         src: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Rank `src` supplies `state`; every rank returns an identical copy (one flat-buffer broadcast)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not dist.is_initialized():
        assert state is not None
        return OrderedDict((k, np.ascontiguousarray(state[k], dtype=np.float32)) for k, _, _ in spec)
    total = sum(int(np.prod(s)) for _, s, _ in spec)
    rank = dist.get_rank()
    if rank == src:
        assert state is not None, "source rank must provide the state dict"
        flat = flatten_state(state, spec).to(device)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    return unflatten_state(flat, spec)


def all_gather_rows(local: torch.Tensor) -> torch.Tensor:
    """Concatenate equally-shaped per-rank result tensors along dim 0, rank-major."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local.contiguous())
    return torch.cat(out, dim=0)


def interleave_shards(gathered: torch.Tensor, world: int) -> torch.Tensor:
    """Undo the strided sharding: rows [rank-major] -> original board order (equal shard sizes)."""
    per = gathered.shape[0] // world
    return gathered.reshape(world, per, *gathered.shape[1:]).transpose(0, 1).reshape(per * world, *gathered.shape[1:])


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def count_ranks(device: torch.device) -> int:
    """All-reduce of a one per rank: the number of ranks that really took part in a collective (1 without a process group)."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def shutdown() -> None:
    """Tear the process group down (quietly a no-op for single-process runs)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def barrier(device: torch.device) -> None:
    if dist.is_initialized():
        if device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
