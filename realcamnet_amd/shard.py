"""Frame sharding across the GPUs of one node.

Frames are independent in eval(), so the path shards by frame with NO collective inside the forward
(one process per GPU, weights replicated).  RCCL (torch.distributed backend "nccl" on ROCm) is used
only for the result gather and the max-over-ranks timing reduce.  The same code runs on gloo/CPU
tensors, which is how the N>1 logic is tested without GPUs.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def frame_shard(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition [start, stop) of n_frames over `world` ranks; the first
    n_frames % world ranks take one extra frame.  Ranks beyond n_frames get an empty range."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError(f"bad shard request n_frames={n_frames} rank={rank} world={world}")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).
    Single-process runs (no WORLD_SIZE) do not create a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """All-reduce(MAX) of a host scalar (the reported wall time is the slowest rank's)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or _default_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def _default_device() -> torch.device:
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_frames(local: torch.Tensor, n_frames: int) -> torch.Tensor:
    """All-gather the per-rank output frames (dim 0) back into global frame order.
    Shards may be uneven (frame_shard); they are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [frame_shard(n_frames, r, world) for r in range(world)]
    mx = max(e - s for s, e in sizes)
    s, e = sizes[rank]
    if local.shape[0] != e - s:
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames, expected {e - s}")
    pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: e - s] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: e_ - s_] for p, (s_, e_) in zip(parts, sizes)], dim=0)
