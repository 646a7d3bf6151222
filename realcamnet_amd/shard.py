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
    Single-process runs (no WORLD_SIZE / MASTER_ADDR in the environment) do not create a process group; a 1-rank torchrun job
    does, so the collective path can be exercised on a 1-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    launched = "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ      # under torchrun even a 1-rank job gets a group
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def torchrun_command(script: str, argv, nproc: int, port: Optional[int] = None):
    """The command that runs `script argv...` as `nproc` ranks of one node under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1:
    the container hostname may not resolve).  `port` None = a free port picked here."""
    import socket
    import sys
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script, *argv]


def relaunch_under_torchrun(script: str, argv, nproc: int) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: replace this process by the N-rank torchrun job (never returns).
    The reference's only multi-GPU hook is a one-process nn.DataParallel (upstream models/networks.py:99-106): one command, no launcher -- this keeps that
    calling convention for the one-process-per-GPU path.  Under a launcher (WORLD_SIZE set) this must not be called."""
    if "WORLD_SIZE" in os.environ:
        raise RuntimeError("relaunch_under_torchrun: already under a launcher")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = torchrun_command(script, list(argv), nproc)
    os.execv(cmd[0], cmd)


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


class OverlappedGather:
    """All-gather of each step's output frames to every rank, overlapped with the NEXT step's forward (the frames
    shard with no collective inside the forward; the consumer-side gather of 8 x 3 x 2160 x 3840 bf16 = 398 MB per GPU is
    ~2.6 ms over xGMI's direct links against >= 50 ms of compute, so it hides behind the following forward).

    submit(local): on a CUDA tensor the collective is enqueued on a side stream that waits (event) for `local` to be complete on
    the caller's stream; the caller's stream is never blocked.  wait(): the caller's stream waits for the most recent gather and
    its result (global frame order, (n_frames, ...)) is returned.  Equal shards only (n_frames % world == 0: one
    all_gather_into_tensor, no padding); CPU tensors (gloo) take the same path without streams -- that is how the logic is
    tested without GPUs."""

    def __init__(self, n_frames: int):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("OverlappedGather needs an initialised process group (shard.init_distributed under torchrun)")
        self.world = dist.get_world_size()
        if n_frames % self.world:
            raise ValueError(f"{n_frames} frames do not split evenly over {self.world} ranks")
        self.n_frames = n_frames
        self._stream = None
        self._bufs = [None, None]        # double buffer: gather k+1 may start while the consumer still reads gather k
        self._turn = 0
        self._last = None
        self._work = None

    def submit(self, local: torch.Tensor) -> None:
        per = self.n_frames // self.world
        if local.shape[0] != per:
            raise ValueError(f"rank holds {local.shape[0]} frames, expected {per}")
        local = local.contiguous()
        i = self._turn
        self._turn ^= 1
        buf = self._bufs[i]
        if buf is None or buf.shape[1:] != local.shape[1:] or buf.dtype != local.dtype or buf.device != local.device:
            buf = self._bufs[i] = torch.empty((self.n_frames, *local.shape[1:]), dtype=local.dtype, device=local.device)
        if local.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=local.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(local.device))
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ready)
                local.record_stream(self._stream)          # the allocator must not recycle `local` under the collective
                dist.all_gather_into_tensor(buf, local)    # RCCL: enqueued behind the side stream, returns at once
        else:
            if self._work is not None:
                self._work.wait()
            self._work = dist.all_gather_into_tensor(buf, local, async_op=True)
        self._last = buf

    def wait(self) -> torch.Tensor:
        if self._last is None:
            raise RuntimeError("OverlappedGather.wait() before submit()")
        if self._last.is_cuda:
            torch.cuda.current_stream(self._last.device).wait_stream(self._stream)
        elif self._work is not None:
            self._work.wait()
            self._work = None
        return self._last
