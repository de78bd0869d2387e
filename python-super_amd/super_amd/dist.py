"""Frame / hypothesis sharding across the GPUs of one node.

Frames are independent LM problems, so the data path has NO collective: rank r owns a
contiguous block of the global frame list, solves it locally, and the only exchange is the
end-of-frame all-gather of the solved warps beta (J*7 float64 per frame) over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  One process per GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_frames: int, world: int, rank: int):
    """Contiguous, balanced block [lo, hi) of the global frame list owned by `rank`
    (the first n_frames % world ranks get one extra frame)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def all_gather_betas(local_betas: torch.Tensor, n_frames: int, group=None) -> torch.Tensor:
    """Gather per-rank (n_local, J, 7) float64 solutions into the global (n_frames, J, 7)
    tensor, in global frame order, on every rank.  Ragged shards (n_frames not divisible by
    the world size) are padded to the largest shard for the collective and trimmed after."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_frames, world, rank)
    if local_betas.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} owns {hi - lo} frames, got {local_betas.shape[0]}")
    n_max = -(-n_frames // world)
    J = local_betas.shape[1]
    pad = torch.zeros((n_max, J, 7), dtype=local_betas.dtype, device=local_betas.device)
    pad[: hi - lo] = local_betas
    out = torch.empty((world * n_max, J, 7), dtype=local_betas.dtype, device=local_betas.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(n_frames, world, r)
        parts.append(out[r * n_max: r * n_max + (b - a)])
    return torch.cat(parts, dim=0)


class _DeviceView:
    """A raw device buffer owned by libsuper_lm.so, exposed through ``__cuda_array_interface__`` so that
    ``torch.as_tensor`` aliases it (no copy): the collective then runs in place on the library's memory."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_view(ptr: int, n: int, device) -> torch.Tensor:
    return torch.as_tensor(_DeviceView(ptr, n), device=device)


def default_collectives(group=None):
    """(all_reduce_sum, broadcast_from_0) for float64 device tensors over ``torch.distributed``.  Backend "nccl"
    (RCCL over xGMI on the GPU box) works on the device tensor in place; "gloo" (CPU tests, or several ranks
    sharing one GPU) stages through host memory."""
    if dist.get_backend(group) == "gloo":
        def all_reduce(t):
            h = t.cpu()
            dist.all_reduce(h, group=group)
            t.copy_(h)

        def broadcast(t):
            h = t.cpu()
            dist.broadcast(h, src=0, group=group)
            t.copy_(h)
        return all_reduce, broadcast
    return (lambda t: dist.all_reduce(t, group=group)), (lambda t: dist.broadcast(t, src=0, group=group))
