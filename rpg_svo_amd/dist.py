"""Multi-GPU plumbing: the path shards across independent (reference, current)
problems -- frames of a replay, cameras of a rig -- one rank per GPU, and the only
exchange is a gather of the resulting SE(3) poses (RCCL over xGMI on the GPU node;
any torch.distributed backend works, gloo is used in the CPU tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of problems 0..n_total-1 (sizes differ by <= 1)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_poses(T_local: torch.Tensor, counts: list[int] | None = None, group=None) -> torch.Tensor:
    """All-gather [n_r, 12] pose blocks into [sum n_r, 12] in rank order.  Equal
    shard sizes use one all_gather_into_tensor; ragged shards are padded."""
    world = dist.get_world_size(group)
    n = T_local.shape[0]
    if counts is None:
        counts = [n] * world
    nmax = max(counts)
    if all(c == nmax for c in counts) and dist.get_backend(group) != "gloo":
        out = torch.empty(world * nmax, T_local.shape[1], dtype=T_local.dtype, device=T_local.device)
        dist.all_gather_into_tensor(out, T_local.contiguous(), group=group)
        return out
    pad = torch.zeros(nmax, T_local.shape[1], dtype=T_local.dtype, device=T_local.device)
    pad[:n] = T_local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
