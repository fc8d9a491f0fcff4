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


class OverlappedPoseGather:
    """Double-buffered asynchronous all-gather of per-step pose blocks.

    The gather of step i runs on the collective's own stream (RCCL over xGMI) while the kernels of
    step i+1 execute on the compute stream; a local buffer is handed out again only after the
    gather that reads it has completed.  Usage per step i:

        T = g.local(i)        # [n_local, width] tensor the kernels of this step write
        ... launch kernels writing T ...
        g.submit(i)           # enqueue the gather of T, returns immediately
        ...
        all_T = g.result(i)   # [world * n_local, width], waits for that gather only

    With the nccl backend wait() orders streams without blocking the host; with gloo (CPU tests)
    it blocks, which keeps the semantics identical.
    """

    def __init__(self, n_local: int, width: int = 12, dtype=torch.float64, device="cpu", group=None, depth: int = 2):
        self.group = group
        self.world = dist.get_world_size(group)
        self.depth = depth
        self._into_tensor = dist.get_backend(group) != "gloo"
        self._local = [torch.zeros(n_local, width, dtype=dtype, device=device) for _ in range(depth)]
        self._all = [torch.zeros(self.world * n_local, width, dtype=dtype, device=device) for _ in range(depth)]
        self._work = [None] * depth

    def _wait(self, k: int) -> None:
        if self._work[k] is not None:
            self._work[k].wait()
            self._work[k] = None

    def local(self, i: int) -> torch.Tensor:
        k = i % self.depth
        self._wait(k)  # the gather that last read this buffer
        return self._local[k]

    def submit(self, i: int) -> None:
        k = i % self.depth
        if self._into_tensor:
            self._work[k] = dist.all_gather_into_tensor(self._all[k], self._local[k], group=self.group, async_op=True)
        else:
            n = self._local[k].shape[0]
            bufs = list(self._all[k].split(n, dim=0))
            self._work[k] = dist.all_gather(bufs, self._local[k], group=self.group, async_op=True)

    def result(self, i: int) -> torch.Tensor:
        k = i % self.depth
        self._wait(k)
        return self._all[k]

    def drain(self) -> None:
        for k in range(self.depth):
            self._wait(k)
