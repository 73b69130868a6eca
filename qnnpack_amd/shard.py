"""Batch sharding for multi-GPU runs of the hot path (one process per GPU).

Every output pixel of the q8 conv / GEMM / depthwise operators depends on one
image only -- the reference's own tilers treat batch (or batch x pixels) as an
independent grid dimension (src/operator-run.c:675-679, 797-802, 837-842) -- so a
batch splits into contiguous per-rank slices with NO data-path collective: each
rank creates its own operators (weights replicated), runs its slice, and the only
cross-rank traffic is the harness's barrier and max-over-ranks timing.
"""
from __future__ import annotations

from typing import Tuple


def shard_batch(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [start, start + count) of `total` images for `rank`.
    The first `total % world_size` ranks take one extra image."""
    if world_size <= 0 or not (0 <= rank < world_size) or total < 0:
        raise ValueError(f"bad shard request total={total} world_size={world_size} rank={rank}")
    base, extra = divmod(total, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def job_time_ms(local_ms: float, world_size: int) -> float:
    """Whole-job time = slowest rank (all-reduce MAX over ranks). Uses torch.distributed
    when a process group is initialised; identity otherwise."""
    if world_size <= 1:
        return float(local_ms)
    import torch
    import torch.distributed as dist
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(local_ms)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
