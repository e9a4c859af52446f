"""Image-sharded multi-GPU evaluation helpers.

The path has no cross-image state, so N GPUs = N replicas that each take a contiguous slice of the images
(`keypoints/bin/eval.py:18-49` loops over a dataset sequentially); there is no collective on the data path.
Only the tiny per-image result lists travel to rank 0 at the end.
"""
from __future__ import annotations

import torch.distributed as dist


def shard_range(num_items: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced slice of `range(num_items)` for `rank` (first `num_items % world_size` ranks get one more)."""
    base, extra = divmod(num_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_results(local: list, dst: int = 0) -> list | None:
    """Concatenate per-rank result lists on `dst` in rank order (== dataset order with `shard_range`)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    if out is None:
        return None
    return [r for part in out for r in part]
