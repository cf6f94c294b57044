"""Symbol sharding across the GPUs of one node.

The path has no cross-symbol dependency (independent capital pools): rank r of G owns the contiguous symbol range
[floor(N*r/G), floor(N*(r+1)/G)) -- with the symbol-major layout a contiguous byte range of every column -- and
computes on it with no communication.  The only exchange is the per-symbol summary table ([n, 8] f64, 64 B per
symbol): one all_gather over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_symbols: int, rank: int, world: int) -> tuple[int, int]:
    return (n_symbols * rank) // world, (n_symbols * (rank + 1)) // world


def gather_summaries(local: torch.Tensor, n_symbols: int, group=None) -> torch.Tensor:
    """local: [n_local, 8] summary rows of this rank's shard -> [n_symbols, 8] on every rank (symbol order)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(n_symbols, r, world)[1] - shard_range(n_symbols, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((n_symbols, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged shards: pad to the largest, gather, trim
    m = max(sizes)
    pad = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)], dim=0)
