"""Template-axis sharding of the bank across the GPUs of one node (SURVEY.md §8 e).

The reference has no collective on its inference path (its DDP shards *queries*,
configs/machine/trainer/local.yaml:9).  Here every rank holds the full weights and all B
query/reference embeddings, generates and scores its own contiguous slice of the N
templates -- the bank slice (16-32 KB per hypothesis) never leaves local HBM -- and the only
exchange is one all-gather of the (B, N/G) f32 scores (<= 1 MiB in total), after which every
rank holds the full (B, N) similarity that `retrieval` returns and the harness saves
(model.py:323,369-375).  One process per GPU, `torch.distributed` backend "nccl" (= RCCL over
xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced split of range(n): first (n % world) ranks get one extra."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_scores(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local (B, n_local) f32 slice of this rank -> (B, n_total) on every rank."""
    rank, ws = world(group)
    if ws == 1:
        assert local.shape[1] == n_total
        return local
    B = local.shape[0]
    nmax = (n_total + ws - 1) // ws
    send = local.new_zeros((B, nmax))
    send[:, : local.shape[1]] = local
    parts = [torch.empty_like(send) for _ in range(ws)]
    dist.all_gather(parts, send.contiguous(), group=group)
    out = local.new_empty((B, n_total))
    for r in range(ws):
        lo, hi = shard_range(n_total, r, ws)
        out[:, lo:hi] = parts[r][:, : hi - lo]
    return out
