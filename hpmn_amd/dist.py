"""Data-parallel helpers (one process per GPU, ``torch.distributed``; backend ``nccl`` == RCCL
over xGMI on the GPU box, ``gloo`` in the CPU tests).

The reference has no distributed code; the scheme is build-side
(SURVEY.md section 8e): every sample is independent (no batch statistics: BN runs in inference
mode, code/hpmn.py:190), so a global batch is split into contiguous per-rank slices, each rank
runs forward+BPTT on its slice, ONE sum all-reduce over the flat gradient buffer restores the
single-process gradient, and clip + Adam then run replicated and deterministic on every rank.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as td


def rank_world() -> Tuple[int, int]:
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def shard_bounds(lo: int, hi: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of the global batch [lo, hi) owned by ``rank``."""
    n = hi - lo
    return lo + (n * rank) // world, lo + (n * (rank + 1)) // world


def shard_sizes(n: int, world: int) -> List[int]:
    return [(n * (r + 1)) // world - (n * r) // world for r in range(world)]


def sharded_loss(ll_sum: torch.Tensor, mem_loss: torch.Tensor, global_batch: int, memory_reg: float):
    """Per-rank loss whose gradients SUM (over ranks) to the gradient of code/hpmn.py:202-207 on the
    global batch: the log-loss is a mean over the GLOBAL batch, the memory regulariser a plain sum."""
    return ll_sum / float(global_batch) + memory_reg * mem_loss


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """One collective over the whole flat gradient buffer (dense variables + embedding table)."""
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.all_reduce(flat, op=td.ReduceOp.SUM)
    return flat


def allreduce_sum_async(flat: torch.Tensor):
    """Start a sum all-reduce of a contiguous range; ``.wait()`` on the result orders the current stream
    after it (world > 1 only)."""
    return td.all_reduce(flat, op=td.ReduceOp.SUM, async_op=True)


def chunk_bounds(n: int, chunks: int, align: int = 1) -> List[Tuple[int, int]]:
    """[0, n) cut into at most ``chunks`` contiguous ranges whose interior boundaries are multiples of
    ``align``; never returns an empty range."""
    if n <= 0:
        return []
    step = -(-n // max(1, chunks))
    step = -(-step // align) * align
    return [(a, min(n, a + step)) for a in range(0, n, step)]


def gather_predictions(pred: torch.Tensor, n_global: int) -> torch.Tensor:
    """All-gather per-rank prediction slices (sizes from ``shard_sizes``) into the global order.
    Uses one equal-size all_gather_into_tensor (slices padded to the largest)."""
    rank, world = rank_world()
    if world == 1:
        return pred
    sizes = shard_sizes(n_global, world)
    cap = max(sizes)
    mine = torch.zeros(cap, device=pred.device, dtype=pred.dtype)
    mine[:pred.shape[0]] = pred
    allp = torch.empty(world * cap, device=pred.device, dtype=pred.dtype)
    td.all_gather_into_tensor(allp, mine)
    return torch.cat([allp[r * cap:r * cap + sizes[r]] for r in range(world)])


def reduce_scatter_sum(flat: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Sum over the ranks of ``flat`` (numel divisible by world); returns THIS rank's 1/world slice of the result.
    RCCL: one reduce-scatter; backends without it (gloo): all-reduce, then the local slice."""
    shard = flat.numel() // world
    if td.get_backend() == "nccl":
        out = torch.empty(shard, device=flat.device, dtype=flat.dtype)
        td.reduce_scatter_tensor(out, flat, op=td.ReduceOp.SUM)
        return out
    td.all_reduce(flat, op=td.ReduceOp.SUM)
    return flat[rank * shard:(rank + 1) * shard]


def all_gather_shards_(flat: torch.Tensor, lo: int, shard: int) -> None:
    """Every rank contributes flat[lo:lo+shard] (its own slice); afterwards ``flat`` holds all slices in rank order."""
    mine = flat[lo:lo + shard].clone()
    td.all_gather_into_tensor(flat, mine)
