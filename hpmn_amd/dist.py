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


# Runtime settings of a data-parallel process (r6; through r5 they lived in bench.py only).  The data-parallel step has seven
# streams (launch, auxiliary, two helpers, the plan's, the communicators') and the HIP runtime four hardware queues by default:
# streams that share a queue serialise.  Measured with one rank on RCCL in bench.py's timed loop (r5, rows exchange, C3):
# 4 / 5 / 6 / 8 queues -> 3.01 / 2.63 / 3.14 / 3.41 ms per step.  Both variables are read ONCE, at the first HIP call.
DP_RUNTIME_ENV = {"GPU_MAX_HW_QUEUES": "5", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def apply_runtime_env(force: bool = False) -> dict:
    """Set DP_RUNTIME_ENV (defaults only: a caller's own values win) when this process is one rank of a multi-process job
    (WORLD_SIZE > 1) or ``force``.  Called by ``import hpmn_amd`` and by init_data_parallel(); returns what it set.  Warns when
    HIP is already initialised and a value would have changed (too late to take effect)."""
    import os
    import warnings
    if not force and int(os.environ.get("WORLD_SIZE", "1") or 1) <= 1:
        return {}
    done = {}
    late = torch.cuda.is_available() and torch.cuda.is_initialized()
    for k, v in DP_RUNTIME_ENV.items():
        if k not in os.environ:
            if late:
                warnings.warn("hpmn_amd.dist: %s is unset and HIP is already initialised -- import hpmn_amd (or call "
                              "dist.apply_runtime_env()) before the first CUDA/HIP call; the data-parallel step is ~14 %% "
                              "slower on the runtime's default of four hardware queues" % k)
            os.environ[k] = v
            done[k] = v
    return done


def init_data_parallel(backend: str = "nccl"):
    """One rank of a one-node data-parallel job launched by ``python -m torch.distributed.run --nproc-per-node N
    --master-addr 127.0.0.1 ...``: runtime settings, device, process group (RCCL over xGMI; ``gloo`` for dry runs in which
    ranks share a device).  Returns (rank, world, device).  A single process (no WORLD_SIZE) gets (0, 1, cuda:0) and no group."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    rank = int(os.environ.get("RANK", "0") or 0)
    local = int(os.environ.get("LOCAL_RANK", "0") or 0)
    apply_runtime_env()
    if backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not (td.is_available() and td.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            td.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, device


def rank_world() -> Tuple[int, int]:
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def forced() -> bool:
    """HPMN_DP_FORCE_COLLECTIVES=1 inside an initialised process group: no world-size-1 short cuts, every collective is
    issued (a 1-GPU box runs the RCCL calls of the data-parallel step this way)."""
    import os
    return bool(os.environ.get("HPMN_DP_FORCE_COLLECTIVES") == "1" and td.is_available() and td.is_initialized())


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
    if td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or forced()):
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
    if world == 1 and not forced():
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


# ---------------------------------------------------------------------------------------------------------------------
# Touched-rows exchange of the embedding-table gradient ("large tables -> allGather of (unique row ids,
# summed grad rows) per rank, then local scatter-add"; the gradient of code/hpmn.py:421-422 is an IndexedSlices over the
# batch's rows before :204-205 densifies it).  A step only ever touches the rows its ids name, so for a table sized to
# HBM the dense all-reduce (2 (N-1)/N V E 4 bytes per rank) is replaced by an all-gather of the touched rows.

def gather_ids(ids: torch.Tensor, cap: int) -> torch.Tensor:
    """All ranks' id tensors as one [world, cap] tensor of the ids' own width (int32, or int64 for tables beyond 2^31 - 1
    rows), -1 where a rank had fewer than ``cap`` entries (a short last batch).  No host synchronisation: ``cap`` comes from
    the batch geometry (``shard_sizes``)."""
    _, world = rank_world()
    idt = torch.int64 if ids.dtype == torch.int64 else torch.int32
    flat = ids.reshape(-1).to(idt)
    assert flat.numel() <= cap
    mine = torch.full((cap,), -1, device=ids.device, dtype=idt)
    mine[:flat.numel()] = flat
    if world == 1 and not forced():
        return mine.view(1, cap)
    out = torch.empty(world * cap, device=ids.device, dtype=idt)
    td.all_gather_into_tensor(out, mine)
    return out.view(world, cap)


def exchange_counts(n: int, device) -> List[int]:
    """Every rank's ``n`` (one small collective + one host read: the row lists of the next call are sized by it)."""
    _, world = rank_world()
    if world == 1 and not forced():
        return [int(n)]
    mine = torch.tensor([int(n)], device=device, dtype=torch.int64)
    out = torch.empty(world, device=device, dtype=torch.int64)
    td.all_gather_into_tensor(out, mine)
    return [int(x) for x in out.tolist()]


class _Counts:
    """Every rank's count(s), on their way to pinned host memory: ``result()`` waits for the copy (an event, not a device
    synchronisation) and returns the list -- one int per rank, or one list per rank for a vector of counts."""

    def __init__(self, host, event, fallback=None, width=1):
        self._host, self._event, self._list, self._width = host, event, fallback, width

    def result(self):
        if self._list is None:
            self._event.synchronize()
            flat = [int(x) for x in self._host.tolist()]
            w = self._width
            self._list = flat if w == 1 else [flat[i:i + w] for i in range(0, len(flat), w)]
        return self._list


def exchange_counts_async(n, device) -> "_Counts":
    """``n``: an int, or a device tensor of one or several counts.  exchange_counts whose host read is deferred: the
    all-gather and a non-blocking copy into pinned memory are enqueued on the CURRENT stream now, the caller asks for
    ``result()`` when it needs the numbers (data-parallel step: enqueued at the start of the step underneath the forward, read
    when the row exchange is sized -- long after the copy completed, so the host never waits on the device's critical
    path).  A vector of C counts per rank comes back as ``[world][C]``."""
    _, world = rank_world()
    on_device = isinstance(n, torch.Tensor)              # (a device-side count: it is never read on the host here)
    if world == 1 and not forced() and not on_device:
        return _Counts(None, None, [int(n)])
    mine = n.reshape(-1).to(torch.int64) if on_device else torch.tensor([int(n)], device=device, dtype=torch.int64)
    width = int(mine.numel())
    if world == 1 and not forced():
        out = mine
    else:
        out = torch.empty(world * width, device=device, dtype=torch.int64)
        td.all_gather_into_tensor(out, mine.contiguous())
    if out.is_cuda:
        host = torch.empty(out.numel(), dtype=torch.int64, pin_memory=True)
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return _Counts(host, ev, width=width)
    flat = [int(x) for x in out.tolist()]
    return _Counts(None, None, flat if width == 1 else [flat[i:i + width] for i in range(0, len(flat), width)], width)


def exchange_rows(rows: torch.Tensor, grads: torch.Tensor, counts: List[int], wide_ids: bool = False, async_op: bool = False):
    """All-gather of (row ids [n] -- int32 on the wire, int64 with ``wide_ids`` (tables beyond 2^31 - 1 rows; the same on every
    rank: it follows from the table size) --, gradient rows [n, E]) over the ranks, padded to the largest count.
    Returns (ids [world, cap] with -1 padding, grads [world, cap, E]); rank r's valid entries are the first counts[r].
    ``async_op``: the two collectives are only STARTED; a third return value holds their handles (``.wait()`` orders the
    current stream behind them) -- several exchanges in flight while the caller consumes the first."""
    rank, world = rank_world()
    cap = max(1, max(counts))
    E = grads.shape[1]
    n = rows.numel()
    assert n == counts[rank] and grads.shape[0] == n
    idt = torch.int64 if wide_ids else torch.int32
    ids_mine = torch.full((cap,), -1, device=rows.device, dtype=idt)
    ids_mine[:n] = rows.to(idt)
    g_mine = torch.zeros(cap, E, device=grads.device, dtype=grads.dtype)
    g_mine[:n] = grads
    if world == 1 and not forced():
        return (ids_mine.view(1, cap), g_mine.view(1, cap, E)) + (([],) if async_op else ())
    ids_all = torch.empty(world * cap, device=rows.device, dtype=idt)
    g_all = torch.empty(world * cap * E, device=grads.device, dtype=grads.dtype)
    if async_op:
        works = [td.all_gather_into_tensor(ids_all, ids_mine, async_op=True),
                 td.all_gather_into_tensor(g_all, g_mine.view(-1), async_op=True)]
        return ids_all.view(world, cap), g_all.view(world, cap, E), works
    td.all_gather_into_tensor(ids_all, ids_mine)
    td.all_gather_into_tensor(g_all, g_mine.view(-1))
    return ids_all.view(world, cap), g_all.view(world, cap, E)


def sum_rows_into_(dst: torch.Tensor, ids_all: torch.Tensor, g_all: torch.Tensor, counts: List[int], row_of=None) -> None:
    """dst[row_of(id)] += every rank's gradient rows, in RANK ORDER 0..world-1 on every rank (so that the replicas add
    the same numbers in the same order and stay bit-identical).  ``dst`` must hold zeros in the rows named (the caller's
    own contribution arrives through ``g_all`` like everybody else's).  ``row_of`` maps table ids to rows of ``dst``
    (identity for the dense [V, E] gradient, a searchsorted into the union for a compact buffer)."""
    for r, n in enumerate(counts):
        if n == 0:
            continue
        idx = ids_all[r, :n].long()
        if row_of is not None:
            idx = row_of(idx)
        dst.index_add_(0, idx, g_all[r, :n])


def rows_exchange_bytes(counts: List[int], E: int, wide_ids: bool = False) -> int:
    """Bytes one rank RECEIVES in exchange_rows (ids + rows of every other rank, padded to the cap)."""
    world = len(counts)
    return (world - 1) * max(1, max(counts)) * ((8 if wide_ids else 4) + 4 * E)


def dense_allreduce_bytes(numel: int, world: int) -> int:
    """Bytes one rank sends (= receives) in a ring all-reduce of ``numel`` fp32 values."""
    return int(2 * (world - 1) / world * numel * 4) if world > 1 else 0


# ---------------------------------------------------------------------------------------------------------------------
# r5: the rows exchange without framework kernels around it.  The deterministic scatter's plan holds a rank's
# distinct rows (ascending) in a buffer of the batch geometry's capacity; that buffer and a small vector of counts are
# all-gathered AS THEY ARE at the start of the step (underneath the forward), the compact gradient rows chunk by chunk behind
# BPTT -- slices of the plan's own buffer, padding never read -- and hpmn_rows_sum_adam consumes the gathered buffers directly:
# no padded copies, no index_fill_ / index_add_, no dense gradient table.

def all_gather_fixed(mine: torch.Tensor, async_op: bool = False, group=None):
    """[world, *mine.shape]: every rank's ``mine`` (same shape and dtype everywhere).  One rank without forced collectives:
    a view, no copy.  ``async_op``: also returns the handle (``.wait()`` orders the current stream behind the collective).
    ``group``: a second process group = a second communicator with its own stream (side_group())."""
    _, world = rank_world()
    mine = mine.contiguous()
    if world == 1 and not forced():
        out = mine.view((1,) + tuple(mine.shape))
        return (out, None) if async_op else out
    out = torch.empty((world,) + tuple(mine.shape), device=mine.device, dtype=mine.dtype)
    if async_op:
        return out, td.all_gather_into_tensor(out.view(-1), mine.view(-1), async_op=True, group=group)
    td.all_gather_into_tensor(out.view(-1), mine.view(-1), group=group)
    return out


_side_group = None


def side_group():
    """A second process group over all ranks, created once (every rank must reach this at the same point of its program):
    collectives on it use their own communicator and stream, so an exchange prepared a step ahead never queues in front of
    the collectives of the step that is running."""
    global _side_group
    if _side_group is None and td.is_available() and td.is_initialized():
        try:
            _side_group = td.new_group()
        except Exception as e:                     # (a backend that cannot make a second communicator: the main one serves)
            import warnings
            warnings.warn("hpmn_amd.dist: no second process group (%s: %s); the step-ahead exchange uses the default one"
                          % (type(e).__name__, e))
            _side_group = False
    return _side_group or None


class HostCopy:
    """A small device tensor on its way to pinned host memory (non-blocking copy + event on the current stream):
    ``result()`` waits for the EVENT (not the device) and returns a nested list."""

    def __init__(self, t: torch.Tensor):
        self._shape = tuple(t.shape)
        if t.is_cuda:
            self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._host.copy_(t, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = t.clone(), None

    def result(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return self._host.tolist()


def rows_windows(counts2d: List[List[int]]):
    """From every rank's [total, c_0 .. c_{C-1}] (distinct rows, and per chunk of the table's row range): the list lengths,
    and per chunk c the window (first[r], n[r]) of every rank's list plus the rows every rank sends (the largest n)."""
    lens = [int(c[0]) for c in counts2d]
    C = len(counts2d[0]) - 1
    first = [0] * len(counts2d)
    windows = []
    for c in range(C):
        n = [int(x[1 + c]) for x in counts2d]
        windows.append((list(first), n, max(n)))
        first = [a + b for a, b in zip(first, n)]
    assert first == lens, "chunk counts do not add up to the list lengths"
    return lens, windows


def rows_exchange_bytes_windows(windows, E: int, wide_ids: bool, world: int, cap_ids: int) -> int:
    """Bytes one rank RECEIVES in the r5 exchange: the early id lists (capacity-sized) + the gradient rows of every chunk."""
    return (world - 1) * (cap_ids * (8 if wide_ids else 4) + sum(w[2] for w in windows) * 4 * E)
