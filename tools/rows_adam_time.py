"""hpmn_rows_sum_adam alone at the C3 shape: world = 1 (the single-GPU tail) and world = 2 / 4 / 8 lists as a data-parallel
step of 500 sequences per rank would hand them over (uniform item ids over 3.27 M rows + one uid per sequence), against the
dense late pass it replaces (hpmn_adam_step_table pass 1 over a dense gradient table)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpmn_amd import ops

dev = torch.device("cuda:0")
V, E = 3308019, 16
rng = np.random.default_rng(0)
p = torch.randn(V, E, device=dev); m = torch.zeros(V, E, device=dev); v = torch.zeros(V, E, device=dev)


def timed(fn, prep, n=20):
    ts = []
    for _ in range(n):
        prep()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


for world in (1, 2, 4, 8):
    lists = []
    for r in range(world):
        items = rng.integers(39002, V, size=500 * 1001)
        uids = rng.integers(0, 20000, size=500) + 19002
        lists.append(np.unique(np.concatenate([items, uids])).astype(np.int32))
    cap = max(len(x) for x in lists)
    ids_all = torch.full((world, cap), -1, dtype=torch.int32)
    for r, x in enumerate(lists):
        ids_all[r, :len(x)] = torch.as_tensor(x)
    ids_all = ids_all.to(dev)
    counts = torch.tensor([len(x) for x in lists], dtype=torch.int32, device=dev)
    rows_all = torch.randn(world, cap, E, device=dev)
    flags = ops.table_flags(V, dev)
    union = len(np.unique(np.concatenate(lists)))
    bk = ops.RowBuckets(world, V, cap, dev)
    mark = lambda: ops.table_mark_ranks(ids_all, counts, flags, buckets=bk)
    t_mark = timed(mark, lambda: flags.zero_())
    t_nb = timed(lambda: ops.rows_sum_adam(p, m, v, flags, ids_all, rows_all, 1e-3, counts=counts), mark)
    t = timed(lambda: ops.rows_sum_adam(p, m, v, flags, ids_all, rows_all, 1e-3, counts=counts, buckets=bk), mark)
    print("   (without the bucket index: %.1f us; buckets: shift %d, %d per rank)" % (t_nb, bk.shift, bk.nb))
    byt = union * E * 4 * 6 + sum(len(x) for x in lists) * (E * 4 + 4 + 1)
    # the dense form: gradient table + pass 1
    g = torch.zeros(V, E, device=dev)
    t_dense = timed(lambda: ops.adam_step_table(p, g, m, v, flags, 1, 1e-3), mark)
    print("world %d: lists %s, union %d rows; mark %.1f us; rows_sum_adam %.1f us (%.2f TB/s of %.0f MB); "
          "dense pass 1 (flags sweep + marked rows, gradient already summed) %.1f us"
          % (world, [len(x) for x in lists][:3], union, t_mark, t, byt / t / 1e6, byt / 1e6, t_dense), flush=True)
