"""hpmn_rows_sum_adam / hpmn_table_mark_ranks (ABI v12, csrc/rows_adam.hip) through the C ABI: the update of the touched
table rows from every rank's compact gradient rows must be, BIT FOR BIT, the dense path it replaces -- the ranks' rows added
in rank order into a dense gradient table (what dist.sum_rows_into_ did with index_add_), then clip + dense TF-form Adam
(code/hpmn.py:204-214) -- with no dense gradient table anywhere.  Integer/bit-exact bar: torch.equal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from hpmn_amd import _lib, build
    build.build_library()
    _lib.load()
    return torch.device("cuda:0")


def _lists(rng, world, V, lo, hi, idt, hot=0):
    """Per rank: ascending distinct rows (a few `hot` rows held by every rank, the rest random), ragged lengths."""
    out = []
    for r in range(world):
        n = int(rng.integers(lo, hi + 1))
        rows = rng.choice(V, size=min(n, V), replace=False)
        if hot:
            rows = np.union1d(rows, np.arange(hot) * 7 % V)
        out.append(np.sort(np.unique(rows)).astype(idt))
    return out


def _dense_reference(p, m, v, lists, grads, lr_t, dev):
    """Rank-ordered dense sum, then the DENSE one-sweep hpmn_adam_step over the whole table."""
    from hpmn_amd import ops
    V, E = p.shape
    g = torch.zeros(V, E, device=dev)
    for rows, gr in zip(lists, grads):                       # rank order 0..world-1, one rank at a time: fixed addend order
        g.index_add_(0, torch.as_tensor(rows.astype(np.int64), device=dev), gr)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    ops.adam_step(p2.view(-1), g.view(-1), m2.view(-1), v2.view(-1), lr_t, clip=1.0)
    return p2, m2, v2


@pytest.mark.parametrize("bucketed", [False, True])
@pytest.mark.parametrize("world,E,idt,chunks", [(1, 16, np.int32, 1), (2, 16, np.int32, 1), (3, 16, np.int64, 3),
                                                 (8, 16, np.int32, 4), (8, 4, np.int32, 2), (5, 64, np.int64, 1)])
def test_rows_sum_adam_is_the_dense_two_pass_update_bit_for_bit(dev, world, E, idt, chunks, bucketed):
    from hpmn_amd import ops
    rng = np.random.default_rng(1000 * world + E + chunks)
    V = 20011
    lists = _lists(rng, world, V, 50, 6000, idt, hot=37)
    cap = max(len(x) for x in lists) + 5
    tdt = torch.int64 if idt == np.int64 else torch.int32
    ids_all = torch.full((world, cap), -1, dtype=tdt)
    for r, rows in enumerate(lists):
        ids_all[r, :len(rows)] = torch.as_tensor(rows)
    ids_all = ids_all.to(dev)
    counts = torch.tensor([len(x) for x in lists], dtype=torch.int32, device=dev)
    grads = [torch.as_tensor(rng.standard_normal((len(x), E)).astype(np.float32) * 0.7, device=dev) for x in lists]
    p = torch.as_tensor(rng.standard_normal((V, E)).astype(np.float32), device=dev)
    m = torch.as_tensor(rng.standard_normal((V, E)).astype(np.float32) * 0.1, device=dev)
    v = torch.as_tensor(rng.random((V, E)).astype(np.float32) * 0.01, device=dev)
    lr_t = 0.0017
    want = _dense_reference(p, m, v, lists, grads, lr_t, dev)

    flags = ops.table_flags(V, dev)
    # (bucketed: the marking pass also builds the bucket index the late launch searches through -- a tiny capacity here, so that
    #  buckets hold many entries, some none, and the lists of a few ranks end far below the last bucket)
    bk = ops.RowBuckets(world, V, 2048 if world % 2 else 64, dev) if bucketed else None
    ops.table_mark_ranks(ids_all, counts, flags, buckets=bk)
    torch.cuda.synchronize()
    if bk is not None:
        st = bk.start.cpu().numpy()
        for r, rows in enumerate(lists):
            want_st = np.searchsorted(rows >> bk.shift, np.arange(bk.nb + 1), side="left")
            assert (st[r, :bk.nb + 1] == want_st).all()
    f = flags.cpu().numpy()
    mask = np.zeros(V, np.uint8)
    for r, rows in enumerate(lists):
        mask[rows] |= 1 << r
    assert (f == mask).all()
    # pass 0: the rows nobody touches (gradient taken as zero, never read: no gradient table is passed)
    ops.adam_step_table(p, None, m, v, flags, 0, lr_t, clip=1.0)
    # the touched rows, chunk of the table's row range by chunk (what the chunked exchange hands over)
    bounds = [(V * k) // chunks for k in range(chunks + 1)]
    for c in range(chunks):
        first = [int(np.searchsorted(x, bounds[c])) for x in lists]
        n = [int(np.searchsorted(x, bounds[c + 1])) - a for x, a in zip(lists, first)]
        capc = max(max(n), 1)
        rows_all = torch.full((world, capc, E), float("nan"), device=dev)      # (padding is never read)
        for r in range(world):
            rows_all[r, :n[r]] = grads[r][first[r]:first[r] + n[r]]
        if c % 2 == 0:
            ops.rows_sum_adam(p, m, v, flags, ids_all, rows_all, lr_t, counts=counts, first=first, n=n, buckets=bk)
        else:                                                                   # host-side lengths instead of device counts
            ops.rows_sum_adam(p, m, v, flags, ids_all, rows_all, lr_t, lens=[len(x) for x in lists], first=first, n=n,
                              buckets=bk)
    torch.cuda.synchronize()
    assert torch.equal(p, want[0]) and torch.equal(m, want[1]) and torch.equal(v, want[2])
    assert int(flags.count_nonzero()) == 0                                      # left all-zero for the next step


def test_rows_sum_adam_single_rank_takes_device_count_and_scatter_rows(dev):
    """world == 1, the single-GPU tail: list and gradient rows straight from the deterministic scatter's plan, the count
    only on the device, the grid sized by the capacity (entries past the count leave)."""
    from hpmn_amd import ops
    rng = np.random.default_rng(5)
    V, E, B, T, F = 3001, 16, 6, 40, 2
    ids = torch.as_tensor(rng.integers(0, V, size=(B, T, F)).astype(np.int32), device=dev)
    ids[:, :, 0] = ids[:, :1, 0]                                               # a constant uid column: long runs
    d_x = torch.as_tensor(rng.standard_normal((B, T, F * E)).astype(np.float32), device=dev)
    plan = ops.ScatterPlan(ids, E, want_rows=True)
    ops.embed_grad_segsum(plan, (B, T, F), d_x, None, 0, False)
    p = torch.as_tensor(rng.standard_normal((V, E)).astype(np.float32), device=dev)
    m = torch.zeros(V, E, device=dev)
    v = torch.zeros(V, E, device=dev)
    # reference: the same plan into a dense gradient table, then the dense sweep
    g = torch.zeros(V, E, device=dev)
    ops.embed_grad_segsum(plan, (B, T, F), d_x, g, 0, False)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    ops.adam_step(p2.view(-1), g.view(-1), m2.view(-1), v2.view(-1), 0.003, clip=1.0)
    flags = ops.table_flags(V, dev)
    ops.table_mark_rows(ids, flags)
    ops.adam_step_table(p, None, m, v, flags, 0, 0.003, clip=1.0)
    ops.rows_sum_adam(p, m, v, flags, plan.rows.view(1, -1), plan.out_rows.view(1, -1, E), 0.003, counts=plan.count)
    torch.cuda.synchronize()
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)
    assert int(flags.count_nonzero()) == 0


@pytest.mark.parametrize("wide", [False, True])
def test_plan_built_by_the_library_equals_the_plan_built_from_framework_ops(dev, monkeypatch, wide):
    """hpmn_scatter_plan_build (radix sort of (id, lookup) pairs + scan + plan kernel + per-chunk counts, csrc/plan_build.hip)
    against the r4 construction out of torch.sort(stable=True) / cumsum / searchsorted: perm, seg, start, rows, count and
    the distinct rows per chunk of the table's row range -- integers, so equal."""
    from hpmn_amd import ops
    rng = np.random.default_rng(77)
    V = 2_500_000_000 if wide else 70_001
    ids = rng.integers(0, V, size=(9, 113, 3)).astype(np.int64 if wide else np.int32)
    ids[:, :, 0] = ids[:, :1, 0]                                     # long runs
    ids[2:4, 40:, 1] = 0                                             # the padding id, many times
    t = torch.as_tensor(ids).to(dev)
    bounds = [0, V // 3, V // 3 + 5, V - 1, V]
    plans = []
    for fast in (True, False):
        monkeypatch.setattr(ops, "FAST_PLAN", fast)
        plans.append(ops.ScatterPlan(t, 16, want_rows=True, row_bounds=bounds, rows_capacity=4000, V=V))
    torch.cuda.synchronize()
    a, b = plans
    assert a.workspace is not None and b.workspace is None
    U = int(a.count.item())
    assert U == int(b.count.item()) == len(np.unique(ids))
    assert torch.equal(a.perm, b.perm) and torch.equal(a.seg, b.seg)
    assert torch.equal(a.start[:U + 1], b.start[:U + 1]) and torch.equal(a.rows[:U], b.rows[:U])
    assert torch.equal(a.counts_vec, b.counts_vec) and int(a.counts_vec[1:].sum()) == U
    assert a.rows.numel() == 4000 and a.rows.dtype == t.dtype
