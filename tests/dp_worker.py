"""Worker of tests/test_gpu_dp.py: one rank of a 2-process data-parallel run of the PRODUCT train/eval
path.  Both ranks use cuda:0 and the gloo backend (NCCL/RCCL refuses two ranks on one device; what
is under test is the sharding, the two-range gradient exchange and the replicated update, not the
transport).  Usage: launched by torch.distributed.run; argv[1] = output .npz (written by rank 0)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(tmp):
    from hpmn_amd import datasets
    from hpmn_amd.hpmn import Hpmn
    tr, te, fs = datasets.make_synthetic_amazon(n_samples=600, n_item=300, n_cate=20, n_user=200, max_len=100,
                                                seed=datasets.SEED_BASE + 7, as_arrays=True)
    m = Hpmn(tmp, tr, te, fs, 3, 2, 100, 100, 0.003, 32, 16, 3, [2, 2, 5, 5, 1], [2, 2, 5, 5, 1], 3, 3, True, False,
             l2_reg=0., memory_reg=1e-5, verbose=False, seed=3)
    return m, tr, te


def run(m, tr, te, steps=5, batch=50):
    ds = m._dev(tr)
    from hpmn_amd import dist
    step = 0
    # HPMN_DP_NEXT_IDS=1: every step is told the next batch's ids (what Hpmn.train() does): the next step's scatter plan and
    # the exchange of the ranks' distinct-row lists then run a step ahead -- incl. the step whose NEXT shard is empty
    ahead = os.environ.get("HPMN_DP_NEXT_IDS") == "1"
    allb = list(ds.batches(batch))
    order = allb[:steps] + [(0, 1)] + allb[steps:steps + 1]
    # (the entry before the last: a "last batch" of ONE sample -- with two ranks one shard is empty, and that rank must still
    #  take part in every collective of the step; the entry BEHIND it is the next epoch's first step: the rank whose shard was
    #  empty holds the other ranks' gradient sum in its flat gradient and must not accumulate onto it -- ADVICE r5)
    for k, (lo, hi) in enumerate(order):
        a, b = dist.shard_bounds(lo, hi, m.rank, m.world)
        nxt = {}
        if ahead and k + 1 < len(order):
            lo2, hi2 = order[k + 1]
            a2, b2 = dist.shard_bounds(lo2, hi2, m.rank, m.world)
            nxt = dict(next_ids=ds.ids[a2:b2], next_global_batch=hi2 - lo2)
        m.train_step(ds.ids[a:b], ds.label[a:b], keep_prob=1.0, global_batch=hi - lo, **nxt)
        step += 1
    auc, ll, mem = m.eval(te, 64)
    out = {k: v.detach().cpu().numpy() for k, v in m.params.items()}
    out["__eval__"] = np.array([auc, ll, mem])
    return out


BIG_V = 2_200_000_000        # more rows than an int32 id can name (HPMN_DP_BIG=1)


def build_big(tmp):
    """configs[4]'s size class under data parallel: a 2.2 G-row table (rows of 4 floats: 35 GB per buffer), lazy (row-wise)
    table Adam -- param + m + v, no dense gradient -- and int64 ids.  Only the top 400 rows are ever named."""
    from hpmn_amd.hpmn import Hpmn_Industry
    rng = np.random.default_rng(5)
    n = 48
    ids = rng.integers(BIG_V - 400, BIG_V, size=(n, 41, 4), dtype=np.int64)
    ids[:, :, 0] = ids[:, -1:, 0]
    tr = dict(ids=ids, label=rng.integers(0, 2, size=n).astype(np.int32))
    m = Hpmn_Industry(tmp, tr, tr, BIG_V, 4, 1, 41, 1, 0.003, 64, 4, 3, [2] * 10 + [1], [1], 3, 1, True, False,
                      memory_reg=5e-5, verbose=False, seed=3, lazy_table_adam=True)
    return m, tr, tr


def run_big(m, tr, te, steps=3, batch=16):
    from hpmn_amd import dist
    ds = m._dev(tr)
    assert ds.ids.dtype == torch.int64
    for step, (lo, hi) in enumerate(ds.batches(batch)):
        a, b = dist.shard_bounds(lo, hi, m.rank, m.world)
        m.train_step(ds.ids[a:b], ds.label[a:b], keep_prob=1.0, global_batch=hi - lo)
        if step + 1 == steps:
            break
    out = {k: v.detach().cpu().numpy() for k, v in m.params.items() if k != "Embedding/emb_mtx"}
    emb = m.params["Embedding/emb_mtx"]
    out["top_rows"] = emb[BIG_V - 400:].detach().cpu().numpy()
    out["below"] = emb[BIG_V - 100400:BIG_V - 400].abs().max().reshape(1).cpu().numpy()
    out["m_top"] = m.flat_m[(BIG_V - 400) * 4:BIG_V * 4].detach().cpu().numpy()
    out["__eval__"] = np.array(m.eval(te, 16))
    return out


def build_eval(tmp):
    """HPMN_DP_EVAL=1: evaluation alone -- an XLong-shaped graph at H = 64 from seeded weights, 6 800 rows: a single process
    takes the 16-sequence tile kernels in one pass, two ranks 3 400 rows each, four ranks 1 700 each (all on the tile kernels:
    a row's arithmetic does not depend on which tile it lands in)."""
    from hpmn_amd.hpmn import Hpmn_Industry
    rng = np.random.default_rng(17)
    n, T, V = 6800, 41, 900
    ids = rng.integers(40, V, size=(n, T, 2)).astype(np.int32)
    ids[:, :, 0] = ids[:, :1, 0] % 20 + 1
    te = dict(ids=ids, label=rng.integers(0, 2, size=n).astype(np.int32))
    m = Hpmn_Industry(tmp, te, te, V, 2, 1, T, 1, 0.003, 64, 16, 3, [2] * 10 + [1], [1], 4, 1, True, False,
                      memory_reg=5e-5, verbose=False, seed=3)
    # (weights of a trained model's size.  Variable by variable: the flat buffer pads the table region to a multiple of
    #  256 x world, so an offset into it is not the same variable in a single process and under data parallel)
    g = torch.Generator(device="cpu").manual_seed(7)
    for name in sorted(m.params):
        if name != "Embedding/emb_mtx":
            m.params[name] += 0.2 * torch.randn(m.params[name].shape, generator=g).to(m.device)
    return m, te, te


def run_eval(m, tr, te):
    return {"__eval__": np.array(m.eval(te, 500), dtype=np.float64)}


def main():
    # HPMN_DP_BACKEND=nccl: one GPU per rank over RCCL (needs >= 2 devices); default: gloo, both ranks on cuda:0
    backend = os.environ.get("HPMN_DP_BACKEND", "gloo")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        td.init_process_group("gloo")
        torch.cuda.set_device(0)
    big = os.environ.get("HPMN_DP_BIG") == "1"
    ev = os.environ.get("HPMN_DP_EVAL") == "1"
    m, tr, te = (build_eval if ev else build_big if big else build)(sys.argv[1] + ".model%d" % td.get_rank())
    assert m.world == int(os.environ.get("WORLD_SIZE", "1"))
    out = (run_eval if ev else run_big if big else run)(m, tr, te)
    if td.get_rank() == 0:
        np.savez(sys.argv[1], **out)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
