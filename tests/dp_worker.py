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
    for lo, hi in ds.batches(batch):
        a, b = dist.shard_bounds(lo, hi, m.rank, m.world)
        m.train_step(ds.ids[a:b], ds.label[a:b], keep_prob=1.0, global_batch=hi - lo)
        step += 1
        if step == steps:
            break
    # a "last batch" of ONE sample: with two ranks one shard is empty, and that rank must still take part in
    # every collective of the step
    a, b = dist.shard_bounds(0, 1, m.rank, m.world)
    m.train_step(ds.ids[a:b], ds.label[a:b], keep_prob=1.0, global_batch=1)
    auc, ll, mem = m.eval(te, 64)
    out = {k: v.detach().cpu().numpy() for k, v in m.params.items()}
    out["__eval__"] = np.array([auc, ll, mem])
    return out


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
    m, tr, te = build(sys.argv[1] + ".model%d" % td.get_rank())
    assert m.world == 2
    out = run(m, tr, te)
    if td.get_rank() == 0:
        np.savez(sys.argv[1], **out)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
