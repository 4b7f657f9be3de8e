"""Train the C3 (XLong) graph on planted-signal synthetic data and report held-out AUC over steps.
Usage: python tools/xlong_auc.py [steps] [lines_train] [lines_test]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hpmn_amd import datasets as D  # noqa: E402
from hpmn_amd.hpmn import Hpmn_Industry  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    n_test = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    t0 = time.time()
    ids, label = D.make_synthetic_xlong_arrays(n_train, seed=1)
    tids, tlabel = D.make_synthetic_xlong_arrays(n_test, seed=2)
    emb = D.make_synthetic_graph_emb(seed=3)
    print("data %.1fs" % (time.time() - t0), flush=True)
    V = D.xlong_feature_size()
    init = np.concatenate((emb, np.zeros((D.XLONG_USERS, 16), np.float32), np.zeros((D.XLONG_PV_CNT, 16), np.float32)), 0)
    tmp = tempfile.mkdtemp()
    m = Hpmn_Industry(tmp, dict(ids=ids, label=label), dict(ids=tids, label=tlabel), V, 2, 1, 1001, 1, 0.001, 64, 16, 3,
                      [2] * 10 + [1], [1], 7, 1, True, False, emb_initializer=init, l2_reg=0, memory_reg=5e-5,
                      verbose=False, seed=0)
    ds = m._dev(m.trainset)
    step = 0
    print("step 0 test auc/loss/mem", m.eval(m.testset, 2000), flush=True)
    t0 = time.time()
    for lo, hi in ds.batches(500):
        step += 1
        m.train_step(ds.ids[lo:hi], ds.label[lo:hi], keep_prob=0.5)
        if step % 10 == 0:
            print("step", step, "test", m.eval(m.testset, 2000), "%.1fs" % (time.time() - t0), flush=True)
        if step >= steps:
            break


if __name__ == "__main__":
    main()
