"""Soak: a few thousand training steps through the product harness (periodic evals, early-stop logic, fused
forward, in-kernel dropout) on synthetic Amazon-style and XLong-style data; checks finiteness and learning."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpmn_amd import datasets as D
from hpmn_amd.hpmn import Hpmn, Hpmn_Industry

t0 = time.time()
tr, te, fs = D.make_synthetic_amazon(n_samples=40000, n_item=5000, n_cate=80, n_user=8000, max_len=100,
                                     seed=D.SEED_BASE + 21, as_arrays=True)
m = Hpmn(tempfile.mkdtemp(), tr, te, fs, 3, 2, 100, 100, 0.003, 32, 16, 3, [2, 2, 5, 5, 1], [2, 2, 5, 5, 1], 3, 3,
         True, False, l2_reg=0., memory_reg=1e-5, verbose=False, seed=5)
best = m.train(6, 128)
auc, ll, mem = m.eval(te, 512)
print("amazon-style: %d steps max, best test AUC %.4f, final %.4f logloss %.4f (%.1fs)" % (6 * len(tr["label"]) // 128, best, auc, ll, time.time() - t0))
assert np.isfinite([auc, ll, mem]).all() and best > 0.75

t0 = time.time()
ids, label = D.make_synthetic_xlong_arrays(150000, seed=31)
tids, tlabel = D.make_synthetic_xlong_arrays(2500, seed=32)
emb = D.make_synthetic_graph_emb(seed=33)
init = np.concatenate((emb, np.zeros((D.XLONG_USERS, 16), np.float32), np.zeros((D.XLONG_PV_CNT, 16), np.float32)), 0)
m = Hpmn_Industry(tempfile.mkdtemp(), dict(ids=ids, label=label), dict(ids=tids, label=tlabel), D.xlong_feature_size(), 2, 1, 1001, 1,
                  0.001, 64, 16, 3, [2] * 10 + [1], [1], 7, 1, True, False, emb_initializer=init, l2_reg=0, memory_reg=5e-5,
                  verbose=False, seed=0)
m.eval_every = 100
ds = m._dev(m.trainset)
step = 0
for lo, hi in ds.batches(500):
    step += 1
    out, ce = m.train_step(ds.ids[lo:hi], ds.label[lo:hi], keep_prob=0.5)
    if step % 100 == 0:
        a, l, mm = m.eval(m.testset, 2000)
        print("xlong-style step %d: ce %.4f test AUC %.4f logloss %.4f memloss %.2f" % (step, float(ce), a, l, mm), flush=True)
        assert np.isfinite([float(ce), a, l, mm]).all()
print("xlong-style: %d steps in %.1fs" % (step, time.time() - t0))
assert a > 0.97

# H = 128 (configs[4]'s kernels: four-wave scans at two workgroups per CU, column-split weight gradients, staged input gradient)
t0 = time.time()
ids, label = D.make_synthetic_xlong_arrays(100000, seed=41)
tids, tlabel = D.make_synthetic_xlong_arrays(2500, seed=42)
m = Hpmn_Industry(tempfile.mkdtemp(), dict(ids=ids, label=label), dict(ids=tids, label=tlabel), D.xlong_feature_size(), 2, 1, 1001, 1,
                  0.001, 128, 16, 3, [2] * 10 + [1], [1], 7, 1, True, False, emb_initializer=init, l2_reg=0, memory_reg=5e-5,
                  verbose=False, seed=0)
ds = m._dev(m.trainset)
step = 0
for lo, hi in ds.batches(500):
    step += 1
    out, ce = m.train_step(ds.ids[lo:hi], ds.label[lo:hi], keep_prob=0.5)
    if step % 100 == 0:
        a, l, mm = m.eval(m.testset, 2000)
        print("H=128 step %d: ce %.4f test AUC %.4f logloss %.4f memloss %.2f" % (step, float(ce), a, l, mm), flush=True)
        assert np.isfinite([float(ce), a, l, mm]).all()
print("H=128: %d steps in %.1fs" % (step, time.time() - t0))
assert a > 0.95
