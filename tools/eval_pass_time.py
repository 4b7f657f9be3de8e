"""Where Hpmn.eval's time goes at the C3 shape: python tools/eval_pass_time.py"""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from hpmn_amd import hpmn as H
dev = torch.device("cuda:0")
c = dict(bench.CONFIGS["c3"]); c["config_id"] = "c3"
m = bench.build_model(c, tempfile.mkdtemp(), dev)
rng = np.random.default_rng(1)
n = 16000
ds = dict(ids=rng.integers(1, c["V"], size=(n, c["T"], c["F"])).astype(np.int32), label=rng.integers(0, 2, size=n).astype(np.int32))
def t(fn, k=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("eval(ds, 2000): %.1f ms" % t(lambda: m.eval(ds, 2000)))
d = m._dev(ds)
print("4 passes of forward_inference(4000 rows): %.1f ms" % t(lambda: [m.forward_inference(d.ids[i:i + 4000], want_logit=False, want_att=False) for i in range(0, n, 4000)]))
p = torch.rand(n, device=dev)
print("device_auc + log_loss + tolist: %.1f ms" % t(lambda: torch.stack([H.device_auc(p.double(), d.label), H.device_log_loss(p.double(), d.label)]).tolist()))
print("_dev lookup: %.3f ms" % t(lambda: m._dev(ds)))
m.TILED_EVAL_MIN_ROWS = 0
print("eval(ds, 2000) per-sequence kernels, batch by batch: %.1f ms" % t(lambda: m.eval(ds, 2000)))
# --- the same inside a process that has trained (bench.py measured 61 ms per eval() where this script measures 12.6)
import gc
m.TILED_EVAL_MIN_ROWS = 1536
batches = bench.synth_batches(c, 4, c["batch"], 1, dev)
for i in range(10):
    m.train_step(batches[i % 4][0], batches[i % 4][1], keep_prob=0.5, global_batch=c["batch"])
torch.cuda.synchronize()
print("after training: eval(ds, 2000): %.1f ms" % t(lambda: m.eval(ds, 2000)))
gc.collect(); gc.freeze()
print("after gc.collect + gc.freeze: %.1f ms" % t(lambda: m.eval(ds, 2000)))
gc.disable()
print("gc disabled: %.1f ms" % t(lambda: m.eval(ds, 2000)))
gc.enable()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); m.eval(ds, 2000); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
