"""How much of a training step is CPU issue time vs GPU time (C3 shape)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
c = dict(bench.CONFIGS["c3"])
dev = torch.device("cuda", 0)
m = bench.build_model(c, tempfile.mkdtemp(), dev)
batches = bench.synth_batches(c, 2, c["batch"], 1, dev)
for i in range(3):
    m.train_step(*batches[i % 2], keep_prob=0.5)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for i in range(n):
    m.train_step(*batches[i % 2], keep_prob=0.5)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("chunks=%s issue %.2f ms/step, total %.2f ms/step" % (os.environ.get("HPMN_PIPELINE_CHUNKS", "4"), (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
