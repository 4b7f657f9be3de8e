"""Feasibility / upper bound of a captured training step (hipGraph through torch.cuda.CUDAGraph): capture ONE train_step of a
config with its per-step scalars frozen (lr_t, dropout seed -- so the replayed arithmetic is NOT a valid training run; this only
measures what the launch gaps cost) and time replays against the ordinary host-driven loop.  python tools/graph_probe.py c1"""
import os, sys, time, tempfile, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
c = dict(bench.CONFIGS[cfg]); c["config_id"] = cfg
m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
batches = bench.synth_batches(c, 8, c["batch"], 20190521 + 3, dev)
for i in range(12):
    m.train_step(*batches[i % 8], keep_prob=0.5, global_batch=c["batch"])
torch.cuda.synchronize(); gc.collect(); gc.freeze()
n = 300
t0 = time.perf_counter()
for i in range(n):
    m.train_step(*batches[i % 8], keep_prob=0.5, global_batch=c["batch"])
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / n * 1e3
ids, label = batches[0][0].clone(), batches[0][1].clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        m.train_step(ids, label, keep_prob=0.5, global_batch=c["batch"])
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        m.train_step(ids, label, keep_prob=0.5, global_batch=c["batch"])
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400]); sys.exit(0)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / n * 1e3
print("%s: host-driven %.4f ms/step, captured graph replay %.4f ms/step (%.1f %%)" % (cfg, plain, graph, 100 * (graph / plain - 1)))
