import sys, os, tempfile, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import torch, bench
c = dict(bench.CONFIGS["c1"])
dev = torch.device("cuda:0")
m = bench.build_model(c, tempfile.mkdtemp(), dev)
batches = bench.synth_batches(c, 8, c["batch"], 20190521 + 3, dev)
def step(i):
    ids, label = batches[i % 8]
    return m.train_step(ids, label, keep_prob=0.5, global_batch=c["batch"])
for i in range(20): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop %.1f us/step, with drain %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(200): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
