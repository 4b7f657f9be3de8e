"""Where an asynchronous step loop spends its first steps: per-25-step wall time and device allocations (C1 by default)."""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import torch, bench
_gc = []
def _cb(phase, info):
    _gc.append((phase, info["generation"], time.perf_counter()))
gc.callbacks.append(_cb)
c = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c1"])
dev = torch.device("cuda:0")
m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
batches = bench.synth_batches(c, 8, c["batch"], 20190521 + 3, dev)
for i in range(5):
    m.train_step(*batches[i % 8], keep_prob=0.5, global_batch=c["batch"])
torch.cuda.synchronize()
out = []
t00 = time.perf_counter()
for blk in range(8):
    t0 = time.perf_counter()
    for i in range(25):
        m.train_step(*batches[(5 + blk * 25 + i) % 8], keep_prob=0.5, global_batch=c["batch"])
    st = torch.cuda.memory_stats()
    out.append("%.1f ms host, %d device allocs, %d MB reserved" % ((time.perf_counter() - t0) * 1e3, st["num_device_alloc"],
                                                                   st["reserved_bytes.all.current"] >> 20))
torch.cuda.synchronize()
print("\n".join(out))
for i in range(0, len(_gc) - 1, 2):
    if _gc[i + 1][2] - _gc[i][2] > 1e-3:
        print("gc gen %d: %.1f ms" % (_gc[i][1], (_gc[i + 1][2] - _gc[i][2]) * 1e3))
print("200 steps: %.3f ms/step" % ((time.perf_counter() - t00) * 1e3 / 200))
