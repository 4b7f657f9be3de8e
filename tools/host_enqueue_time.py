"""Host time to ENQUEUE a training step against the device time of the step (is the small-table step host-bound?):
python tools/host_enqueue_time.py [config]"""
import gc, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
c = dict(bench.CONFIGS[cfg]); c["config_id"] = cfg
m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
batches = bench.synth_batches(c, 8, c["batch"], 20190521 + 3, dev)
step = lambda i: m.train_step(batches[i % 8][0], batches[i % 8][1], keep_prob=0.5, global_batch=c["batch"],
                              next_ids=batches[(i + 1) % 8][0], next_global_batch=c["batch"])
gc.collect(); gc.freeze()
for i in range(10): step(i)
torch.cuda.synchronize()
for n in (50, 200, 200):
    t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: %d steps: host enqueue %.1f us/step, until the device is done %.1f us/step (device ahead of host by %.1f ms at the end)"
          % (cfg, n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, (t2 - t1) * 1e3), flush=True)
