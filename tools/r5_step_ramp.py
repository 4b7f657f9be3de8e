"""Where a SHORT timed region (the driver's `--steps 20 --warmup 5`) loses its 3-4 % against a 200-step one: per-step device
and host times of the first steps after the synchronisation, under the three candidate causes --
  gap     the idle gap bench.py leaves between warm-up and timing (gc.collect + gc.freeze: ~40 ms with the device idle)
  nogap   gc collected BEFORE the warm-up steps: nothing but a synchronize between warm-up and timing
  fresh   batches the warm-up never saw (n_distinct = 8, warm-up 5: batches 5..7 meet the model inside the timed region)
"""
import gc, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
c = dict(bench.CONFIGS[os.environ.get("RAMP_CONFIG", "c3")]); c["config_id"] = "c3"
W = int(os.environ.get("RAMP_WARMUP", "5"))
K = int(os.environ.get("RAMP_STEPS", "20"))


def run(mode):
    m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
    batches = bench.synth_batches(c, 8, c["batch"], 20190521 + 3, dev)

    def step(i):
        ids, label = batches[i % 8]
        m.train_step(ids, label, keep_prob=0.5, global_batch=c["batch"], next_ids=batches[(i + 1) % 8][0],
                     next_global_batch=c["batch"])
    if mode == "nogap":
        gc.collect(); gc.freeze()
    for i in range(W):
        step(i if mode != "fresh" else i + 8 - W)          # fresh: the warm-up steps are the batches the timed region starts on
    torch.cuda.synchronize()
    if mode != "nogap":
        gc.collect(); gc.freeze()
    if mode == "sleep":
        time.sleep(0.5)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        step(W + i)
        ev[i + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print("%-6s total %.2f ms = %.4f ms/step; device per step: %s" % (mode, total, total / K, " ".join("%.2f" % x for x in d)))
    print("       host enqueue done at (ms): %s" % " ".join("%.1f" % (h * 1e3) for h in host), flush=True)
    gc.unfreeze()
    del m, batches
    torch.cuda.empty_cache()


for mode in os.environ.get("RAMP_MODES", "gap,nogap,fresh,sleep,gap,nogap").split(","):
    run(mode)
