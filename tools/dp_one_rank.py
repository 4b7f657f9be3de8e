"""The data-parallel step's machinery on ONE GPU over RCCL (a one-rank process group with the world-size-1 short cuts off):
plan sort, id gather, counts, chunked rows exchange with itself, late pass per chunk -- against the plain step."""
import os, sys, time, tempfile
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ["HPMN_DP_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")       # (as bench.py sets it for multi-rank runs; 4 = the runtime default)
import torch, torch.distributed as td
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
c = dict(bench.CONFIGS["c3"]); c["config_id"] = "c3"
import gc
for mode in (os.environ.get("DP1_MODES", "plain,plain-dense,rows,allreduce").split(",")):
    # "plain" / "plain-dense": the single-process step (compact gradient rows / dense gradient table) in the same loop
    os.environ["HPMN_DP_FORCE_COLLECTIVES"] = "0" if mode.startswith("plain") else "1"
    os.environ["HPMN_TABLE_GRAD"] = "dense" if mode in ("plain-dense", "allreduce") else os.environ.get("DP1_TABLE_GRAD", "compact")
    os.environ["HPMN_TABLE_EXCHANGE"] = "auto" if mode.startswith("plain") else mode
    m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
    m.table_exchange_chunks = int(os.environ.get("DP1_CHUNKS", "4"))
    batches = bench.synth_batches(c, 8, c["batch"], 20190521 + 3, dev)
    def step(i):
        ids, label = batches[i % 8]
        nxt = dict(next_ids=batches[(i + 1) % 8][0], next_global_batch=c["batch"]) if os.environ.get("DP1_NEXT_IDS", "1") != "0" else {}
        m.train_step(ids, label, keep_prob=0.5, global_batch=c["batch"], **nxt)
    for i in range(8): step(i)
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()                                  # (as bench.py: a generation-2 pass inside the loop is ~35 ms)
    t0 = time.perf_counter()
    n = int(os.environ.get("DP1_STEPS", "100"))
    for i in range(n): step(8 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3                  # (r5: r4 read the clock after the 20 host-time steps below -- x1.2)
    if os.environ.get("DP1_PROFILE") == "1" and mode == "rows":
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for i in range(20): step(300 + i)
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    th0 = time.perf_counter()
    for i in range(20): step(200 + i)
    host = (time.perf_counter() - th0) / 20 * 1e3
    torch.cuda.synchronize()
    print("host enqueue time per step (no sync inside the loop): %.3f ms" % host)
    print("one rank, forced collectives, exchange=%s (ran as %s): %.3f ms/step, dp step used: %s" % (mode, getattr(m, "last_exchange_mode", None), ms, m._dp_two_pass(batches[0][0])), flush=True)
    del m
td.destroy_process_group()
