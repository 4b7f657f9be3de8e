"""Latency/throughput of hpmn_memory_update at the C3 model shape (H=64, K=7, D0=32)."""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hpmn_amd.online import OnlineMemory

dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["c3"]); c["V"] = 100000
m = bench.build_model(c, tempfile.mkdtemp(), dev)
for B in (1, 64, 500, 4096, 20000):
    store = OnlineMemory(m, n_users=max(B, 1))
    users = torch.arange(B, dtype=torch.int32, device=dev)
    ids = torch.randint(1, c["V"], (B, 2), dtype=torch.int32, device=dev)
    for _ in range(3):
        store.update(users, ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 64
    for _ in range(n):
        store.update(users, ids)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("B=%6d  %.1f us/call  %.2f M events/s" % (B, dt * 1e6, B / dt / 1e6))
