"""Deterministic scatter at the C3 shape: plan build, the two segsum passes and the atomic kernel it replaces.
    python tools/segsum_time.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops
build.build_library()
dev = torch.device("cuda:0")
B, T, F, E, V, Z = 500, 1001, 2, 16, 3308019, 23
g = torch.Generator(device=dev).manual_seed(0)
ids = torch.randint(1, V - 30000, (B, T, F), device=dev, dtype=torch.int32, generator=g)
ids[:, :, 0] = torch.randint(V - 30000, V, (B, 1), device=dev, dtype=torch.int32, generator=g)
dx = torch.randn(B, Z + T, F * E, device=dev, generator=g)
demb = torch.zeros(V, E, device=dev)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t_plan = timed(lambda: ops.ScatterPlan(ids, E, want_rows=True))
plan = ops.ScatterPlan(ids, E, want_rows=True)
t_seg = timed(lambda: ops.embed_grad_segsum(plan, (B, T, F), dx, demb, Z, False))
plan2 = ops.ScatterPlan(ids, E, want_rows=False)
t_seg2 = timed(lambda: ops.embed_grad_segsum(plan2, (B, T, F), dx, demb, Z, False))
t_atomic = timed(lambda: ops.embed_grad_scatter(ids, dx, demb, Z, False))
print("plan %.1f us   segsum (dense + rows) %.1f us   segsum (dense only) %.1f us   atomic scatter %.1f us   U=%d"
      % (t_plan, t_seg, t_seg2, t_atomic, plan.count_host()))
