"""Forward-only (eval) timing at the C3 shape: python tools/eval_time.py [B]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops
build.build_library()
dev = torch.device("cuda:0")
H, E, V, K, T, F = 64, 16, 50000, 7, 1001, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=(2,) * 10 + (1,), front_zero=23, mask_id0=False, last_index=-2)
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(V, E, device=dev, generator=g) * 0.3
ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
weights = []
for i in range(K):
    D = spec.D0 if i == 0 else H
    weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]
for _ in range(2):
    ops.scan_forward_inference(spec, ids, emb, weights)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    ops.scan_forward_inference(spec, ids, emb, weights)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("B=%d: build_memory inference %.3f ms -> %.0f k sequences/s" % (B, dt * 1e3, B / dt / 1e3))
