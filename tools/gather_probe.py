"""Cold-cache timing of the in-place gather probe (hpmn_embed_gather_sum) at the C3 id shape: python tools/gather_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops
build.build_library()
dev = torch.device("cuda:0")
B, T, F = 500, 1001, 2
V = 64 * 1024 * 1024
tab = torch.empty(V, 16, device=dev).normal_(0.0, 0.1)
g = torch.Generator(device=dev).manual_seed(99)
ids = [torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g) for _ in range(8)]
out = torch.zeros(B, F * 16, device=dev)
st = torch.cuda.current_stream()
for i in range(4):
    ops.embed_gather_sum(ids[i % 8], tab, False, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
n = 32
for i in range(n):
    ops.embed_gather_sum(ids[i % 8], tab, False, out=out)
e1.record(st); e1.synchronize()
ms = e0.elapsed_time(e1) / n
alg = B * T * F * 68
print("slices=%s: %.1f us -> %.0f GB/s algorithmic = %.3f of 8 TB/s" % (os.environ.get("HPMN_GSUM_SLICES", "auto"), ms * 1e3, alg / ms / 1e6, alg / ms / 1e6 / 8000))
