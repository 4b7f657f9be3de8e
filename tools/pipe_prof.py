"""Per-segment cycle profile of the forward pipe kernel (library built with -DHPMN_PIPE_PROF)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import ops, _lib
dev = torch.device("cuda:0")
H, E, V = 64, 16, 5000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T, F, fz = 1001, 2, 23
periods = (2,) * 10 + (1,)
spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=periods, front_zero=fz, mask_id0=False, last_index=-1)
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(V, E, device=dev, generator=g) * 0.3
ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
weights = []
for i in range(K):
    D = spec.D0 if i == 0 else H
    weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]
for train in (True, False):
    ops.pipe_forward(spec, ids, emb, weights, train=train)
    ops.pipe_forward(spec, ids, emb, weights, train=train)
    torch.cuda.synchronize()
    buf = ops._pipe_sync_buffer(K, B, dev)
    one = (4 * (2 + K * ((B + 15) // 16) * 4) + 255) // 256 * 256
    prof = buf[2 * one:].view(torch.int64)[1024:1024 + K * 3 * 8].cpu().view(K, 3, 8)
    lens = spec.layer_lengths()
    print("train=%d  (s_memtime ticks per step; R: recur|sigmoid|split+write|barrierA|project|barrierB ; C: project(+park)|barrierA|recur|tanh+split+write|stores|barrierB|publish)" % train)
    w0 = min(int(prof[i, 2, 2]) for i in range(K))
    for i in range(K):
        print("  layer %d: loop %.1f us (%.0f cyc/step, clock %.2f GHz), loop start +%.1f us" % (
            i, float(prof[i, 2, 1]) / 100.0, float(prof[i, 2, 0]) / lens[i], float(prof[i, 2, 0]) / max(1.0, float(prof[i, 2, 1])) / 10.0,
            (int(prof[i, 2, 2]) - w0) / 100.0))
    for i in range(0):
        for r, name in enumerate("RUC"):
            print("  layer %d %s: %s   total %.0f" % (i, name, " ".join("%6.0f" % (float(v) / lens[i]) for v in prof[i, r]), float(prof[i, r].sum()) / lens[i]))
