"""Forward-only build_memory at the C3 shape: the per-sequence inference chain (scan_forward_inference) against the
16-sequence-tile MFMA kernel run layer group by layer group (ops.tiled_forward_inference), over batch sizes.
    python tools/tile_eval_time.py [B ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops
build.build_library()
dev = torch.device("cuda:0")
H, E, V, K, T, F = 64, 16, 3308019, 7, 1001, 2
Bs = [int(a) for a in sys.argv[1:]] or [500, 1000, 2000, 4000, 8000]
spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=(2,) * 10 + (1,), front_zero=23, mask_id0=False, last_index=-2)
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(V, E, device=dev, generator=g) * 0.1
weights = []
for i in range(K):
    D = spec.D0 if i == 0 else H
    weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]


def timed(fn, n=4):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in Bs:
    ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
    ref_mem, ref_last = ops.scan_forward_inference(spec, ids, emb, weights)
    t_ref = timed(lambda: ops.scan_forward_inference(spec, ids, emb, weights))
    line = "B=%5d  per-sequence %7.3f ms %6.0f k/s" % (B, t_ref * 1e3, B / t_ref / 1e3)
    for grp in (1, 2, 7):
        mem, last = ops.tiled_forward_inference(spec, ids, emb, weights, group=grp)
        err = float((mem - ref_mem).abs().max())
        assert torch.equal(last, ref_last)
        t = timed(lambda: ops.tiled_forward_inference(spec, ids, emb, weights, group=grp))
        line += " | tiled g=%d %7.3f ms %6.0f k/s (max|dmem| %.1e)" % (grp, t * 1e3, B / t / 1e3, err)
    print(line, flush=True)
    del ids
