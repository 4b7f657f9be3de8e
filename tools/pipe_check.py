"""Developer check of the all-layers-in-one-launch scan against the per-layer kernels (and timing of both).
Usage (GPU box): python tools/pipe_check.py [B] [T] [K] [F]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops  # noqa: E402

build.build_library()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
K = int(sys.argv[3]) if len(sys.argv) > 3 else 7
F = int(sys.argv[4]) if len(sys.argv) > 4 else 2
industry = T == 1001 or T == 41
H, E, V = 64, 16, 5000
periods = tuple([2] * 10 + [1]) if industry else (2, 2, 3, 5, 5, 1)
spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=periods[:K] if not industry else periods, front_zero=23 if industry else 0,
                    mask_id0=not industry, last_index=-2 if industry else -1)
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(V, E, device=dev, generator=g) * 0.3
ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
weights = []
for i in range(K):
    D = spec.D0 if i == 0 else H
    weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]
print("pipe supported:", ops.pipe_supported(spec), "lens", spec.layer_lengths())


def run(pipe):
    ops.PIPE = "all" if pipe is True else ("0" if pipe is False else pipe)
    mem, last, saved = ops.scan_forward_train(spec, ids, emb, weights)
    torch.cuda.synchronize()
    return mem, last, saved


mem_p, last_p, saved_p = run(True)
print("fwd error word:", ops.pipe_error_word(K, B, dev))
mem_l, last_l, saved_l = run(False)
print("memory max|diff| pipe vs layers: %.3e   (max |memory| %.3f)" % (float((mem_p - mem_l).abs().max()), float(mem_l.abs().max())))
print("last equal:", bool(torch.equal(last_p, last_l)))
for i in range(K):
    print("layer %d  hs %.3e  gates %.3e  x %.3e" % (i, float((saved_p[i][1] - saved_l[i][1]).abs().max()),
                                                    float((saved_p[i][2] - saved_l[i][2]).abs().max()),
                                                    float((saved_p[i][0] - saved_l[i][0]).abs().max())))
# backward
d_mem = torch.randn(B, K, H, device=dev, generator=g) * 0.01
d_last = torch.randn(B, spec.D0, device=dev, generator=g) * 0.01


def grads(pipe, saved):
    ops.PIPE = "all" if pipe is True else ("0" if pipe is False else pipe)
    out = [torch.zeros(V, E, device=dev)] + [torch.zeros_like(w) for w in weights]
    ops.scan_backward(spec, ids, saved, weights, d_mem, d_last, out)
    torch.cuda.synchronize()
    return out


g_l = grads(False, saved_l)
g_p = grads(True, saved_l)          # same saved states: isolates the reverse kernels
print("bwd error word:", ops.pipe_error_word(K, B, dev))
for j, (a, b) in enumerate(zip(g_p, g_l)):
    print("grad %2d  max|diff| %.3e  of max %.3e" % (j, float((a - b).abs().max()), float(b.abs().max())))


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g_u = grads("upper", saved_l)
print("upper mode grads vs layers: max rel %.3e" % max(float((a - b).abs().max() / b.abs().max().clamp(min=1e-20)) for a, b in zip(g_u[1:], g_l[1:])))
mem_u, last_u, saved_u = run("upper")
print("upper mode memory vs layers: %.3e" % float((mem_u - mem_l).abs().max()))
for pipe in ("all", "upper", "0"):
    ops.PIPE = pipe
    tf = timeit(lambda: ops.scan_forward_train(spec, ids, emb, weights))
    sv = saved_l
    out = [torch.zeros(V, E, device=dev)] + [torch.zeros_like(w) for w in weights]
    tb = timeit(lambda: ops.scan_backward(spec, ids, sv, weights, d_mem, d_last, out))
    print("pipe=%s  forward %.3f ms   backward %.3f ms" % (pipe, tf, tb))
