"""Time build_memory's training forward (hpmn_scan_fwd_train) and BPTT (hpmn_scan_bwd) alone, HIP events on the launch
stream, at the C3 shape (or `B T K F`).  The kernel switches (HPMN_PAIR_FWD, HPMN_PAIR_BWD, ...) are read at library load:
one process per setting.   python tools/fwd_time.py [B [T [K [F]]]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops  # noqa: E402

build.build_library()
dev = torch.device("cuda:0")
H, E, V = 64, 16, 50000
arg = [int(a) for a in sys.argv[1:]]
B = arg[0] if len(arg) > 0 else 500
T = arg[1] if len(arg) > 1 else 1001
K = arg[2] if len(arg) > 2 else 7
F = arg[3] if len(arg) > 3 else 2
fz = 23 if T == 1001 else 0
spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=(2,) * 10 + (1,), front_zero=fz, mask_id0=False, last_index=-2)
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(V, E, device=dev, generator=g) * 0.3
ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
weights = []
for i in range(K):
    D = spec.D0 if i == 0 else H
    weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]
st = torch.cuda.current_stream()


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


keep = {}


def fwd():
    keep["out"] = ops.abi_forward_train(spec, ids, emb, weights)


tf = timed(fwd)
mem, last, saved = keep["out"]
d_mem = torch.randn(B, K, H, device=dev, generator=g) * 0.01
d_last = torch.zeros(B, spec.D0, device=dev)
grad_out = [torch.zeros_like(emb)] + [torch.zeros_like(w) for w in weights]


def bwd():
    ops.abi_backward(spec, ids, saved, weights, d_mem, d_last, grad_out)


tb = timed(bwd) if not os.environ.get("FWD_ONLY") else 0.0
print("B=%d T=%d K=%d F=%d  PAIR_FWD=%s PAIR_BWD=%s:  forward %.1f us   BPTT incl. weight gradients + scatter %.1f us"
      % (B, T, K, F, os.environ.get("HPMN_PAIR_FWD", "-"), os.environ.get("HPMN_PAIR_BWD", "-"), tf, tb), flush=True)
