"""Time the pipe kernels alone (HIP events on the launch stream).  python tools/pipe_time.py [B]"""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops, _lib
build.build_library()
dev = torch.device("cuda:0")
H, E, V = 64, 16, 5000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
lib = _lib.load()

def case(T, K, F, fz, periods, train=True, bwd=True):
    spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=periods, front_zero=fz, mask_id0=False, last_index=-1)
    g = torch.Generator(device=dev).manual_seed(0)
    emb = torch.randn(V, E, device=dev, generator=g) * 0.3
    ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
    weights = []
    for i in range(K):
        D = spec.D0 if i == 0 else H
        weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                    torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]
    ops.PIPE = True
    mem, last, saved = ops.pipe_forward(spec, ids, emb, weights, train=True)
    p, lens = ops._pipe_desc(spec, B, weights, train)
    f32 = dict(device=dev, dtype=torch.float32)
    y = [saved[i + 1][0] if i + 1 < K else None for i in range(K)]
    p.x0, p.memory = saved[0][0].data_ptr(), mem.data_ptr()
    d_mem = torch.randn(B, K, H, **f32) * 0.01
    d_act = [torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    d_x = [torch.empty(B, lens[i], H if i else spec.D0, **f32) for i in range(K)]
    p.d_memory = d_mem.data_ptr()
    for i in range(K):
        p.y[i] = y[i].data_ptr() if y[i] is not None else None
        p.hs[i], p.gates[i] = saved[i][1].data_ptr(), saved[i][2].data_ptr()
        p.d_act[i], p.d_x[i] = d_act[i].data_ptr(), d_x[i].data_ptr()
    sync = ops._pipe_sync_buffer(K, B, dev)
    p.sync = sync.data_ptr()
    st = torch.cuda.current_stream()
    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n): fn()
        e1.record(st); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = t(lambda: _lib.check(lib.hpmn_pipe_fwd(C.byref(p), st.cuda_stream), "fwd"))
    tb = t(lambda: _lib.check(lib.hpmn_pipe_bwd(C.byref(p), st.cuda_stream), "bwd")) if bwd else 0
    tg = t(lambda: ops.embed_gather_seq(ids, emb, fz, False, out=saved[0][0]))
    steps = lens[0]
    print("T=%d K=%d F=%d train=%d: fwd %.1f us (%.0f cyc/step @2.4GHz)  bwd %.1f us (%.0f cyc/step)  gather %.1f us"
          % (T, K, F, train, tf, tf * 2400 / steps, tb, tb * 2400 / steps, tg), flush=True)

case(1001, 1, 2, 23, (1,))
case(1001, 1, 2, 23, (1,), train=False, bwd=False)
case(1001, 1, 4, 23, (1,))
case(1001, 2, 2, 23, (2, 1))
case(1001, 7, 2, 23, (2,) * 10 + (1,))
case(1001, 7, 2, 23, (2,) * 10 + (1,), train=False, bwd=False)
