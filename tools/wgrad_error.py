"""Weight gradients of one H = 64 GRU layer: error against float64 and time, for the kernel the library dispatches
(HPMN_WGRAD_BF16=1 split-bf16 matrix pipe / =0 fp32 matrix pipe).  Worst-case inputs: mixed-sign values whose magnitudes
span 1e-6 .. 1 (the gradient scale of a training step), 1024-step reductions.
    python tools/wgrad_error.py [D [H]]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops
build.build_library()
dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(7)


def mixed(shape):
    return (rng.normal(size=shape) * 10.0 ** rng.uniform(-6, 0, size=shape)).astype(np.float32)


def run(B, T, timing=False):
    x, hs = mixed((B, T, D)), rng.uniform(-1, 1, size=(B, T + 1, H)).astype(np.float32)
    gates = rng.uniform(0, 1, size=(B, T, 3 * H)).astype(np.float32)
    dact = mixed((B, T, 3 * H))
    t = lambda a: torch.as_tensor(a).to(dev)
    tx, ths, tg, tda = t(x), t(hs), t(gates), t(dact)
    wg, wc = torch.zeros(D + H, 2 * H, device=dev), torch.zeros(D + H, H, device=dev)
    outs = [torch.zeros(D + H, 2 * H, device=dev), torch.zeros(2 * H, device=dev), torch.zeros(D + H, H, device=dev), torch.zeros(H, device=dev)]
    ops.gru_param_grads(tx, ths, tg, tda, wg, wc, *outs, want_dx=False, whole_cu=True)
    torch.cuda.synchronize()
    if timing:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gru_param_grads(tx, ths, tg, tda, wg, wc, *outs, want_dx=False, whole_cu=True)
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3
    x64, h64, r64, da64 = x.astype(np.float64), hs[:, :T].astype(np.float64), gates[:, :, :H].astype(np.float64), dact.astype(np.float64)
    zg = np.concatenate([x64, h64], 2).reshape(B * T, D + H)
    zc = np.concatenate([x64, r64 * h64], 2).reshape(B * T, D + H)
    da = da64.reshape(B * T, 3 * H)
    want = [zg.T @ da[:, :2 * H], da[:, :2 * H].sum(0), zc.T @ da[:, 2 * H:], da[:, 2 * H:].sum(0)]
    cond = [np.abs(zg).T @ np.abs(da[:, :2 * H]), np.abs(da[:, :2 * H]).sum(0), np.abs(zc).T @ np.abs(da[:, 2 * H:]), np.abs(da[:, 2 * H:]).sum(0)]
    res = []
    for name, g, w, cnd in zip(("d_wg", "d_bg", "d_wc", "d_bc"), outs, want, cond):
        err = np.abs(g.cpu().numpy().astype(np.float64) - w)
        res.append((name, err.max() / np.abs(w).max(), (err / cnd).max(), np.percentile(err / np.abs(w).max(), [50, 90, 99, 100])))
    return res


mode = os.environ.get("HPMN_WGRAD_BF16", "1")
for B, T in ((8, 1024), (3, 41)):
    for name, rel_max, rel_cond, pct in run(B, T):
        print(("bf16=%s H=" + str(H) + " B=%d T=%d D=%d %-5s max|err|/max|grad| %.2e   max|err|/sum|terms| %.2e   percentiles of |err|/max|grad| (50/90/99/100) %s")
              % (mode, B, T, D, name, rel_max, rel_cond, " ".join("%.1e" % p for p in pct)))
print("bf16=%s time B=500 T=1024 D=%d H=%d whole_cu: %.1f us" % (mode, D, H, run(500, 1024, timing=True)))
