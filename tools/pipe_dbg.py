import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build, ops
build.build_library()
dev = torch.device("cuda:0")
H, E, V = 64, 16, 500
def case(B, T, K, F, fz, periods):
    spec = ops.ScanSpec(F=F, E=E, H=H, K=K, T=T, periods=periods, front_zero=fz, mask_id0=False, last_index=-1)
    g = torch.Generator(device=dev).manual_seed(0)
    emb = torch.randn(V, E, device=dev, generator=g) * 0.3
    ids = torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=g)
    weights = []
    for i in range(K):
        D = spec.D0 if i == 0 else H
        weights += [torch.randn(D + H, 2 * H, device=dev, generator=g) * 0.15, torch.ones(2 * H, device=dev),
                    torch.randn(D + H, H, device=dev, generator=g) * 0.15, torch.zeros(H, device=dev)]
    ops.PIPE = "all"
    mp, lp, sp = ops.scan_forward_train(spec, ids, emb, weights); torch.cuda.synchronize()
    err = ops.pipe_error_word(K, B, dev)
    ops.PIPE = "0"
    ml, ll, sl = ops.scan_forward_train(spec, ids, emb, weights); torch.cuda.synchronize()
    msg = "B=%d T=%d K=%d F=%d fz=%d per=%s err=%d mem %.2e |" % (B, T, K, F, fz, periods[:K], err, float((mp - ml).abs().max()))
    for i in range(K):
        d = (sp[i][1] - sl[i][1]).abs().amax(dim=(0, 2))      # per time row
        bad = torch.nonzero(d > 1e-4)
        first = int(bad[0]) if bad.numel() else -1
        db = (sp[i][1] - sl[i][1]).abs().amax(dim=(1, 2))
        badb = torch.nonzero(db > 1e-4).flatten().tolist()
        msg += " L%d max %.1e first_t %d bad_b %s |" % (i, float(d.max()), first, badb[:6] + (["..."] if len(badb) > 6 else []))
    print(msg, flush=True)
case(3, 6, 1, 2, 0, (1,))
case(3, 8, 2, 2, 0, (2, 1))
case(40, 8, 1, 2, 0, (1,))
case(3, 1001, 1, 2, 23, (1,))
case(3, 1001, 2, 2, 23, (2, 1))
case(40, 64, 3, 2, 0, (2, 2, 1))
case(500, 64, 1, 2, 0, (1,))
case(500, 1001, 1, 2, 23, (1,))
case(500, 1001, 7, 2, 23, (2,) * 10 + (1,))
case(128, 300, 5, 4, 0, (2, 2, 3, 5, 5, 1))
case(128, 100, 4, 3, 0, (2, 2, 5, 5, 1))
