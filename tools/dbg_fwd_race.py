"""Does the training forward give the same result when an unrelated heavy kernel runs beside it?"""
import sys, os, tempfile, pathlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
from hpmn_amd import ops
dev = torch.device("cuda:0")
tmp = pathlib.Path(tempfile.mkdtemp())
cfg = T.cfg_industry(H=64, K=3, T=41, V=600)
p = T.f32_params(cfg, 151)
B = int(os.environ.get("B", "6"))
ids, label = T.rand_ids(cfg, B, 152)
m = T.make_model(cfg, tmp, p)
ti = torch.as_tensor(ids).to(dev)
emb = m.params["Embedding/emb_mtx"]; w = m._gru_weights()
junk = torch.empty(640_000_000, device=dev); js = torch.cuda.Stream()
ref = None
bad = 0
for it in range(int(os.environ.get("N","40"))):
    conc = it % 2 == 1
    if conc:
        js.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(js): junk.zero_()
    memory, last, saved = ops.scan_forward_train(m.spec, ti, emb, w)
    hs = [s[1].clone() for s in saved]
    torch.cuda.synchronize()
    cur = [memory.clone(), last.clone()] + hs
    if ref is None: ref = cur
    else:
        d = [float((a - b).abs().max()) for a, b in zip(ref, cur)]
        if max(d) > 0 or any(x != x for x in d):
            bad += 1
            print("iter", it, "concurrent" if conc else "alone", "diffs", d)
print("bad", bad)
