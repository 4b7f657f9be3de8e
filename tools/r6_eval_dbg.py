"""r6: the evaluation kernels at a short sequence length (T = 41 -> 64 / 32 / 16 / 8 steps): per-sequence kernels against the two
tile kernels (<= 256 tiles: twelve-wave kernel; more: four-wave kernel, two tiles per CU) on the same rows."""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dp_worker
from hpmn_amd import ops
torch.cuda.set_device(0)
m, te, _ = dp_worker.build_eval(tempfile.mkdtemp())
ds = m._dev(te)
for T_note in ("T=41",):
    m.TILED_EVAL_MIN_ROWS = 0
    ref = m.forward_inference(ds.ids[:6800])
    m.TILED_EVAL_MIN_ROWS = 1536
    a = m.forward_inference(ds.ids[:6800])
    b = torch.cat([m.forward_inference(ds.ids[:3400])["prediction"], m.forward_inference(ds.ids[3400:6800])["prediction"]])
    c = torch.cat([m.forward_inference(ds.ids[i:i + 1700])["prediction"] for i in range(0, 6800, 1700)])
    for name, x in (("6800 rows one pass", a["prediction"]), ("2 x 3400", b), ("4 x 1700", c)):
        d = (x - ref["prediction"]).abs()
        print(T_note, name, "max |pred - per-sequence kernels| %.3g" % float(d.max()), "rows off by > 1e-3:", int((d > 1e-3).sum()),
              "first bad rows", torch.nonzero(d > 1e-3).flatten()[:8].tolist())
    dm = (a["memory"] - ref["memory"]).abs()
    print("memory, 6800 one pass: max diff per layer", [float(dm[:, k].max()) for k in range(dm.shape[1])])
print("eval single:", m.eval(te, 500))
m.TILED_EVAL_MIN_ROWS = 0
print("eval per-sequence kernels:", m.eval(te, 500))
