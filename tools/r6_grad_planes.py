"""r6: how far every variable's gradient is from the all-fp32 kernels' (bench.ALL_FP32_ENV), in units of the gradient's
largest element, with three bf16 planes (default) and with two (rounds 4/5).  Four processes of tests/grad_planes_worker.py."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

# the reference: fp32 kernels for exactly the products under test (GRU weight gradients, input gradients, H = 128 projections);
# the read path's launch is the default one in every run, so that its own (fp32-equivalent) difference does not travel through
# BPTT into the comparison
FP32_GRU = bench.FP32_GRU_ENV
modes = {"planes3": {}, "planes2": {"HPMN_WGRAD_PLANES": "2", "HPMN_DX_PLANES": "2", "HPMN_PROJ_PLANES": "2"},
         "fp32": FP32_GRU, "fp32_again": dict(FP32_GRU, HPMN_WGRAD_SOLO="0")}
res = {}
with tempfile.TemporaryDirectory() as tmp:
    for tag, extra in modes.items():
        env = dict(os.environ); env.update(extra); env["HPMN_DET_SCATTER"] = "1"
        dst = os.path.join(tmp, tag + ".npz")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "grad_planes_worker.py"), dst], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[tag] = dict(np.load(dst))
ref = res["fp32"]
for tag in ("planes3", "planes2", "fp32_again"):
    worst = {}
    for k, b in ref.items():
        if k.endswith("/ce"):
            continue
        a = res[tag][k]
        e = float(np.abs(a - b).max()) / max(1e-30, float(np.abs(b).max()))
        cfg = k.split("/")[0]
        if e > worst.get(cfg, (0, ""))[0]:
            worst[cfg] = (e, k)
    print(tag, " ".join("%s: %.2e (%s)" % (c, e, k.split("/")[1]) for c, (e, k) in sorted(worst.items())))
    if tag != "fp32_again":
        gru = {}
        for k, b in ref.items():
            if "GRU" in k or "emb_mtx" in k:
                e = float(np.abs(res[tag][k] - b).max()) / max(1e-30, float(np.abs(b).max()))
                cfg = k.split("/")[0]
                gru[cfg] = max(gru.get(cfg, 0), e)
        print("   GRU variables + table only:", " ".join("%s: %.2e" % kv for kv in sorted(gru.items())))
