import subprocess, sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import bench
for nb, lr in ((25, 0.0001), (25, 0.0003), (8, 0.0003)):
    outs = []
    for tag, extra in (("split", {}), ("fp32", bench.ALL_FP32_ENV), ("fp32b", bench.ALL_FP32_ENV)):
        env = dict(os.environ); env.update(extra); env["TRAJ_NB"] = str(nb); env["TRAJ_LR"] = str(lr)
        env["HPMN_DET_SCATTER"] = os.environ.get("DET", "1")
        dst = "/tmp/traj_%s.npz" % tag
        subprocess.run([sys.executable, "/root/repo/tests/traj_worker.py", dst], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(np.load(dst)["loss"])
    a, b, c = outs
    rel = np.abs(a - b) / np.abs(b); rel2 = np.abs(c - b) / np.abs(b)
    print("nb %d lr %g: loss first/last %.4f %.4f; split vs fp32 max rel %.2e (first > 1e-3 at step %s); fp32 vs fp32 max rel %.2e"
          % (nb, lr, b[:10].mean(), b[-10:].mean(), rel.max(), (np.argmax(rel > 1e-3) if (rel > 1e-3).any() else None), rel2.max()))
