"""The read path's training launch ALONE on the chip (nothing on another stream), HIP events around 20 launches:
python tools/read_alone.py [c3|c1|c2] [batch]     (HPMN_LIB_PATH=<variant> for the clock build: its phase line is printed)"""
import os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hpmn_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
c = dict(bench.CONFIGS[name])
if c["V"] > 1000000:
    c["V"] = 1000000
B = int(sys.argv[2]) if len(sys.argv) > 2 else c["batch"]
dev = torch.device("cuda:0")
m = bench.build_model(c, tempfile.mkdtemp(), dev)
K, H, D0 = c["K"], c["H"], c["F"] * 16
g = torch.Generator(device=dev).manual_seed(0)
memory = torch.randn(B, K, H, device=dev, generator=g) * 0.3
last = torch.randn(B, D0, device=dev, generator=g) * 0.3
label = torch.randint(0, 2, (B,), device=dev, dtype=torch.int32, generator=g)
loss = torch.zeros(2, device=dev)


def run(defer):
    return ops.read_fwd_bwd(m._read_desc, m._read_params, m._read_grads, memory, last, label, None, 1.0, 1.0 / B,
                            c["memory_reg"], loss_out=loss, defer_param_grads=defer)


for defer in (True, False):
    for _ in range(3):
        run(defer)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run(defer)
    e1.record()
    torch.cuda.synchronize()
    print("%s B=%d: %s %.1f us per call" % (name, B, "training launch alone" if defer else "with the weight-gradient launches",
                                            e0.elapsed_time(e1) * 1e3 / 20))
