"""Worker of test_split_gradient_kernels_track_the_fp32_kernels_over_200_steps: 200 training steps of a small XLong-shaped
graph (H = 64: the shapes whose gradient kernels have a bf16-split and an fp32 form) on planted-signal batches; writes the
loss curve and the final parameters.  The kernel switches come from the environment (read once per process)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(dst):
    from hpmn_amd.hpmn import Hpmn_Industry
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(411)
    V, T, B, nb = 3000, 233, 96, int(os.environ.get("TRAJ_NB", "25"))                      # 233 + 23 = 256 steps: 256, 128, 64, 32
    batches = []
    for _ in range(nb):
        ids = rng.integers(40, V, size=(B, T, 2)).astype(np.int32)
        ids[:, :, 0] = ids[:, :1, 0] % 20 + 1           # a uid column
        label = rng.integers(0, 2, size=B).astype(np.int32)
        ids[label == 1, -2, 1] = ids[label == 1, -3, 1]  # planted: a positive's target repeats its last item
        batches.append((torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)))
    m = Hpmn_Industry(dst + ".model", [], [], V, 2, 1, T, 1, float(os.environ.get("TRAJ_LR", "0.0001")), 64, 16, 3, [2] * 10 + [1], [1], 4, 1, True, False,
                      memory_reg=5e-5, verbose=False, seed=11)
    loss = []
    for step in range(200):
        ids, label = batches[step % nb]
        _, ce = m.train_step(ids, label, keep_prob=1.0)
        loss.append(ce)
    torch.cuda.synchronize()
    out = {"loss": np.array([float(x) for x in loss])}
    # what the trained model SAYS (not where its parameters wandered: Adam moves an element by ~lr per step whatever the size of
    # its gradient, so elements with near-zero gradients random-walk apart between any two runs that differ in the last bits)
    out["pred_final"] = torch.cat([m.forward_inference(batches[i][0])["prediction"] for i in range(4)]).cpu().numpy()
    np.savez(dst, **out)


if __name__ == "__main__":
    main(sys.argv[1])
