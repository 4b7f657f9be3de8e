"""Worker of test_split_gradient_kernels_are_fp32_equivalent: one compute_gradients of an XLong-shaped graph at H = 64 (layer 0
D = 32: the in-loop input gradient; layers above D = 64) and at H = 128 (the bf16 projection / input-gradient / weight-gradient
kernels of configs[4]) from seeded weights of a trained model's size; writes every variable's gradient.  The kernel switches
(HPMN_WGRAD_PLANES, HPMN_DX_PLANES, HPMN_PROJ_PLANES, bench.ALL_FP32_ENV) come from the environment, read once per process."""
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
    rng = np.random.default_rng(131)
    out = {}
    for tag, H, K, T, B in (("h64", 64, 5, 489, 37), ("h128", 128, 4, 233, 21)):
        V = 2000
        ids = rng.integers(40, V, size=(B, T, 2)).astype(np.int32)
        ids[:, :, 0] = ids[:, :1, 0] % 20 + 1
        label = rng.integers(0, 2, size=B).astype(np.int32)
        m = Hpmn_Industry(dst + "." + tag, [], [], V, 2, 1, T, 1, 1e-3, H, 16, 3, [2] * 10 + [1], [1], K, 1, True, False,
                          memory_reg=5e-5, verbose=False, seed=5)
        g = torch.Generator(device="cpu").manual_seed(7)
        n_emb = m.params["Embedding/emb_mtx"].numel()
        m.flat_param[n_emb:] += 0.2 * torch.randn(m.flat_param.numel() - n_emb, generator=g).to(dev)
        res, ce = m.compute_gradients(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=1.0)
        torch.cuda.synchronize()
        out[tag + "/ce"] = np.array(float(ce))
        out[tag + "/memory"] = res["memory"].detach().cpu().numpy()
        for name in m.params:
            out[tag + "/" + name.replace("/", "|")] = m.grads[name].detach().cpu().numpy()
    np.savez(dst, **out)


if __name__ == "__main__":
    main(sys.argv[1])
