"""Worker of test_read_training_launch_on_bf16_fragments_tracks_the_fp32_launch: one compute_gradients of a small XLong-shaped
graph (K = 7 slots, H = 64: two samples x seven slots = 14 of a 16-row tile; hop = 3) and of an Amazon-shaped one (H = 32,
K = 4) from seeded weights; writes predictions, the loss and the dense variables' gradients.  HPMN_READ_BF16 (read once per
process) picks the read path's training launch."""
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
    rng = np.random.default_rng(97)
    out = {}
    for tag, H, K, T in (("xlong", 64, 7, 489), ("small", 32, 4, 41)):
        V, B = 2000, 37                                                      # (odd batch: the last workgroup holds one sample)
        ids = rng.integers(40, V, size=(B, T, 2)).astype(np.int32)
        ids[:, :, 0] = ids[:, :1, 0] % 20 + 1
        label = rng.integers(0, 2, size=B).astype(np.int32)
        m = Hpmn_Industry(dst + "." + tag, [], [], V, 2, 1, T, 1, 1e-3, H, 16, 3, [2] * 10 + [1], [1], K, 1, True, False,
                          memory_reg=5e-5, verbose=False, seed=5)
        # (weights of a trained model's size, not the initialiser's: every product carries signal)
        g = torch.Generator(device="cpu").manual_seed(7)
        n_emb = m.params["Embedding/emb_mtx"].numel()
        m.flat_param[n_emb:] += 0.2 * torch.randn(m.flat_param.numel() - n_emb, generator=g).to(dev)
        res, ce = m.compute_gradients(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=1.0)
        torch.cuda.synchronize()
        out[tag + "_pred"] = res["prediction"].cpu().numpy()
        out[tag + "_ce"] = np.array(float(ce))
        out[tag + "_grad"] = m.flat_grad[n_emb:].cpu().numpy() if m.flat_grad.numel() > n_emb else m.flat_grad.cpu().numpy()
        out[tag + "_table_grad_abs"] = np.array(float(m.table_gradient().abs().sum()))
    np.savez(dst, **out)


if __name__ == "__main__":
    main(sys.argv[1])
