"""Memory read path for the graphs that execute the ITEM branch (``item=True``: dual mode, or item only).

The default graph of every reference configuration (``user=True, item=False``) reads memory in one hand-written HIP
launch (csrc/read_path.hip).  With the item side on, the head sees ``repre = [user_repre, item_repre]``
(code/hpmn.py:452-462, :307-317) and two attention stacks feed it; that variant runs here as PyTorch-ROCm device
ops under autograd -- the role north_star assigns to PyTorch ("autograd / optimizer") -- while both branches'
scans, their BPTT, the embedding scatter and Adam stay on the HIP kernels.  No CPU path: tensors live on the GPU.

Every function cites the reference lines it follows (paths relative to the reference checkout).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

BN_EPS = 1e-3        # tf.layers.batch_normalization default (code/hpmn.py:190)
LOGLOSS_EPS = 1e-7   # tf.losses.log_loss default (code/hpmn.py:202)


def covreg(memory: torch.Tensor) -> torch.Tensor:
    """code/hpmn.py:161-170: sum over the batch of the Frobenius norms of the off-diagonal slot covariance."""
    H = memory.shape[2]
    c = memory - memory.mean(dim=2, keepdim=True)
    cov = torch.matmul(c, c.transpose(1, 2)) / float(H)
    cov = cov - torch.diag_embed(torch.diagonal(cov, dim1=1, dim2=2))
    return torch.sqrt((cov * cov).sum(dim=(1, 2))).sum()


def attention(p: Dict[str, torch.Tensor], scope: str, first: int, memory, query):
    """code/hpmn.py:133-146 with key = value = memory; three fresh dense layers per call (dense_{first..first+2})."""
    B, K, H = memory.shape
    q = query.unsqueeze(1).expand(B, K, H)
    inp = torch.cat([q, memory, q - memory, q * memory], dim=-1)
    d = scope + "/dense_%d/"
    fc1 = torch.relu(inp @ p[d % first + "kernel"] + p[d % first + "bias"])
    fc2 = torch.relu(fc1 @ p[d % (first + 1) + "kernel"] + p[d % (first + 1) + "bias"])
    fc3 = fc2 @ p[d % (first + 2) + "kernel"] + p[d % (first + 2) + "bias"]
    score = torch.softmax(fc3.reshape(B, K), dim=1)
    return (memory * score.unsqueeze(2)).sum(dim=1), score


def query_memory(p, scope: str, hop: int, last, memory):
    """code/hpmn.py:172-182: q = dense(last); hop x (read = attention(memory, q); q = q map + read)."""
    q = last @ p[scope + "/dense/kernel"] + p[scope + "/dense/bias"]
    w0 = None
    for h in range(hop):
        read, w = attention(p, scope, 3 * h + 1, memory, q)
        q = q @ p[scope + "/map"] + read
        if h == 0:
            w0 = w
    return q, w0


def head(p, repre, keep_prob: float, masks: Optional[Tuple[torch.Tensor, torch.Tensor]], generator=None):
    """code/hpmn.py:190-199: inference-mode batch-norm (moving mean 0 / variance 1, never updated) = an affine map,
    200 ELU, dropout, 80 ELU, dropout, 1 sigmoid.  ``masks`` = explicit 0/1 keep masks, else Bernoulli(keep_prob)."""
    bn = p["output/bn1/gamma"] * repre / math.sqrt(1.0 + BN_EPS) + p["output/bn1/beta"]
    fc1 = torch.nn.functional.elu(bn @ p["output/fc1/kernel"] + p["output/fc1/bias"])
    fc1 = _dropout(fc1, keep_prob, None if masks is None else masks[0], generator)
    fc2 = torch.nn.functional.elu(fc1 @ p["output/fc2/kernel"] + p["output/fc2/bias"])
    fc2 = _dropout(fc2, keep_prob, None if masks is None else masks[1], generator)
    logit = (fc2 @ p["output/fc3/kernel"] + p["output/fc3/bias"]).reshape(-1)
    return logit, torch.sigmoid(logit)


def _dropout(x, keep_prob, mask, generator):
    if mask is not None:
        return x * mask / keep_prob
    if keep_prob >= 1.0:
        return x
    keep = (torch.rand(x.shape, device=x.device, generator=generator) < keep_prob).to(x.dtype)
    return x * keep / keep_prob          # tf.nn.dropout: kept units scaled by 1/keep_prob


def read(p, branches: List[Tuple[str, int, torch.Tensor, torch.Tensor]], keep_prob=1.0, masks=None, generator=None):
    """branches = [(scope, hop, memory [B,K,H], last [B,D0]) ...] in the order user, item.
    -> dict(prediction, logit, memory_loss, weights = {scope: first-hop attention weights})."""
    parts, mem_loss, weights = [], 0.0, {}
    for scope, hop, memory, last in branches:
        q, w0 = query_memory(p, scope, hop, last, memory)
        parts += [q, last]                      # user_repre = concat([query, last]) (code/hpmn.py:442, :450)
        mem_loss = mem_loss + covreg(memory)    # imloss + umloss (:455, :310)
        weights[scope] = w0
    logit, pred = head(p, torch.cat(parts, dim=-1), keep_prob, masks, generator)
    return dict(prediction=pred, logit=logit, memory_loss=mem_loss, weights=weights)


def log_loss_sum(pred, label):
    """Sum over the (local) batch of the terms tf.losses.log_loss averages (code/hpmn.py:202)."""
    y = label.to(pred.dtype)
    return (-y * torch.log(pred + LOGLOSS_EPS) - (1.0 - y) * torch.log(1.0 - pred + LOGLOSS_EPS)).sum()
