"""PyTorch-CPU eager restatement of the HPMN hot path (TEST INFRASTRUCTURE ONLY).

Two jobs, both on the checker side of the fence (see ``oracle/__init__.py``):

* gradient oracle: run in float64 under ``torch.autograd`` to check the HIP backward
  kernels (SURVEY.md section 4 "gradient parity");
* ``cpu_baseline`` of ``bench.py``: run in float32, step by step exactly like the
  reference graph executes (one small op per TF op per time step, K sequential
  scans, dense TF-form Adam over every variable) -- BASELINE.md section 3.

PARITY UNPINNED: same caveat as ``hpmn_oracle.py``; it must agree with that file
(tests/test_oracle.py) and cites the same reference lines.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .hpmn_oracle import BN_EPS, LOGLOSS_EPS, HpmnConfig


def to_torch(params, dtype=torch.float64, requires_grad=False) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in params.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def embedding(cfg: HpmnConfig, p, user_inp: torch.Tensor) -> torch.Tensor:
    # code/hpmn.py:414-423 / :266-276, zero prefix :288-289
    emb = p["Embedding/emb_mtx"]
    B, T, F = user_inp.shape
    rows = torch.nn.functional.embedding(user_inp, emb)
    if cfg.mask_id0:
        rows = rows * (user_inp != 0).unsqueeze(-1).to(emb.dtype)
    x = rows.reshape(B, T, F * cfg.embedding_size)
    z = cfg.front_zero_steps
    if z:
        x = torch.cat([torch.zeros(B, z, x.shape[2], dtype=x.dtype), x], dim=1)
    return x


def gru_cell(x, h, wg, bg, wc, bc):
    # code/util.py:81-110 without line 108
    H = h.shape[1]
    g = torch.sigmoid(torch.cat([x, h], dim=1) @ wg + bg)
    r, u = g[:, :H], g[:, H:]
    c = torch.tanh(torch.cat([x, r * h], dim=1) @ wc + bc)
    return u * h + (1.0 - u) * c


def dynamic_rnn(x, wg, bg, wc, bc):
    # code/hpmn.py:119-120; code/rnn.py:583-588,754-768,796
    B, T, _ = x.shape
    h = torch.zeros(B, bc.shape[0], dtype=x.dtype)
    xt = x.transpose(0, 1)           # time-major like code/rnn.py:560-563
    outs = []
    for t in range(T):
        h = gru_cell(xt[t], h, wg, bg, wc, bc)
        outs.append(h)
    return torch.stack(outs, dim=1), h


def get_covreg(memory):
    # code/hpmn.py:161-170
    H = memory.shape[2]
    c = memory - memory.mean(dim=2, keepdim=True)
    cov = torch.matmul(c, c.transpose(1, 2)) / float(H)
    cov = cov - torch.diag_embed(torch.diagonal(cov, dim1=1, dim2=2))
    return torch.sqrt((cov * cov).sum(dim=(1, 2))).sum()


def build_memory(cfg: HpmnConfig, p, inp):
    # code/hpmn.py:113-131
    mem = []
    for i in range(cfg.user_num_layers):
        pre = "%s/GRU%d/" % (cfg.scope, i)
        outs, state = dynamic_rnn(inp, p[pre + "gates/kernel"], p[pre + "gates/bias"],
                                  p[pre + "candidate/kernel"], p[pre + "candidate/bias"])
        mem.append(state.unsqueeze(1))
        li = cfg.user_layers[i]
        B, T, H = outs.shape
        inp = outs.reshape(B, T // li, li, H)[:, :, li - 1, :]
    memory = torch.cat(mem, dim=1)
    return memory, get_covreg(memory)


def attention(p, first_dense, memory, query, scope="User"):
    # code/hpmn.py:133-146
    B, K, H = memory.shape
    q = query.unsqueeze(1).expand(B, K, H)
    inp = torch.cat([q, memory, q - memory, q * memory], dim=-1)
    n = first_dense
    d = scope + "/dense_%d/"
    fc1 = torch.relu(inp @ p[d % n + "kernel"] + p[d % n + "bias"])
    fc2 = torch.relu(fc1 @ p[d % (n + 1) + "kernel"] + p[d % (n + 1) + "bias"])
    fc3 = fc2 @ p[d % (n + 2) + "kernel"] + p[d % (n + 2) + "bias"]
    score = torch.softmax(fc3.reshape(B, K), dim=1)
    return (memory * score.unsqueeze(2)).sum(dim=1), score


def query_memory(cfg, p, last, memory):
    # code/hpmn.py:172-182
    sc = cfg.scope
    q = last @ p[sc + "/dense/kernel"] + p[sc + "/dense/bias"]
    w0 = None
    for hop in range(cfg.hop):
        read, w = attention(p, 3 * hop + 1, memory, q, sc)
        q = q @ p[sc + "/map"] + read
        if hop == 0:
            w0 = w
    return q, w0


def fc_net(p, repre, mask1=None, mask2=None, keep_prob=1.0):
    # code/hpmn.py:190-199
    bn = p["output/bn1/gamma"] * repre / math.sqrt(1.0 + BN_EPS) + p["output/bn1/beta"]
    fc1 = torch.nn.functional.elu(bn @ p["output/fc1/kernel"] + p["output/fc1/bias"])
    if mask1 is not None:
        fc1 = fc1 * mask1 / keep_prob
    fc2 = torch.nn.functional.elu(fc1 @ p["output/fc2/kernel"] + p["output/fc2/bias"])
    if mask2 is not None:
        fc2 = fc2 * mask2 / keep_prob
    logit = (fc2 @ p["output/fc3/kernel"] + p["output/fc3/bias"]).reshape(-1)
    return logit, torch.sigmoid(logit)


def forward(cfg: HpmnConfig, p, user_inp, label=None, mask1=None, mask2=None, keep_prob=1.0):
    uinp = embedding(cfg, p, user_inp)
    memory, mem_loss = build_memory(cfg, p, uinp)
    last = uinp[:, cfg.last_index, :]
    q, w0 = query_memory(cfg, p, last, memory)
    repre = torch.cat([q, last], dim=-1)
    logit, pred = fc_net(p, repre, mask1, mask2, keep_prob)
    out = dict(uinp=uinp, memory=memory, memory_loss=mem_loss, query=q, user_weights=w0,
               logit=logit, prediction=pred)
    if label is not None:
        y = label.to(pred.dtype)
        ll = torch.mean(-y * torch.log(pred + LOGLOSS_EPS) - (1.0 - y) * torch.log(1.0 - pred + LOGLOSS_EPS))
        out["log_loss"] = ll
        ce = ll + cfg.memory_reg * mem_loss
        if cfg.l2_reg:
            ce = ce + cfg.l2_reg * sum(0.5 * (v * v).sum() for v in p.values())
        out["cross_entropy"] = ce
    return out


def forward_dual(ucfg: HpmnConfig, icfg: HpmnConfig, p, user_inp, item_inp, label=None, user=True, item=True,
                 mask1=None, mask2=None, keep_prob=1.0):
    """build_graph with the item side executed (code/hpmn.py:444-462 / :297-317): both branches share the embedding
    table, repre = [user_repre, item_repre] (dual) or item_repre alone, memory_loss = imloss + umloss / imloss."""
    parts, mem_loss, out = [], 0.0, {}
    for on, cfg, inp, tag in ((user, ucfg, user_inp, "user"), (item, icfg, item_inp, "item")):
        if not on:
            continue
        x = embedding(cfg, p, inp)
        memory, ml = build_memory(cfg, p, x)
        last = x[:, cfg.last_index, :]
        q, w0 = query_memory(cfg, p, last, memory)
        parts += [q, last]
        mem_loss = mem_loss + ml
        out[tag + "_memory"], out[tag + "_weights"] = memory, w0
    logit, pred = fc_net(p, torch.cat(parts, dim=-1), mask1, mask2, keep_prob)
    out.update(memory_loss=mem_loss, logit=logit, prediction=pred)
    if label is not None:
        y = label.to(pred.dtype)
        ll = torch.mean(-y * torch.log(pred + LOGLOSS_EPS) - (1.0 - y) * torch.log(1.0 - pred + LOGLOSS_EPS))
        out["log_loss"] = ll
        out["cross_entropy"] = ll + ucfg.memory_reg * mem_loss
    return out


class TFAdam:
    """code/hpmn.py:209-214: clip_by_value(g,-1,1) then TF-form Adam, dense over all
    variables (embedding table included)."""

    def __init__(self, params: Dict[str, torch.Tensor], lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = params, lr, beta1, beta2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, p in self.p.items():
            g = grads.get(k)
            if g is None:
                g = torch.zeros_like(p)
            g = g.clamp(-1.0, 1.0)
            self.m[k].mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))


def train_step(cfg: HpmnConfig, p, opt: TFAdam, user_inp, label, mask1=None, mask2=None, keep_prob=1.0):
    """One sess.run(train_step) (code/hpmn.py:482): forward, backward, clip, dense Adam."""
    for v in p.values():
        v.requires_grad_(True)
        v.grad = None
    out = forward(cfg, p, user_inp, label, mask1, mask2, keep_prob)
    out["cross_entropy"].backward()
    grads = {k: v.grad for k, v in p.items()}
    opt.step(grads)
    return out, grads
