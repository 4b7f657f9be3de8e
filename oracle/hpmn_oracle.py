"""NumPy float64 restatement of the HPMN hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (see ``oracle/__init__.py``): TensorFlow 1.4 cannot be run
here and the reference holds no golden vectors; every function below cites the
reference lines it restates instead.

All citations are relative to ``/root/reference/``.

Variable layout follows what TF1.4 would create for ``code/hpmn.py``:

====================================  =====================  ==========================
name                                  shape                  reference
====================================  =====================  ==========================
``Embedding/emb_mtx``                 [V, E]                 code/hpmn.py:415-416
``User/GRU{i}/gates/kernel``          [D_i+H, 2H]            code/util.py:88-93 (_Linear)
``User/GRU{i}/gates/bias``            [2H]  (init 1.0)       code/util.py:84-86
``User/GRU{i}/candidate/kernel``      [D_i+H, H]             code/util.py:99-106
``User/GRU{i}/candidate/bias``        [H]   (init 0.0)       code/util.py:99-106
``User/dense/{kernel,bias}``          [D0, H], [H]           code/hpmn.py:173
``User/map``                          [H, H]                 code/hpmn.py:174
``User/dense_{3h+1..3h+3}/...``       4H->80->40->1          code/hpmn.py:137-139 (hop h)
``output/bn1/{gamma,beta}``           [H+D0]                 code/hpmn.py:190
``output/fc{1,2,3}/{kernel,bias}``    (H+D0)->200->80->1     code/hpmn.py:191-195
====================================  =====================  ==========================
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

BN_EPS = 1e-3          # tf.layers.batch_normalization default epsilon (code/hpmn.py:190)
LOGLOSS_EPS = 1e-7     # tf.losses.log_loss default epsilon (code/hpmn.py:202)
ATT_FC1, ATT_FC2 = 80, 40      # code/hpmn.py:137-138
HEAD_FC1, HEAD_FC2 = 200, 80   # code/hpmn.py:191,193


@dataclass
class HpmnConfig:
    """Shapes/constants of one HPMN "User" branch (code/hpmn.py:218-239, 576-663)."""
    feature_size: int            # V
    user_dim: int                # F  ids per step
    user_maxlen: int             # T  steps fed by the loader
    hidden_size: int             # H
    embedding_size: int = 16     # E
    hop: int = 3
    user_layers: Sequence[int] = (2, 2, 5, 5, 1)   # li_layer
    user_num_layers: int = 3     # K
    industry: bool = False       # Hpmn_Industry graph (code/hpmn.py:284-320) vs Hpmn (:432-465)
    memory_reg: float = 1e-5
    l2_reg: float = 0.0
    # the same container describes an ITEM-side branch (code/hpmn.py:444-450 scope "item", :297-305 scope "Item"):
    # user_dim/user_maxlen/user_layers/user_num_layers then hold item_dim/item_maxlen/item_layers/item_num_layers
    scope: str = "User"
    front_zero: int = -1         # override; Hpmn_Industry's item side pads 192 - 184 = 8 zero steps (:298-299)
    last_idx: int = 0            # override; every item side reads iinp[:, -1, :] (:302, :447)

    # --- derived -----------------------------------------------------------
    @property
    def front_zero_steps(self) -> int:
        # code/hpmn.py:288-290: 23 zero steps in front, build_memory(..., 1024, ...)
        if self.front_zero >= 0:
            return self.front_zero
        return 23 if self.industry else 0

    @property
    def scan_len(self) -> int:
        return self.user_maxlen + self.front_zero_steps

    @property
    def mask_id0(self) -> bool:
        # Hpmn multiplies rows by mask_lookup_table (0 for id 0) code/hpmn.py:417-422;
        # Hpmn_Industry does a bare lookup code/hpmn.py:274-276.
        return not self.industry

    @property
    def last_index(self) -> int:
        # code/hpmn.py:439 (uinp[:, -1, :]) vs :292 (uinp[:, -2, :])
        if self.last_idx:
            return self.last_idx
        return -2 if self.industry else -1

    @property
    def d0(self) -> int:
        return self.user_dim * self.embedding_size

    def layer_lengths(self) -> List[int]:
        """T_i of code/hpmn.py:122-128 (maxlen /= li_layer[i])."""
        out, t = [], self.scan_len
        for i in range(self.user_num_layers):
            out.append(t)
            if t % self.user_layers[i] != 0:
                raise ValueError("layer %d: length %d not divisible by %d (tf.reshape at "
                                 "code/hpmn.py:124 would fail)" % (i, t, self.user_layers[i]))
            t //= self.user_layers[i]
        return out

    def layer_in_dims(self) -> List[int]:
        return [self.d0] + [self.hidden_size] * (self.user_num_layers - 1)


# ---------------------------------------------------------------------------
# parameter container
# ---------------------------------------------------------------------------
def branch_shapes(cfg: HpmnConfig) -> Dict[str, Tuple[int, ...]]:
    """Variables of one branch (scope ``cfg.scope``): its GRU stack, query projection, map, attention MLPs."""
    H, D0, sc = cfg.hidden_size, cfg.d0, cfg.scope
    shp: Dict[str, Tuple[int, ...]] = {}
    for i, d in enumerate(cfg.layer_in_dims()):
        shp["%s/GRU%d/gates/kernel" % (sc, i)] = (d + H, 2 * H)
        shp["%s/GRU%d/gates/bias" % (sc, i)] = (2 * H,)
        shp["%s/GRU%d/candidate/kernel" % (sc, i)] = (d + H, H)
        shp["%s/GRU%d/candidate/bias" % (sc, i)] = (H,)
    shp[sc + "/dense/kernel"] = (D0, H)
    shp[sc + "/dense/bias"] = (H,)
    shp[sc + "/map"] = (H, H)
    n = 1
    for _ in range(cfg.hop):
        for fin, fout in ((4 * H, ATT_FC1), (ATT_FC1, ATT_FC2), (ATT_FC2, 1)):
            shp["%s/dense_%d/kernel" % (sc, n)] = (fin, fout)
            shp["%s/dense_%d/bias" % (sc, n)] = (fout,)
            n += 1
    return shp


def head_shapes(width: int) -> Dict[str, Tuple[int, ...]]:
    shp = {"output/bn1/gamma": (width,), "output/bn1/beta": (width,)}
    for name, fin, fout in (("fc1", width, HEAD_FC1), ("fc2", HEAD_FC1, HEAD_FC2), ("fc3", HEAD_FC2, 1)):
        shp["output/%s/kernel" % name] = (fin, fout)
        shp["output/%s/bias" % name] = (fout,)
    return shp


def param_shapes(cfg: HpmnConfig, item_cfg: "HpmnConfig | None" = None, user: bool = True) -> Dict[str, Tuple[int, ...]]:
    """Variables that are EXECUTED: the user branch (default), the item branch, or both (code/hpmn.py:452-462:
    repre = [user_repre, item_repre] / user_repre / item_repre)."""
    shp: Dict[str, Tuple[int, ...]] = {"Embedding/emb_mtx": (cfg.feature_size, cfg.embedding_size)}
    width = 0
    if user:
        shp.update(branch_shapes(cfg))
        width += cfg.hidden_size + cfg.d0
    if item_cfg is not None:
        shp.update(branch_shapes(item_cfg))
        width += item_cfg.hidden_size + item_cfg.d0
    shp.update(head_shapes(width))
    return shp


def init_params(cfg: HpmnConfig, seed: int = 0, emb_init: np.ndarray | None = None,
                dtype=np.float64, item_cfg: "HpmnConfig | None" = None, user: bool = True) -> Dict[str, np.ndarray]:
    """TF1.4 default initialisers: glorot-uniform for kernels / get_variable without
    initializer, zeros for dense biases, ones for the GRU gate bias (code/util.py:84-86),
    gamma=1 / beta=0 for batch-norm.  The RNG stream is ours (TF's is unseeded,
    code/hpmn.py:13 seeds only ``random``)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg, item_cfg, user).items():
        if name == "Embedding/emb_mtx" and emb_init is not None:
            assert tuple(emb_init.shape) == shape
            out[name] = np.array(emb_init, dtype=dtype)
        elif name.endswith("gates/bias") or name.endswith("bn1/gamma"):
            out[name] = np.ones(shape, dtype=dtype)
        elif name.endswith("bias") or name.endswith("beta"):
            out[name] = np.zeros(shape, dtype=dtype)
        else:
            fan_in, fan_out = shape[0], shape[1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-lim, lim, size=shape).astype(dtype)
    return out


def randomize_params(params: Dict[str, np.ndarray], seed: int = 1, scale: float = 0.3):
    """Perturb *every* variable (biases, gamma, beta too) so parity tests do not pass
    by accident on zero biases / unit gammas."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in params.items():
        if k == "Embedding/emb_mtx":
            out[k] = rng.normal(0.0, 0.5, size=v.shape).astype(v.dtype)
        elif v.ndim == 1:
            out[k] = (v + rng.normal(0.0, scale, size=v.shape)).astype(v.dtype)
        else:
            out[k] = (v * 1.5 + rng.normal(0.0, 0.05, size=v.shape)).astype(v.dtype)
    return out


# ---------------------------------------------------------------------------
# forward restatement
# ---------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def embedding(cfg: HpmnConfig, p, user_inp: np.ndarray) -> np.ndarray:
    """code/hpmn.py:414-423 (Hpmn, masked) / :266-276 (Industry, bare) and the zero
    prefix of :288-289.  Returns uinp [B, scan_len, F*E]."""
    ids = np.asarray(user_inp)
    B, T, F = ids.shape
    assert T == cfg.user_maxlen and F == cfg.user_dim
    emb = p["Embedding/emb_mtx"]
    rows = emb[ids]                                   # [B,T,F,E]
    if cfg.mask_id0:
        rows = rows * (ids != 0)[..., None].astype(emb.dtype)
    x = rows.reshape(B, T, F * cfg.embedding_size)
    z = cfg.front_zero_steps
    if z:
        x = np.concatenate([np.zeros((B, z, x.shape[2]), dtype=x.dtype), x], axis=1)
    return x


def gru_cell(x, h, wg, bg, wc, bc):
    """TF1.4 GRUCell.call, mirrored at code/util.py:81-110 without line 108:
    [r,u] = sigmoid([x,h] Wg + bg); c = tanh([x, r*h] Wc + bc); h' = u*h + (1-u)*c."""
    H = h.shape[1]
    g = _sigmoid(np.concatenate([x, h], axis=1) @ wg + bg)
    r, u = g[:, :H], g[:, H:]
    c = np.tanh(np.concatenate([x, r * h], axis=1) @ wc + bc)
    return u * h + (1.0 - u) * c


def dynamic_rnn(x, wg, bg, wc, bc):
    """tf.nn.dynamic_rnn without sequence_length (code/hpmn.py:119-120; loop mirrored at
    code/rnn.py:583-588 zero state, :754-768 no masking, :773-774,796 stacked outputs)."""
    B, T, _ = x.shape
    H = bc.shape[0]
    h = np.zeros((B, H), dtype=x.dtype)
    outs = np.empty((B, T, H), dtype=x.dtype)
    for t in range(T):
        h = gru_cell(x[:, t, :], h, wg, bg, wc, bc)
        outs[:, t, :] = h
    return outs, h


def get_covreg(memory):
    """code/hpmn.py:161-170 (sum over the batch of Frobenius norms of the off-diagonal
    covariance between memory slots)."""
    H = memory.shape[2]
    c = memory - memory.mean(axis=2, keepdims=True)
    cov = np.einsum("bkh,bjh->bkj", c, c) / float(H)
    k = cov.shape[1]
    cov = cov * (1.0 - np.eye(k, dtype=cov.dtype))
    return np.sqrt((cov ** 2).sum(axis=(1, 2))).sum()


def build_memory(cfg: HpmnConfig, p, inp, return_outputs: bool = False):
    """code/hpmn.py:113-131: layer i is a plain GRU over every li_layer[i-1]-th output
    of layer i-1; memory[:, i] = final state of layer i."""
    mem, all_outs = [], []
    lens = cfg.layer_lengths()
    for i in range(cfg.user_num_layers):
        assert inp.shape[1] == lens[i]
        pre = "User/GRU%d/" % i
        outs, state = dynamic_rnn(inp, p[pre + "gates/kernel"], p[pre + "gates/bias"],
                                  p[pre + "candidate/kernel"], p[pre + "candidate/bias"])
        mem.append(state[:, None, :])
        all_outs.append(outs)
        li = cfg.user_layers[i]
        B, T, H = outs.shape
        inp = outs.reshape(B, T // li, li, H)[:, :, li - 1, :]    # :124-128
    memory = np.concatenate(mem, axis=1)
    loss = get_covreg(memory)
    if return_outputs:
        return memory, loss, all_outs
    return memory, loss


def attention(p, first_dense: int, memory, query):
    """code/hpmn.py:133-146 with key = value = memory; three *fresh* dense layers per
    call (un-named -> dense_{n}, n = first_dense..first_dense+2)."""
    B, K, H = memory.shape
    q = np.broadcast_to(query[:, None, :], (B, K, H))
    inp = np.concatenate([q, memory, q - memory, q * memory], axis=-1)
    n = first_dense
    fc1 = np.maximum(inp @ p["User/dense_%d/kernel" % n] + p["User/dense_%d/bias" % n], 0.0)
    fc2 = np.maximum(fc1 @ p["User/dense_%d/kernel" % (n + 1)] + p["User/dense_%d/bias" % (n + 1)], 0.0)
    fc3 = fc2 @ p["User/dense_%d/kernel" % (n + 2)] + p["User/dense_%d/bias" % (n + 2)]
    s = fc3.reshape(B, K)
    s = s - s.max(axis=1, keepdims=True)
    e = np.exp(s)
    score = e / e.sum(axis=1, keepdims=True)
    return (memory * score[:, :, None]).sum(axis=1), score


def query_memory(cfg: HpmnConfig, p, last, memory):
    """code/hpmn.py:172-182."""
    q = last @ p["User/dense/kernel"] + p["User/dense/bias"]
    weights = []
    for hop in range(cfg.hop):
        read, w = attention(p, 3 * hop + 1, memory, q)
        q = q @ p["User/map"] + read
        weights.append(w)
    return q, weights[0]


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0.0)))


def fc_net(p, repre, mask1=None, mask2=None, keep_prob: float = 1.0):
    """code/hpmn.py:190-199.  batch_normalization is called with training=False so it is
    the affine map gamma*x/sqrt(1+eps)+beta (moving stats never updated).  Dropout masks
    (0/1) are injected so training-mode parity is testable; None == keep_prob 1."""
    bn = p["output/bn1/gamma"] * repre / math.sqrt(1.0 + BN_EPS) + p["output/bn1/beta"]
    fc1 = _elu(bn @ p["output/fc1/kernel"] + p["output/fc1/bias"])
    if mask1 is not None:
        fc1 = fc1 * mask1 / keep_prob
    fc2 = _elu(fc1 @ p["output/fc2/kernel"] + p["output/fc2/bias"])
    if mask2 is not None:
        fc2 = fc2 * mask2 / keep_prob
    logit = (fc2 @ p["output/fc3/kernel"] + p["output/fc3/bias"]).reshape(-1)
    return logit, _sigmoid(logit)


def log_loss(label, pred):
    """tf.losses.log_loss (code/hpmn.py:202): batch mean with eps 1e-7."""
    y = np.asarray(label, dtype=pred.dtype)
    return np.mean(-y * np.log(pred + LOGLOSS_EPS) - (1.0 - y) * np.log(1.0 - pred + LOGLOSS_EPS))


def forward(cfg: HpmnConfig, p, user_inp, label=None, mask1=None, mask2=None, keep_prob=1.0):
    """Hpmn.build_graph 'User' branch (code/hpmn.py:436-442 / :287-295) + build_fc_net.
    Returns a dict with every tensor the parity tests compare."""
    uinp = embedding(cfg, p, user_inp)
    memory, mem_loss = build_memory(cfg, p, uinp)
    last = uinp[:, cfg.last_index, :]
    q, w0 = query_memory(cfg, p, last, memory)
    repre = np.concatenate([q, last], axis=-1)
    logit, pred = fc_net(p, repre, mask1, mask2, keep_prob)
    out = dict(uinp=uinp, memory=memory, memory_loss=mem_loss, query=q, user_weights=w0,
               logit=logit, prediction=pred)
    if label is not None:
        ll = log_loss(label, pred)
        l2 = sum(0.5 * float((v ** 2).sum()) for v in p.values())  # tf.nn.l2_loss, :204-205
        out["log_loss"] = ll
        out["cross_entropy"] = ll + cfg.l2_reg * l2 + cfg.memory_reg * mem_loss  # :203-207
    return out


# ---------------------------------------------------------------------------
# property-test helper: the single-loop "periodic fire" formulation
# ---------------------------------------------------------------------------
def build_memory_periodic(cfg: HpmnConfig, p, inp):
    """Same result as build_memory but as ONE time loop in which layer i fires when
    (t+1) % prod(li[:i]) == 0 -- the formulation the north_star describes; used by the
    tests to pin 'subsample-then-rerun == periodic fire' (SURVEY.md section 4)."""
    B = inp.shape[0]
    H, K = cfg.hidden_size, cfg.user_num_layers
    periods = [1]
    for i in range(1, K):
        periods.append(periods[-1] * cfg.user_layers[i - 1])
    h = [np.zeros((B, H), dtype=inp.dtype) for _ in range(K)]
    w = [tuple(p["User/GRU%d/%s" % (i, n)] for n in
               ("gates/kernel", "gates/bias", "candidate/kernel", "candidate/bias")) for i in range(K)]
    for t in range(inp.shape[1]):
        x = inp[:, t, :]
        for i in range(K):
            if (t + 1) % periods[i] != 0:
                break
            h[i] = gru_cell(x, h[i], *w[i])
            x = h[i]
    return np.stack(h, axis=1)


# ---------------------------------------------------------------------------
# optimiser restatement
# ---------------------------------------------------------------------------
@dataclass
class AdamState:
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)
    t: int = 0


def adam_step(p, grads, state: AdamState, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """code/hpmn.py:209-214: per-element clip_by_value(g,-1,1) then tf.train.AdamOptimizer
    (TF form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t*m/(sqrt(v)+eps)).  Dense over
    every variable including the whole embedding table."""
    state.t += 1
    t = state.t
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k in p:
        g = np.clip(grads[k], -1.0, 1.0)
        m = state.m.get(k)
        if m is None:
            m = state.m[k] = np.zeros_like(p[k])
            state.v[k] = np.zeros_like(p[k])
        v = state.v[k]
        m *= beta1
        m += (1.0 - beta1) * g
        v *= beta2
        v += (1.0 - beta2) * g * g
        p[k] = p[k] - lr_t * m / (np.sqrt(v) + eps)
    return p
