"""CPU tests of the oracle itself: self-consistency between the float64 NumPy restatement and
the PyTorch restatement, algebraic properties of code/hpmn.py:113-131, hand-derived tiny
cases, the TF-form Adam, and the committed golden vectors."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import hpmn_oracle as O
from oracle import torch_restatement as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def small_cfg(industry=False, H=8, K=3, T=20, F=3, V=40):
    if industry:
        return O.HpmnConfig(feature_size=V, user_dim=F, user_maxlen=T, hidden_size=H, embedding_size=4,
                            hop=2, user_layers=(2,) * 10 + (1,), user_num_layers=K, industry=True,
                            memory_reg=5e-5)
    return O.HpmnConfig(feature_size=V, user_dim=F, user_maxlen=T, hidden_size=H, embedding_size=4, hop=3,
                        user_layers=(2, 2, 5, 5, 1), user_num_layers=K, industry=False, memory_reg=1e-5)


def rand_batch(cfg, B, seed=0, pad=True):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, cfg.feature_size, size=(B, cfg.user_maxlen, cfg.user_dim)).astype(np.int64)
    if pad:
        for b in range(B):
            ids[b, :rng.integers(0, cfg.user_maxlen)] = 0
    label = rng.integers(0, 2, size=B)
    return ids, label


@pytest.mark.parametrize("industry", [False, True])
def test_numpy_and_torch_restatements_agree(industry):
    cfg = small_cfg(industry, T=41 if industry else 20)
    p = O.randomize_params(O.init_params(cfg, seed=1), seed=2)
    ids, label = rand_batch(cfg, 5)
    a = O.forward(cfg, p, ids, label)
    b = R.forward(cfg, R.to_torch(p), torch.as_tensor(ids), torch.as_tensor(label))
    for k in ("memory", "logit", "prediction", "user_weights", "memory_loss", "cross_entropy", "query"):
        np.testing.assert_allclose(a[k], b[k].detach().numpy(), rtol=1e-10, atol=1e-12, err_msg=k)


def test_float32_restatement_close_to_float64():
    cfg = small_cfg()
    p = O.randomize_params(O.init_params(cfg, seed=1), seed=2)
    ids, label = rand_batch(cfg, 4)
    a = O.forward(cfg, p, ids, label)
    b = R.forward(cfg, R.to_torch(p, torch.float32), torch.as_tensor(ids), torch.as_tensor(label))
    np.testing.assert_allclose(a["logit"], b["logit"].numpy(), atol=1e-5)
    np.testing.assert_allclose(a["memory"], b["memory"].numpy(), atol=1e-5)


def test_gru_cell_hand_derived():
    """H=1, D=1 by hand: TF order (r,u), reset applied BEFORE the candidate matmul, bias as given."""
    x, h = np.array([[0.5]]), np.array([[0.25]])
    wg = np.array([[0.3, -0.2], [0.7, 0.1]])     # rows: x, h ; cols: r, u
    bg = np.array([1.0, 1.0])
    wc = np.array([[0.4], [-0.6]])
    bc = np.array([0.05])
    r = 1 / (1 + math.exp(-(0.5 * 0.3 + 0.25 * 0.7 + 1.0)))
    u = 1 / (1 + math.exp(-(0.5 * -0.2 + 0.25 * 0.1 + 1.0)))
    c = math.tanh(0.5 * 0.4 + (r * 0.25) * -0.6 + 0.05)
    want = u * 0.25 + (1 - u) * c
    got = O.gru_cell(x, h, wg, bg, wc, bc)
    assert abs(got[0, 0] - want) < 1e-15
    # and it is NOT the cuDNN/PyTorch variant r*(h Wc_h)
    torch_variant = u * 0.25 + (1 - u) * math.tanh(0.5 * 0.4 + r * (0.25 * -0.6) + 0.05)
    assert abs(torch_variant - want) < 1e-15  # scalar H=1: the two coincide ...
    x2, h2 = np.array([[0.5]]), np.array([[0.25, -0.5]])
    wg2 = np.array([[0.3, -0.2, 0.1, 0.2], [0.7, 0.1, -0.3, 0.4], [0.2, -0.1, 0.5, 0.6]])
    wc2 = np.array([[0.4, 0.1], [-0.6, 0.3], [0.2, -0.7]])
    g = 1 / (1 + np.exp(-(np.concatenate([x2, h2], 1) @ wg2 + 1.0)))
    r2 = g[:, :2]
    tf_c = np.tanh(np.concatenate([x2, r2 * h2], 1) @ wc2)
    cudnn_c = np.tanh(x2 @ wc2[:1] + r2 * (h2 @ wc2[1:]))
    assert np.abs(tf_c - cudnn_c).max() > 1e-3    # ... but differ as soon as H > 1
    got2 = O.gru_cell(x2, h2, wg2, np.ones(4), wc2, np.zeros(2))
    np.testing.assert_allclose(got2, g[:, 2:] * h2 + (1 - g[:, 2:]) * tf_c, atol=1e-15)


@pytest.mark.parametrize("industry", [False, True])
def test_subsample_rerun_equals_periodic_fire(industry):
    """code/hpmn.py:113-131 (K sequential scans over subsampled outputs) == one time loop in which
    layer i fires when (t+1) % prod(li[:i]) == 0."""
    cfg = small_cfg(industry, K=4, T=105 if industry else 100)    # 128 / 100 steps
    p = O.randomize_params(O.init_params(cfg, seed=3), seed=4)
    ids, _ = rand_batch(cfg, 3)
    x = O.embedding(cfg, p, ids)
    mem_a, _ = O.build_memory(cfg, p, x)
    mem_b = O.build_memory_periodic(cfg, p, x)
    np.testing.assert_allclose(mem_a, mem_b, atol=1e-14)


def test_single_layer_is_plain_gru_final_state():
    cfg = small_cfg(K=1)
    p = O.randomize_params(O.init_params(cfg, seed=3), seed=4)
    ids, _ = rand_batch(cfg, 3)
    x = O.embedding(cfg, p, ids)
    mem, _ = O.build_memory(cfg, p, x)
    _, state = O.dynamic_rnn(x, p["User/GRU0/gates/kernel"], p["User/GRU0/gates/bias"],
                             p["User/GRU0/candidate/kernel"], p["User/GRU0/candidate/bias"])
    np.testing.assert_array_equal(mem[:, 0], state)


def test_padding_steps_are_not_noops_but_sample_independent():
    """No sequence_length is passed (code/hpmn.py:119-120): the GRU runs over the zero padding, so
    the state after p pad steps is non-zero, and identical for every sample with the same p."""
    cfg = small_cfg(K=1, T=20)
    p = O.randomize_params(O.init_params(cfg, seed=3), seed=4)
    ids, _ = rand_batch(cfg, 2, pad=False)
    ids[:, :7] = 0
    x = O.embedding(cfg, p, ids)
    assert np.all(x[:, :7] == 0)                       # id 0 rows are exactly zero (mask, :417-422)
    outs, _ = O.dynamic_rnn(x, p["User/GRU0/gates/kernel"], p["User/GRU0/gates/bias"],
                            p["User/GRU0/candidate/kernel"], p["User/GRU0/candidate/bias"])
    assert np.abs(outs[:, 6]).max() > 1e-3
    np.testing.assert_array_equal(outs[0, :7], outs[1, :7])


def test_industry_embedding_has_no_mask_and_23_zero_steps():
    cfg = small_cfg(industry=True, T=41)
    p = O.randomize_params(O.init_params(cfg, seed=3), seed=4)
    ids = np.zeros((2, 41, 3), dtype=np.int64)
    x = O.embedding(cfg, p, ids)
    assert x.shape == (2, 64, 12)
    assert np.all(x[:, :23] == 0)
    np.testing.assert_array_equal(x[0, 23, :4], p["Embedding/emb_mtx"][0])   # id 0 is a real row here
    assert cfg.last_index == -2 and cfg.layer_lengths() == [64, 32, 16]


def test_layer_lengths_reference_configs():
    amazon = O.HpmnConfig(10, 3, 100, 32, user_layers=(2, 2, 5, 5, 1), user_num_layers=4)
    assert amazon.layer_lengths() == [100, 50, 25, 5]
    taobao = O.HpmnConfig(10, 4, 300, 64, user_layers=(2, 2, 3, 5, 5, 1), user_num_layers=5)
    assert taobao.layer_lengths() == [300, 150, 75, 25, 5]
    xlong = O.HpmnConfig(10, 2, 1001, 64, user_layers=(2,) * 10 + (1,), user_num_layers=7, industry=True)
    assert xlong.layer_lengths() == [1024, 512, 256, 128, 64, 32, 16]
    bad = O.HpmnConfig(10, 3, 100, 32, user_layers=(3, 2), user_num_layers=2)
    with pytest.raises(ValueError):
        bad.layer_lengths()


def test_bn_is_inference_affine_and_loss_terms():
    cfg = small_cfg()
    p = O.randomize_params(O.init_params(cfg, seed=1), seed=2)
    ids, label = rand_batch(cfg, 6)
    out = O.forward(cfg, p, ids, label)
    # log_loss is a MEAN over the batch, memory_loss a SUM (code/hpmn.py:170,202,207)
    pred = out["prediction"]
    ll = np.mean(-label * np.log(pred + 1e-7) - (1 - label) * np.log(1 - pred + 1e-7))
    assert abs(out["log_loss"] - ll) < 1e-15
    per_sample = [O.get_covreg(out["memory"][b:b + 1]) for b in range(6)]
    assert abs(out["memory_loss"] - sum(per_sample)) < 1e-12
    assert abs(out["cross_entropy"] - (ll + cfg.memory_reg * out["memory_loss"])) < 1e-15


def test_adam_tf_form_matches_closed_form_and_torch():
    p = {"w": np.array([1.0, -2.0, 3.0])}
    st = O.AdamState()
    g = {"w": np.array([0.5, -3.0, 0.0])}       # -3 is clipped to -1 (code/hpmn.py:212)
    O.adam_step(p, g, st, lr=0.01)
    gc = np.array([0.5, -1.0, 0.0])
    m, v = 0.1 * gc, 0.001 * gc ** 2
    lr_t = 0.01 * math.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p["w"], np.array([1.0, -2.0, 3.0]) - lr_t * m / (np.sqrt(v) + 1e-8), atol=1e-15)
    # second step + torch restatement
    tp = {"w": torch.tensor([1.0, -2.0, 3.0], dtype=torch.float64)}
    opt = R.TFAdam(tp, 0.01)
    opt.step({"w": torch.tensor([0.5, -3.0, 0.0], dtype=torch.float64)})
    O.adam_step(p, g, st, lr=0.01)
    opt.step({"w": torch.tensor([0.5, -3.0, 0.0], dtype=torch.float64)})
    np.testing.assert_allclose(p["w"], tp["w"].numpy(), atol=1e-15)
    # a zero-gradient element still moves while its m != 0 (dense Adam over the table)
    before = p["w"].copy()
    O.adam_step(p, {"w": np.zeros(3)}, st, lr=0.01)
    assert abs(p["w"][0] - before[0]) > 0 and p["w"][2] == before[2]


@pytest.mark.parametrize("name,industry", [("oracle_c0.npz", False), ("oracle_industry.npz", True)])
def test_golden_vectors_reproduce(name, industry):
    z = np.load(os.path.join(GOLD, name))
    p = {k[len("param:"):]: z[k].astype(np.float64) for k in z.files if k.startswith("param:")}
    if industry:
        cfg = O.HpmnConfig(feature_size=500, user_dim=2, user_maxlen=41, hidden_size=64, embedding_size=16,
                           hop=3, user_layers=(2,) * 10 + (1,), user_num_layers=4, industry=True,
                           memory_reg=5e-5)
    else:
        cfg = O.HpmnConfig(feature_size=300, user_dim=3, user_maxlen=100, hidden_size=32, embedding_size=16,
                           hop=3, user_layers=(2, 2, 5, 5, 1), user_num_layers=3, memory_reg=1e-5)
    out = O.forward(cfg, p, z["ids"], z["label"])
    for k in ("memory", "logit", "prediction", "user_weights", "memory_loss", "cross_entropy"):
        np.testing.assert_allclose(out[k], z[k], rtol=1e-12, atol=1e-13, err_msg=k)


def test_torch_train_step_decreases_loss():
    cfg = small_cfg()
    p = R.to_torch(O.init_params(cfg, seed=1), torch.float64)
    ids, label = rand_batch(cfg, 16)
    ids_t, label_t = torch.as_tensor(ids), torch.as_tensor(label)
    opt = R.TFAdam(p, 0.01)
    first = None
    for _ in range(30):
        out, _ = R.train_step(cfg, p, opt, ids_t, label_t)
        first = first if first is not None else float(out["cross_entropy"])
    assert float(out["cross_entropy"]) < first
