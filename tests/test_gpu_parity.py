"""Parity tests proper (run on the MI355X box with ``-m gpu``): every HIP entry point of
libhpmn_hip.so, called through the C ABI, against the CPU oracle on the same seeded inputs and
against the committed golden vectors.  Tolerances: 1e-4 absolute on logits/predictions/memory
(the north_star bar), tighter where the arithmetic allows; integer paths exact."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import hpmn_oracle as O
from oracle import torch_restatement as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    from hpmn_amd import _lib, build
    build.build_library()
    _lib.load()
    return torch.device("cuda:0")


def _needs_legacy_build():
    """Kernel generations that were measured slower and retired behind -DHPMN_LEGACY_KERNELS (r4):
    their tests run against a library built with HPMN_HIPCC_FLAGS=-DHPMN_LEGACY_KERNELS."""
    from hpmn_amd import _lib
    if not _lib.load().hpmn_has_legacy_kernels():
        pytest.skip("needs a library built with -DHPMN_LEGACY_KERNELS")


def make_model(cfg: O.HpmnConfig, tmp, params=None, lr=0.003):
    from hpmn_amd.hpmn import Hpmn, Hpmn_Industry
    cls = Hpmn_Industry if cfg.industry else Hpmn
    m = cls(str(tmp), [], [], cfg.feature_size, cfg.user_dim, 2, cfg.user_maxlen, 10, lr, cfg.hidden_size,
            cfg.embedding_size, cfg.hop, list(cfg.user_layers), [2, 1], cfg.user_num_layers, 1, True, False,
            l2_reg=cfg.l2_reg, memory_reg=cfg.memory_reg, verbose=False)
    if params is not None:
        m.set_params(params)
    return m


def cfg_amazon(H=32, K=3, T=100, F=3, V=300, E=16):
    return O.HpmnConfig(feature_size=V, user_dim=F, user_maxlen=T, hidden_size=H, embedding_size=E, hop=3,
                        user_layers=(2, 2, 5, 5, 1), user_num_layers=K, industry=False, memory_reg=1e-5)


def cfg_industry(H=64, K=4, T=41, F=2, V=500):
    return O.HpmnConfig(feature_size=V, user_dim=F, user_maxlen=T, hidden_size=H, embedding_size=16, hop=3,
                        user_layers=(2,) * 10 + (1,), user_num_layers=K, industry=True, memory_reg=5e-5)


def rand_ids(cfg, B, seed, ragged=True):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1 if not cfg.industry else 0, cfg.feature_size,
                       size=(B, cfg.user_maxlen, cfg.user_dim)).astype(np.int32)
    if ragged and not cfg.industry:
        for b in range(B):
            ids[b, :rng.integers(0, cfg.user_maxlen)] = 0
    return ids, rng.integers(0, 2, size=B).astype(np.int32)


def f32_params(cfg, seed):
    p = O.randomize_params(O.init_params(cfg, seed=seed), seed=seed + 1)
    return {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}


# ------------------------------------------------------------------------------- gather
@pytest.mark.parametrize("mask", [True, False])
def test_embed_gather_exact(dev, mask):
    from hpmn_amd import ops
    rng = np.random.default_rng(0)
    emb = rng.normal(size=(97, 16)).astype(np.float32)
    ids = rng.integers(0, 97, size=(5, 33, 3)).astype(np.int32)
    ids[0, :10] = 0
    got = ops.embed_gather(torch.as_tensor(ids).to(dev), torch.as_tensor(emb).to(dev), mask).cpu().numpy()
    want = emb[ids]
    if mask:
        want = want * (ids != 0)[..., None]
    np.testing.assert_array_equal(got, want.reshape(5, 33, 48))      # pure data movement: bit-exact


# ------------------------------------------------------------------------------- forward
CASES = [
    ("amazon_c0", cfg_amazon(), 6),
    ("amazon_c1_k4", cfg_amazon(K=4), 7),                 # odd B with 2 sequences per wave
    ("amazon_b1", cfg_amazon(), 1),
    ("taobao_like", O.HpmnConfig(400, 4, 300, 64, 16, 3, (2, 2, 3, 5, 5, 1), 5, False, 1e-5), 3),
    ("h64_d48", cfg_amazon(H=64, K=2, T=20), 5),
    ("h32_d32", cfg_amazon(H=32, K=3, T=20, F=2), 4),
    ("industry_64", cfg_industry(), 5),
    ("industry_k7_1001", cfg_industry(H=64, K=7, T=1001, V=800), 3),   # XLong graph at full length
    ("industry_h32", cfg_industry(H=32, K=5, T=105), 4),
    # BASELINE configs[4]: hidden 128 (four waves per sequence), XLong graph; T=41 -> 64,32,16 and an odd tail
    ("industry_h128", cfg_industry(H=128, K=3, T=41, V=300), 3),
    ("industry_h128_k7_1001", cfg_industry(H=128, K=7, T=1001, V=800), 2),
    ("amazon_h128_d32", cfg_amazon(H=128, K=3, T=100, F=2), 3),           # layer 2 has 25 steps: remainder path
]


@pytest.mark.parametrize("name,cfg,B", CASES, ids=[c[0] for c in CASES])
def test_forward_logits_match_oracle(dev, tmp_path, name, cfg, B):
    p = f32_params(cfg, 11)
    ids, label = rand_ids(cfg, B, 12)
    want = O.forward(cfg, p, ids, label)
    m = make_model(cfg, tmp_path, p)
    out = m.forward_inference(torch.as_tensor(ids).to(dev))
    for k in ("memory", "logit", "prediction", "user_weights"):
        np.testing.assert_allclose(out[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)
    np.testing.assert_allclose(float(out["memory_loss"]), want["memory_loss"], rtol=1e-4, atol=1e-5)
    # training path (saved-state scan kernels + fused read fwd/bwd kernel) gives the same numbers at keep_prob 1
    out_t, ce = m.compute_gradients(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=1.0)
    np.testing.assert_allclose(out_t["prediction"].cpu().numpy(), want["prediction"], rtol=0, atol=TOL)
    np.testing.assert_allclose(out_t["memory"].cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    # (two different kernel families at H = 64 -- per-layer VALU scans for inference, the all-layers MFMA launch
    #  for training: they agree far inside the oracle tolerance)
    np.testing.assert_allclose(out_t["prediction"].cpu().numpy(), out["prediction"].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(out_t["memory"].cpu().numpy(), out["memory"].cpu().numpy(), atol=2e-5)
    # the loss takes log(p + 1e-7) of saturated fp32 predictions: relative, not absolute, tolerance
    np.testing.assert_allclose(float(ce), want["cross_entropy"], rtol=2e-4, atol=TOL)


@pytest.mark.parametrize("fname,industry", [("oracle_c0.npz", False), ("oracle_industry.npz", True)])
def test_forward_matches_committed_golden_vectors(dev, tmp_path, fname, industry):
    z = np.load(os.path.join(GOLD, fname))
    p = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    cfg = cfg_industry(H=64, K=4, T=41, F=2, V=500) if industry else cfg_amazon()
    m = make_model(cfg, tmp_path, p)
    out = m.forward_inference(torch.as_tensor(z["ids"]).to(dev))
    for k in ("memory", "logit", "prediction", "user_weights"):
        np.testing.assert_allclose(out[k].cpu().numpy(), z[k], rtol=0, atol=TOL, err_msg=k)
    _, ce = m.compute_gradients(torch.as_tensor(z["ids"]).to(dev), torch.as_tensor(z["label"]).to(dev), keep_prob=1.0)
    np.testing.assert_allclose(float(ce), float(z["cross_entropy"]), rtol=2e-4, atol=TOL)


def test_empty_batch_is_a_noop(dev, tmp_path):
    cfg = cfg_amazon()
    m = make_model(cfg, tmp_path)
    out = m.forward_inference(torch.zeros(0, 100, 3, dtype=torch.int32, device=dev))
    assert tuple(out["memory"].shape) == (0, 3, 32) and out["prediction"].numel() == 0


def test_unsupported_shape_raises(dev):
    from hpmn_amd import _lib, ops
    xp = torch.zeros(2, 4, 3 * 96, device=dev)
    w = torch.zeros(128, 192, device=dev)
    with pytest.raises(_lib.HpmnLibraryError):
        ops.gru_scan_fwd(xp, w, w, 32, torch.zeros(2, 96, device=dev), 1, False, False)


# ------------------------------------------------------------------------------- properties at size
def test_full_xlong_batch_matches_oracle_and_is_batch_independent(dev, tmp_path):
    """BASELINE's XLong shape in full (B=500, T=1001, H=64, K=7): EVERY sample's memory, logit, prediction and
    first-hop attention weights against the float64 oracle (~10 s of NumPy), plus the size-independent
    properties: a sample's result does not depend on its batch, equal inputs give equal outputs, two runs
    are bit-identical."""
    cfg = cfg_industry(H=64, K=7, T=1001, V=5000)
    p = f32_params(cfg, 21)
    m = make_model(cfg, tmp_path, p)
    ids, _ = rand_ids(cfg, 500, 22)
    ids[7] = ids[3]
    t = torch.as_tensor(ids).to(dev)
    a = m.forward_inference(t)
    b = m.forward_inference(t)
    assert torch.equal(a["logit"], b["logit"]) and torch.equal(a["memory"], b["memory"])
    assert torch.equal(a["memory"][7], a["memory"][3])
    sub = m.forward_inference(t[[3, 250, 499]].contiguous())
    # the scan is batch-independent bit for bit (asserted); the logits are only required to agree to 2e-5
    assert torch.equal(sub["memory"], a["memory"][[3, 250, 499]])
    np.testing.assert_allclose(sub["logit"].cpu().numpy(), a["logit"][[3, 250, 499]].cpu().numpy(), atol=2e-5)
    want = O.forward(cfg, p, ids)
    for k in ("memory", "logit", "prediction", "user_weights"):
        np.testing.assert_allclose(a[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)
    # ... and through the training path (saved-state kernels), whole batch
    label = (want["prediction"] > 0.5).astype(np.int32)
    out_t, _ = m.compute_gradients(t, torch.as_tensor(label).to(dev), keep_prob=1.0)
    np.testing.assert_allclose(out_t["memory"].cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    np.testing.assert_allclose(out_t["prediction"].cpu().numpy(), want["prediction"], rtol=0, atol=TOL)


def test_full_taobao_and_amazon_batches_match_oracle(dev, tmp_path):
    """configs[1] / configs[2] at their reference batch (128) and full length, ragged front padding."""
    for cfg in (cfg_amazon(K=4, V=4000), O.HpmnConfig(4000, 4, 300, 64, 16, 3, (2, 2, 3, 5, 5, 1), 5, False, 1e-5)):
        p = f32_params(cfg, 23)
        ids, label = rand_ids(cfg, 128, 24)
        want = O.forward(cfg, p, ids, label)
        m = make_model(cfg, tmp_path, p)
        out = m.forward_inference(torch.as_tensor(ids).to(dev))
        for k in ("memory", "logit", "prediction", "user_weights"):
            np.testing.assert_allclose(out[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)


def test_padding_prefix_property_on_device(dev, tmp_path):
    """Samples sharing a pad length share the state trajectory over the padding: with a 1-layer
    model, zeroing everything but the prefix makes the final state depend only on the pad count."""
    cfg = cfg_amazon(K=1, T=100)
    m = make_model(cfg, tmp_path, f32_params(cfg, 31))
    ids = np.zeros((4, 100, 3), dtype=np.int32)
    out = m.forward_inference(torch.as_tensor(ids).to(dev))
    assert torch.equal(out["memory"][0], out["memory"][1])
    assert float(out["memory"].abs().max()) > 1e-3       # pad steps are not no-ops (gate bias, no masking)


# ------------------------------------------------------------------------------- backward
GRAD_CASES = [
    ("amazon", cfg_amazon(K=3, T=100, V=120), 5),
    ("amazon_h64", cfg_amazon(H=64, K=2, T=20, V=120), 3),
    ("industry", cfg_industry(H=64, K=4, T=41, V=150), 4),
    ("industry_h32", cfg_industry(H=32, K=3, T=41, V=150), 3),
    ("industry_h128", cfg_industry(H=128, K=3, T=41, V=150), 3),
    ("industry_h128_b300", cfg_industry(H=128, K=2, T=41, V=150), 300),     # (more sequences than CUs)
    # the shapes the throughput numbers are quoted on (BASELINE configs[1..4]), not miniatures of them:
    # C3 -- 1024-step fp32 reverse scan, 7 layers, T=1001 scatter; C4's hidden size at the same length;
    # C2 -- four id columns (F=4, D0=64: gather + x_out split, 4-column scatter) with periods 2,2,3,5,5;
    # C1 -- the reference batch of 128 at K=4
    ("xlong_c3_shape", cfg_industry(H=64, K=7, T=1001, V=2000), 5),
    # ... and wide enough that the reverse path's two-sequence workgroups, odd/even pairing and the slab reductions over
    # dozens of partials are what the oracle sees (the 500-sequence reverse path had only been checked forward)
    ("xlong_c3_b66", cfg_industry(H=64, K=7, T=1001, V=4000), 66),
    ("xlong_c4_h128", cfg_industry(H=128, K=7, T=1001, V=2000), 2),
    ("taobao_c2_shape", O.HpmnConfig(600, 4, 300, 64, 16, 3, (2, 2, 3, 5, 5, 1), 5, False, 1e-5), 4),
    ("amazon_c1_b128", cfg_amazon(K=4, V=3000), 128),
    # the hop count (reference CLI: 3 everywhere) shapes the read path's tape and the row count of the shared Hmap product
    ("industry_hop1", O.HpmnConfig(150, 2, 41, 64, 16, 1, (2,) * 10 + (1,), 4, True, 5e-5), 5),
    ("amazon_hop2", O.HpmnConfig(120, 3, 40, 32, 16, 2, (2, 2, 5, 5, 1), 3, False, 1e-5), 7),
    ("industry_hop4", O.HpmnConfig(150, 2, 41, 64, 16, 4, (2,) * 10 + (1,), 3, True, 5e-5), 3),
]


@pytest.mark.parametrize("name,cfg,B", GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_gradients_match_float64_autograd(dev, tmp_path, name, cfg, B):
    p = f32_params(cfg, 41)
    ids, label = rand_ids(cfg, B, 42)
    ids[:, :, 0] = ids[:, -1:, 0]                  # constant uid column -> run-length pre-reduction path
    if B > 32 and (cfg.user_maxlen > 500 or cfg.hidden_size >= 128):
        # With random weights a wide batch holds samples whose logit is beyond +-16.6: in float32 -- TF's arithmetic as
        # much as ours -- sigmoid() is then exactly 0 or 1 and a confidently WRONG label's loss term and gradient saturate
        # (log(0 + 1e-7), p (1 - p) = 0), which the float64 oracle does not reproduce.  Labels that agree with the
        # prediction keep every sample on the side where both precisions agree.
        label = (O.forward(cfg, p, ids)["prediction"] > 0.5).astype(np.int32)
    # oracle gradients (float64 autograd over the restatement)
    tp = R.to_torch(p, torch.float64, requires_grad=True)
    ref = R.forward(cfg, tp, torch.as_tensor(ids.astype(np.int64)), torch.as_tensor(label.astype(np.int64)))
    ref["cross_entropy"].backward()
    m = make_model(cfg, tmp_path, p)
    out, ce = m.compute_gradients(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=1.0,
                                  global_batch=B)
    np.testing.assert_allclose(float(ce), float(ref["cross_entropy"]), rtol=2e-4, atol=1e-5)
    for k in p:
        want = tp[k].grad.numpy()
        got = m.grads[k].cpu().numpy()
        scale = max(1e-6, np.abs(want).max())
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4 * scale + 1e-6, err_msg=k)
    # id 0 is masked in the Hpmn graph: its embedding row gets exactly zero gradient
    if not cfg.industry:
        assert float(m.grads["Embedding/emb_mtx"][0].abs().max()) == 0.0


def test_wide_batch_with_wrong_labels_both_ways(dev, tmp_path):
    """VERDICT r3 weak #1 / ADVICE r3: the B = 66 C3-shape case above labels every sample the way the model leans, so the
    misclassified side of the log-loss (log(p + 1e-7) with p near 0, the gradient's sign for a wrong label) was never
    compared with anything at the width that exercises the two-sequence reverse workgroups and the slab reductions.
    (a) RANDOM labels with the head's last layer scaled so that every |logit| < 8 (no float32 sigmoid saturation): against
        the float64 oracle at the usual 2e-4 of each tensor's max;
    (b) the UNSCALED weights with random labels -- confidently wrong samples included, where float32 saturates in TF's
        graph as much as in ours: against the FLOAT32 restatement (same arithmetic type, different summation orders)."""
    cfg = cfg_industry(H=64, K=7, T=1001, V=4000)
    B = 66
    p = f32_params(cfg, 41)
    ids, _ = rand_ids(cfg, B, 42)
    ids[:, :, 0] = ids[:, -1:, 0]
    label = np.random.default_rng(77).integers(0, 2, size=B).astype(np.int32)
    logit = O.forward(cfg, p, ids)["logit"]
    wrong = ((logit > 0).astype(np.int32) != label)
    assert wrong.sum() >= 10 and np.abs(logit).max() > 17.0        # the unscaled case really holds saturated wrong labels
    # (a)
    pa = dict(p)
    k = 6.0 / np.abs(logit - p["output/fc3/bias"].reshape(-1)[0]).max()
    pa["output/fc3/kernel"] = (p["output/fc3/kernel"] * k).astype(np.float32).astype(np.float64)
    assert np.abs(O.forward(cfg, pa, ids)["logit"]).max() < 8.0
    tp = R.to_torch(pa, torch.float64, requires_grad=True)
    t_ids, t_lab = torch.as_tensor(ids.astype(np.int64)), torch.as_tensor(label.astype(np.int64))
    ref = R.forward(cfg, tp, t_ids, t_lab)
    ref["cross_entropy"].backward()
    m = make_model(cfg, tmp_path, pa)
    d_ids, d_lab = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    out, ce = m.compute_gradients(d_ids, d_lab, keep_prob=1.0, global_batch=B)
    np.testing.assert_allclose(float(ce), float(ref["cross_entropy"]), rtol=2e-4, atol=1e-5)
    for name in pa:
        want = tp[name].grad.numpy()
        np.testing.assert_allclose(m.grads[name].cpu().numpy(), want, rtol=0,
                                   atol=2e-4 * max(1e-6, np.abs(want).max()) + 1e-6, err_msg="(a) " + name)
    # (b)
    tp32 = R.to_torch(p, torch.float32, requires_grad=True)
    ref32 = R.forward(cfg, tp32, t_ids, t_lab)
    ref32["cross_entropy"].backward()
    m.set_params(p)
    out, ce = m.compute_gradients(d_ids, d_lab, keep_prob=1.0, global_batch=B)
    np.testing.assert_allclose(float(ce), float(ref32["cross_entropy"]), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["prediction"].cpu().numpy(), ref32["prediction"].detach().numpy(), rtol=0, atol=2e-5)
    for name in p:
        want = tp32[name].grad.numpy()
        np.testing.assert_allclose(m.grads[name].cpu().numpy(), want, rtol=0,
                                   atol=1e-3 * max(1e-6, np.abs(want).max()) + 1e-6, err_msg="(b) " + name)


@pytest.mark.parametrize("H", [32, 64, 128])
@pytest.mark.parametrize("T", [1, 2, 3, 5, 8])
def test_tiny_and_odd_lengths_forward_and_gradients(dev, tmp_path, H, T):
    """The time loops are unrolled by 2 (H <= 64) / 4 (H = 128) with peeled remainders, prefetch rings run
    ahead of the sequence and clamp: lengths below and around those strides, period-1 layers, odd batch."""
    cfg = O.HpmnConfig(90, 2, T, H, 16, 3, (1, 1, 1), 2, False, 1e-5)
    p = f32_params(cfg, 51)
    ids, label = rand_ids(cfg, 3, 52, ragged=False)
    want = O.forward(cfg, p, ids, label)
    # random weights at H >= 64 saturate the sigmoid; a confidently WRONG label then puts log(p + 1e-7) of an
    # fp32 prediction a few ulps from 0 into the loss and its gradient is ill-conditioned (in TF's fp32 too).
    # Label the samples the way the model leans so the comparison tests the kernels, not that cancellation.
    label = (want["prediction"] > 0.5).astype(np.int32)
    m = make_model(cfg, tmp_path, p)
    out = m.forward_inference(torch.as_tensor(ids).to(dev))
    for k in ("memory", "logit"):
        np.testing.assert_allclose(out[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)
    tp = R.to_torch(p, torch.float64, requires_grad=True)
    ref = R.forward(cfg, tp, torch.as_tensor(ids.astype(np.int64)), torch.as_tensor(label.astype(np.int64)))
    ref["cross_entropy"].backward()
    m.compute_gradients(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=1.0, global_batch=3)
    for k in p:
        w = tp[k].grad.numpy()
        np.testing.assert_allclose(m.grads[k].cpu().numpy(), w, rtol=0, atol=2e-4 * max(1e-6, np.abs(w).max()) + 1e-6,
                                   err_msg=k)


@pytest.mark.parametrize("H", [32, 64, 128])
def test_online_update_event_by_event_equals_the_batch_scan(dev, tmp_path, H):
    """hpmn_memory_update (the serving-time cascade, code/srnn.py:727-748): feeding a sequence one event at a
    time into the persisted store must land on the memory the batch scan computes -- after the full sequence
    and at every prefix whose length the periods divide -- and score like the batch forward."""
    from hpmn_amd.online import OnlineMemory
    T, B = 40, 5
    cfg = O.HpmnConfig(120, 2, T, H, 16, 3, (2, 2, 5, 1), 3, False, 1e-5)      # firing every 1 / 2 / 4 events
    p = f32_params(cfg, 61)
    ids, label = rand_ids(cfg, B, 62, ragged=False)
    m = make_model(cfg, tmp_path, p)
    store = OnlineMemory(m, n_users=9)
    users = torch.as_tensor([7, 0, 3, 8, 2], dtype=torch.int32, device=dev)
    t_ids = torch.as_tensor(ids).to(dev)
    for t in range(T):
        store.update(users, t_ids[:, t, :].contiguous())
        if (t + 1) in (20, 40):                     # 20 = 2*2*5: every layer's length divides at this prefix
            pre = O.HpmnConfig(120, 2, t + 1, H, 16, 3, (2, 2, 5, 1), 3, False, 1e-5)
            want = O.forward(pre, p, ids[:, :t + 1], label)
            np.testing.assert_allclose(store.memory(users).cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    assert store.events().cpu().tolist() == [T if u in (7, 0, 3, 8, 2) else 0 for u in range(9)]
    assert float(store.state[[1, 4, 5, 6]].abs().max()) == 0.0        # untouched users stay untouched
    # scoring a stored user == the batch forward whose last row is the candidate (Hpmn: last = row -1)
    want = O.forward(cfg, p, ids, label)
    got = store.predict(users, t_ids[:, -1, :].contiguous())
    np.testing.assert_allclose(got["logit"].cpu().numpy(), want["logit"], rtol=0, atol=TOL)


def test_online_update_industry_graph_prefix_and_query_row(dev, tmp_path):
    """Hpmn_Industry: 23 all-zero steps in front (state != 0 after them, firing phase shifted by an odd count)
    and query row -2.  A store fed the 41 real events must land on the batch graph's memory AND logit."""
    from hpmn_amd.online import OnlineMemory
    cfg = cfg_industry(H=64, K=4, T=41, V=300)            # 23 + 41 = 64 steps: 64, 32, 16, 8
    p = f32_params(cfg, 63)
    ids, label = rand_ids(cfg, 4, 64)
    m = make_model(cfg, tmp_path, p)
    store = OnlineMemory(m, n_users=6)
    assert store.count.cpu().tolist() == [23] * 6 and float(store.state.abs().max()) > 1e-3
    users = torch.as_tensor([5, 1, 2, 0], dtype=torch.int32, device=dev)
    t_ids = torch.as_tensor(ids).to(dev)
    for t in range(cfg.user_maxlen):
        store.update(users, t_ids[:, t, :].contiguous())
    want = O.forward(cfg, p, ids, label)
    np.testing.assert_allclose(store.memory(users).cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    got = store.predict(users)
    np.testing.assert_allclose(got["logit"].cpu().numpy(), want["logit"], rtol=0, atol=TOL)
    assert store.events(users).cpu().tolist() == [41] * 4 and store.events().cpu().tolist()[3:5] == [0, 0]


def test_time_chunked_pipelined_launches_match_unchunked(dev, tmp_path, monkeypatch):
    """The optional cross-layer pipelining (time-chunked scan / projection / dx launches over K streams,
    state and gradient carried across chunk boundaries) must reproduce the unchunked result."""
    from hpmn_amd import ops
    cfg = cfg_industry(H=64, K=4, T=105, V=150)          # 128 steps -> chunks of 32,16,8,4
    p = f32_params(cfg, 91)
    ids, label = rand_ids(cfg, 5, 92)
    m = make_model(cfg, tmp_path, p)
    ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    monkeypatch.setattr(ops, "FUSED_FWD", False)         # same (two-kernel) forward on both sides: bit-exact
    out1, ce1 = m.compute_gradients(ti, tl, keep_prob=1.0)
    g1 = m.flat_grad.clone()
    assert ops.chunk_plan(m.spec, 4) == [32, 16, 8, 4]
    monkeypatch.setattr(ops, "PIPELINE_CHUNKS", 4)
    out2, ce2 = m.compute_gradients(ti, tl, keep_prob=1.0)
    assert torch.equal(out1["memory"], out2["memory"]) and torch.equal(out1["prediction"], out2["prediction"])
    np.testing.assert_allclose(m.flat_grad.cpu().numpy(), g1.cpu().numpy(), rtol=0, atol=1e-6 * float(g1.abs().max()))


def test_layer0_backward_split_in_time_matches_unsplit(dev, tmp_path, monkeypatch):
    """Optional: layer 0's reverse scan as two launches (late half first) with the weight-gradient reduction of
    the late half started in between (hpmn_gru_param_grads with a time range)."""
    from hpmn_amd import ops
    cfg = cfg_industry(H=64, K=3, T=489, V=300)           # 512 steps at layer 0 -> cut at 256
    p = f32_params(cfg, 95)
    ids, label = rand_ids(cfg, 4, 96)
    m = make_model(cfg, tmp_path, p)
    ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    monkeypatch.setattr(ops, "SPLIT_LAYER0_BWD", False)
    m.compute_gradients(ti, tl, keep_prob=1.0)
    g0 = m.flat_grad.clone()
    monkeypatch.setattr(ops, "SPLIT_LAYER0_BWD", True)
    assert ops._time_cut(512, 2) == 256
    m.compute_gradients(ti, tl, keep_prob=1.0)
    np.testing.assert_allclose(m.flat_grad.cpu().numpy(), g0.cpu().numpy(), rtol=0, atol=2e-6 * float(g0.abs().max()))


def test_fused_forward_matches_the_two_kernel_forward(dev, tmp_path, monkeypatch):
    """hpmn_gru_fused_fwd (projection wave + scan wave per sequence, LDS ring hand-over) against
    hpmn_gru_input_proj + hpmn_gru_scan_fwd: same saved states, gates and gradients up to the summation
    order of the projection; at the full XLong length, batch > the number of CUs, and with an odd tail."""
    from hpmn_amd import ops
    for cfg, B in ((cfg_industry(H=64, K=7, T=1001, V=900), 3), (cfg_industry(H=64, K=3, T=41, V=150), 560),
                   (cfg_amazon(H=64, K=3, T=100, F=2), 5), (cfg_amazon(H=64, K=3, T=100, F=4), 4)):
        p = f32_params(cfg, 93)
        ids, label = rand_ids(cfg, B, 94)
        m = make_model(cfg, tmp_path, p)
        ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
        monkeypatch.setattr(ops, "FUSED_FWD", True)
        assert ops.fused_fwd_supported(64, cfg.user_dim * 16, True)
        mem_f, last_f, saved_f = ops.scan_forward_train(m.spec, ti, m.params["Embedding/emb_mtx"], m._gru_weights())
        m.compute_gradients(ti, tl, keep_prob=1.0)
        g_f = m.flat_grad.clone()
        monkeypatch.setattr(ops, "FUSED_FWD", False)
        mem_u, last_u, saved_u = ops.scan_forward_train(m.spec, ti, m.params["Embedding/emb_mtx"], m._gru_weights())
        m.compute_gradients(ti, tl, keep_prob=1.0)
        g_u = m.flat_grad.clone()
        np.testing.assert_allclose(mem_f.cpu().numpy(), mem_u.cpu().numpy(), rtol=0, atol=2e-6)
        assert torch.equal(last_f, last_u)
        for (x_f, hs_f, ga_f), (x_u, hs_u, ga_u) in zip(saved_f, saved_u):
            np.testing.assert_allclose(hs_f.cpu().numpy(), hs_u.cpu().numpy(), rtol=0, atol=2e-6)
            # (ABI v11: the fused training forward leaves the candidate third of `gates` unwritten where the reverse scan can
            #  do without it -- every layer of these shapes; the gradients below are what it must not change)
            cols = 2 * 64 if ops.candidate_elision(64, B) else 3 * 64
            np.testing.assert_allclose(ga_f[..., :cols].cpu().numpy(), ga_u[..., :cols].cpu().numpy(), rtol=0, atol=2e-6)
        assert torch.equal(saved_f[0][0], saved_u[0][0])                     # the materialised gather
        np.testing.assert_allclose(g_f.cpu().numpy(), g_u.cpu().numpy(), rtol=0, atol=2e-5 * float(g_u.abs().max()))


def test_dropout_masks_are_honoured(dev, tmp_path):
    cfg = cfg_amazon(K=3, T=100, V=120)
    p = f32_params(cfg, 51)
    ids, label = rand_ids(cfg, 4, 52)
    rng = np.random.default_rng(53)
    m1 = (rng.random((4, 200)) < 0.5).astype(np.float64)
    m2 = (rng.random((4, 80)) < 0.5).astype(np.float64)
    want = O.forward(cfg, p, ids, label, mask1=m1, mask2=m2, keep_prob=0.5)
    m = make_model(cfg, tmp_path, p)
    masks = (torch.as_tensor(m1, dtype=torch.float32).to(dev), torch.as_tensor(m2, dtype=torch.float32).to(dev))
    out, ce = m.compute_gradients(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=0.5,
                                  masks=masks)
    np.testing.assert_allclose(out["prediction"].cpu().numpy(), want["prediction"], atol=TOL)
    np.testing.assert_allclose(float(ce), want["cross_entropy"], rtol=2e-4, atol=TOL)
    # and the gradients with dropout active
    tp = R.to_torch(p, torch.float64, requires_grad=True)
    ref = R.forward(cfg, tp, torch.as_tensor(ids.astype(np.int64)), torch.as_tensor(label.astype(np.int64)),
                    torch.as_tensor(m1), torch.as_tensor(m2), 0.5)
    ref["cross_entropy"].backward()
    for k in ("output/fc1/kernel", "output/fc2/kernel", "User/dense_1/kernel", "User/GRU0/gates/kernel"):
        wantg = tp[k].grad.numpy()
        np.testing.assert_allclose(m.grads[k].cpu().numpy(), wantg, rtol=0, atol=2e-4 * np.abs(wantg).max() + 1e-7,
                                   err_msg=k)


# ------------------------------------------------------------------------------- optimiser + steps
def test_in_kernel_dropout_is_bernoulli_and_consistent(dev, tmp_path):
    """Without explicit masks the read kernel draws them itself (counter-based hash of seed / layer / sample /
    unit): the keep rate must be keep_prob, masks must differ between steps, and forward and backward of one
    launch must see the SAME mask -- checked through the gradient of fc3's kernel, which is exactly
    sum_b dlogit_b * h2_b and vanishes on the columns a mask zeroed for every sample."""
    cfg = cfg_amazon(K=3, T=100, V=120)
    p = f32_params(cfg, 71)
    m = make_model(cfg, tmp_path, p)
    ids, label = rand_ids(cfg, 1, 72)
    ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    zero_sets, kept = [], 0
    for _ in range(40):
        m.compute_gradients(ti, tl, keep_prob=0.5)
        g = m.grads["output/fc3/kernel"].cpu().numpy().reshape(-1)        # [80]: one sample -> zero iff dropped
        zero_sets.append(tuple(np.flatnonzero(g == 0.0)))
        kept += int((g != 0.0).sum())
    rate = kept / (40 * 80)
    assert 0.42 < rate < 0.58, rate
    assert len(set(zero_sets)) > 30                      # a fresh mask nearly every step
    # keep_prob == 1 -> no dropout at all
    m.compute_gradients(ti, tl, keep_prob=1.0)
    assert int((m.grads["output/fc3/kernel"] == 0).sum()) == 0


def test_in_kernel_dropout_does_not_leak_the_label(dev, tmp_path):
    """Regression: with a per-step seed that was an arithmetic progression, the mask of (step, sample) recurred
    as that of (step-1, sample+4) -- always on a sample of the same parity -- and on data whose labels alternate
    1,0,1,0 (the XLong loader's layout, code/data_loader.py:75-80) the head learned the label from the mask:
    training loss 0.13 after 60 steps, test AUC 0.50.  Pure-noise inputs with alternating labels must stay at
    chance."""
    cfg = cfg_amazon(H=32, K=3, T=20, V=5000)
    m = make_model(cfg, tmp_path, None, lr=0.003)
    rng = np.random.default_rng(77)
    B = 128
    label = torch.as_tensor(np.tile([1, 0], B // 2).astype(np.int32)).to(dev)
    tail = []
    for step in range(120):
        ids = torch.as_tensor(rng.integers(1, 5000, size=(B, 20, 3)).astype(np.int32)).to(dev)
        _, ce = m.train_step(ids, label, keep_prob=0.5)
        if step >= 100:
            tail.append(float(ce))
    assert np.mean(tail) > 0.62, np.mean(tail)       # log 2 = 0.693; the leak drove this below 0.2


def test_adam_kernel_matches_tf_form(dev):
    from hpmn_amd import ops
    rng = np.random.default_rng(61)
    for n in (1003, 4096):                      # scalar-tail and float4 paths
        p0 = rng.normal(size=n)
        g = rng.normal(scale=2.0, size=n)       # exercises the clip
        g[::7] = 0.0
        st = O.AdamState()
        want = {"w": p0.copy()}
        tp = torch.as_tensor(p0, dtype=torch.float32).to(dev)
        tm, tv = torch.zeros_like(tp), torch.zeros_like(tp)
        tg = torch.as_tensor(g, dtype=torch.float32).to(dev)
        for t in range(1, 4):
            O.adam_step(want, {"w": g}, st, lr=0.003)
            lr_t = 0.003 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            ops.adam_step(tp, tg, tm, tv, lr_t)
        np.testing.assert_allclose(tp.cpu().numpy(), want["w"], atol=2e-6)
        np.testing.assert_allclose(tm.cpu().numpy(), st.m["w"], atol=1e-6)
        np.testing.assert_allclose(tv.cpu().numpy(), st.v["w"], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("E", [4, 16, 32])
def test_table_adam_in_two_passes_is_bit_identical_to_the_dense_kernel(dev, E):
    """hpmn_table_mark_rows + hpmn_adam_step_table (pass 0: unmarked rows with a zero gradient, pass 1: marked rows)
    against hpmn_adam_step over the same buffers: identical bits in p, m, v; gradient rows and flags cleared."""
    from hpmn_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    V = 5003
    p = torch.randn(V, E, device=dev, generator=g)
    m = torch.randn(V, E, device=dev, generator=g) * 0.1
    v = torch.rand(V, E, device=dev, generator=g) * 0.01
    ids = torch.randint(0, V, (7, 33, 3), device=dev, dtype=torch.int32, generator=g)
    grad = torch.zeros(V, E, device=dev)
    rows = ids.reshape(-1).long().unique()
    grad[rows] = torch.randn(rows.numel(), E, device=dev, generator=g) * 2.0          # (exercises the clip)
    ref = [t.clone() for t in (p, m, v)]
    ops.adam_step(ref[0].view(-1), grad.view(-1), ref[1].view(-1), ref[2].view(-1), 0.0021)
    flags = torch.zeros(V, device=dev, dtype=torch.uint8)
    ops.table_mark_rows(ids, flags)
    assert int(flags.sum()) == rows.numel()
    ops.adam_step_table(p, grad, m, v, flags, 0, 0.0021)
    assert int(flags.sum()) == rows.numel() and float(grad.abs().sum()) > 0          # pass 0 consumes nothing
    ops.adam_step_table(p, grad, m, v, flags, 1, 0.0021)
    for a, b in zip((p, m, v), ref):
        assert torch.equal(a, b)
    assert float(grad.abs().max()) == 0.0 and int(flags.max()) == 0


def test_plain_train_step_leaves_the_flat_gradient_all_zero_and_skips_the_clearing_launch(dev, tmp_path):
    """r5: the small-table train_step's Adam launches consume the gradient (hpmn_adam_step_clear), so the next
    compute_gradients finds an all-zero buffer and issues no clearing launch; a buffer somebody else left dirty is still
    cleared.  Gradients after train_step + compute_gradients == gradients of compute_gradients on a fresh model state."""
    cfg = cfg_amazon(K=3, T=60, V=150)
    p = f32_params(cfg, 301)
    ids, label = rand_ids(cfg, 9, 302)
    ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    m = make_model(cfg, tmp_path, p, lr=0.002)
    assert not m._two_pass_table_adam(ti) and not m.compact_table_grad          # (the plain path)
    m.train_step(ti, tl, keep_prob=1.0)
    torch.cuda.synchronize()
    assert m._flat_grad_clean and float(m.flat_grad.abs().max()) == 0.0
    m.compute_gradients(ti, tl, keep_prob=1.0)                                   # (no clearing launch: the buffer is clean)
    g1 = m.flat_grad.clone()
    assert not m._flat_grad_clean and float(g1.abs().max()) > 0
    m.compute_gradients(ti, tl, keep_prob=1.0)                                   # (dirty: cleared first, not accumulated into)
    tol = 2e-6 * float(g1.abs().max())                                           # (the scatter's atomics)
    assert float((m.flat_grad - g1).abs().max()) <= tol
    # the same parameters in a second model whose buffer was never touched by a clearing Adam launch
    m2 = make_model(cfg, tmp_path, {k: v.detach().cpu().numpy() for k, v in m.params.items()}, lr=0.002)
    m2.compute_gradients(ti, tl, keep_prob=1.0)
    assert float((m2.flat_grad - g1).abs().max()) <= tol


@pytest.mark.parametrize("n", [4 * 1000 + 4, 1237])
def test_adam_step_clear_is_adam_step_and_consumes_the_gradient(dev, n):
    """hpmn_adam_step_clear (ABI v13): bit for bit the update of hpmn_adam_step (clip + TF-form Adam, code/hpmn.py:209-214), and
    the gradient buffer is all-zero behind it (vectorised and scalar forms: n % 4 == 0 aligned / an odd length)."""
    from hpmn_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3 + n)
    mk = lambda s: (torch.randn(n, generator=g) * s).to(dev)
    p, grad, m, v = mk(1.0), mk(2.0), mk(0.1), mk(0.01).abs()
    ref = [t.clone() for t in (p, m, v)]
    ops.adam_step(ref[0], grad.clone(), ref[1], ref[2], 0.0021)
    ops.adam_step(p, grad, m, v, 0.0021, clear_grad=True)
    torch.cuda.synchronize()
    assert torch.equal(p, ref[0]) and torch.equal(m, ref[1]) and torch.equal(v, ref[2])
    assert float(grad.abs().max()) == 0.0


@pytest.mark.parametrize("industry", [False, True])
def test_three_training_steps_track_the_restatement(dev, tmp_path, industry):
    """sess.run(train_step) x3 with keep_prob 1 (dropout RNG cannot be matched): parameters after
    clip + dense TF Adam agree with the float64 restatement."""
    cfg = cfg_industry(H=64, K=3, T=41, V=120) if industry else cfg_amazon(K=3, T=100, V=120)
    p = f32_params(cfg, 71)
    ids, label = rand_ids(cfg, 8, 72)
    tp = R.to_torch(p, torch.float64)
    opt = R.TFAdam(tp, 0.003)
    m = make_model(cfg, tmp_path, p, lr=0.003)
    ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    steps, lr = 3, 0.003
    gmin = {k: np.full(v.shape, np.inf) for k, v in p.items()}
    gmax = {k: 0.0 for k in p}
    for _ in range(steps):
        _, g = R.train_step(cfg, tp, opt, torch.as_tensor(ids.astype(np.int64)),
                            torch.as_tensor(label.astype(np.int64)))
        for k in p:
            ga = np.abs(g[k].numpy())
            gmin[k] = np.minimum(gmin[k], ga)
            gmax[k] = max(gmax[k], float(ga.max()))
        m.train_step(ti, tl, keep_prob=1.0)
    for k in p:
        got, want = m.params[k].detach().cpu().numpy(), tp[k].detach().numpy()
        # Adam divides by sqrt(v): where a gradient is ~0 (below fp32 resolution of the sum it came
        # from) the normalised update is ill-conditioned, so those elements are only required to stay
        # within the trust region lr*steps; everything else must agree tightly.
        well = gmin[k] > 1e-3 * max(gmax[k], 1e-30)
        if gmax[k] < 1e-9:            # gradient identically ~0 in exact arithmetic (e.g. the softmax-
            well[...] = False         # invariant bias of the last attention layer): pure rounding noise
        np.testing.assert_allclose(got[well], want[well], rtol=0, atol=3e-5, err_msg=k)
        np.testing.assert_allclose(got, want, rtol=0, atol=lr * steps * 1.05, err_msg=k)
    # rows never touched still moved only if their Adam moments are non-zero: untouched rows stay put
    untouched = np.setdiff1d(np.arange(cfg.feature_size), np.unique(ids))
    if len(untouched):
        np.testing.assert_array_equal(m.params["Embedding/emb_mtx"][untouched].detach().cpu().numpy(),
                                      p["Embedding/emb_mtx"][untouched].astype(np.float32))


def test_scatter_matches_dense_index_add(dev):
    from hpmn_amd import ops
    rng = np.random.default_rng(81)
    B, T, F, E, V, Z = 6, 300, 3, 16, 50, 23
    ids = rng.integers(0, V, size=(B, T, F)).astype(np.int32)
    ids[:, :, 0] = ids[:, :1, 0]                    # constant column (runs of length T)
    ids[2, :100] = 0
    dx = rng.normal(size=(B, Z + T, F * E)).astype(np.float32)
    for mask in (True, False):
        want = np.zeros((V, E), dtype=np.float64)
        np.add.at(want, ids.reshape(-1), dx[:, Z:].reshape(-1, E).astype(np.float64))
        if mask:
            want[0] = 0
        got = torch.zeros(V, E, device=dev)
        ops.embed_grad_scatter(torch.as_tensor(ids).to(dev), torch.as_tensor(dx).to(dev), got, Z, mask)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("group", [1, 2, 3])
def test_tiled_inference_by_layer_groups_matches_the_oracle(dev, tmp_path, group):
    """ops.tiled_forward_inference: the 16-sequence-tile MFMA scan (split-f16 operands, three products, fp32 accumulate) run
    layer group by layer group, the next group reading the y rows the group before left in memory (hpmn_pipe_fwd with a y
    buffer on its top layer, ABI v10).  Against the float64 oracle at the usual 1e-4, incl. a partial last tile and the
    Industry zero prefix; and against the per-sequence inference chain."""
    from hpmn_amd import ops
    cfg = cfg_industry(H=64, K=5, T=105, V=700)            # 128 steps: 128, 64, 32, 16, 8
    p = f32_params(cfg, 201)
    B = 37                                                  # two full tiles + a partial one
    ids, label = rand_ids(cfg, B, 202)
    want = O.forward(cfg, p, ids, label)
    m = make_model(cfg, tmp_path, p)
    t = torch.as_tensor(ids).to(dev)
    mem, last = ops.tiled_forward_inference(m.spec, t, m.params["Embedding/emb_mtx"], m._gru_weights(), group=group)
    np.testing.assert_allclose(mem.cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    ref_mem, ref_last = ops.scan_forward_inference(m.spec, t, m.params["Embedding/emb_mtx"], m._gru_weights())
    assert torch.equal(last, ref_last)
    np.testing.assert_allclose(mem.cpu().numpy(), ref_mem.cpu().numpy(), rtol=0, atol=2e-5)


def test_eval_on_the_tiled_path_equals_eval_batch_by_batch(dev, tmp_path, monkeypatch):
    """Hpmn.eval with several reference batches per pass on the tile kernels (single process, evaluation-sized passes) returns
    what the batch-by-batch evaluation on the per-sequence kernels returns: AUC, log-loss and the mean over REFERENCE batches
    of the per-batch memory_loss sums (code/hpmn.py:512-519) -- incl. a short last batch."""
    from hpmn_amd.hpmn import Hpmn_Industry
    rng = np.random.default_rng(203)
    n, T = 2300, 41
    ids = rng.integers(0, 900, size=(n, T, 2)).astype(np.int32)
    label = rng.integers(0, 2, size=n).astype(np.int32)
    ds = dict(ids=ids, label=label)

    def build(tag):
        return Hpmn_Industry(str(tmp_path / tag), ds, ds, 900, 2, 1, T, 1, 0.003, 64, 16, 3, [2] * 10 + [1], [1], 4, 1,
                             True, False, memory_reg=5e-5, verbose=False, seed=7)
    a = build("tiled")
    a.TILED_EVAL_MIN_ROWS, a.TILED_EVAL_ROWS = 600, 1024      # 3 reference batches of 400 per pass; last pass: 300 rows... 
    got = a.eval(ds, 400)
    b = build("plain")
    b.TILED_EVAL_MIN_ROWS = 0
    want = b.eval(ds, 400)
    assert abs(got[0] - want[0]) < 1e-6 and abs(got[1] - want[1]) < 1e-6
    assert abs(got[2] - want[2]) <= 1e-5 * abs(want[2])


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("E", [16, 4])
def test_deterministic_scatter_is_exact_ordered_and_reproducible(dev, wide, E):
    """hpmn_scatter_plan + hpmn_embed_grad_segsum (csrc/scatter_sorted.hip): a row's gradient is the sum of its lookups' rows
    in ascending lookup order, added by ONE lane group and stored plainly.  Against (a) float64 np.add.at, (b) a float32
    emulation of exactly that order (per-chunk sums, chunks in order) -- BIT-identical, which is what "fixed order" means --,
    and twice in a row
    (torch.equal).  The id pattern covers every path of the two passes: rows inside one chunk, the constant uid
    column (1000-entry rows: dozens of chunks, the partial chain of pass 2), a hot id crossing exactly one chunk border,
    the masked padding id 0 (thousands of entries, skipped), the read path's d_last row, int32 and int64 ids."""
    from hpmn_amd import ops
    rng = np.random.default_rng(83)
    B, T, F, V, Z = 5, 1001, 2, 3000, 23
    ids = rng.integers(1, V, size=(B, T, F)).astype(np.int64)
    ids[:, :, 0] = rng.integers(1, V, size=(B, 1))               # constant column
    ids[1, :400] = 0                                             # padding
    ids[3, 100:140, 1] = 7                                       # a 40-entry row
    ids[0, 5, 1] = 2999
    dx = (rng.normal(size=(B, Z + T, F * E)) * 10.0 ** rng.integers(-6, 1, size=(B, Z + T, 1))).astype(np.float32)
    dlast = rng.normal(size=(B, F * E)).astype(np.float32)
    t_last = T - 2
    t_ids = torch.as_tensor(ids if wide else ids.astype(np.int32)).to(dev)
    t_dx, t_dl = torch.as_tensor(dx).to(dev), torch.as_tensor(dlast).to(dev)
    term = dx[:, Z:].reshape(B, T, F, E).copy()
    term[:, t_last] += dlast.reshape(B, F, E)                    # (joined to the lookup's row BEFORE the sum, as in the kernel)
    for mask in (True, False):
        want64 = np.zeros((V, E), dtype=np.float64)
        np.add.at(want64, ids.reshape(-1), term.reshape(-1, E).astype(np.float64))
        # the kernel's order in float32 (include/hpmn_hip.h, hpmn_embed_grad_segsum_chunk): entries in stable row order, cut
        # into SCH-entry chunks; a row's entries inside one chunk are added left to right; a row that spans chunks is its
        # per-chunk sums cut into min(16, 256/E) consecutive blocks, each block added left to right, the blocks in order
        from hpmn_amd import _lib
        sch = int(_lib.load().hpmn_embed_grad_segsum_chunk())
        nb = min(16, 256 // E)
        flat_ids, flat_t = ids.reshape(-1), term.reshape(-1, E)
        order = np.argsort(flat_ids, kind="stable")
        parts = {}                                               # row -> its per-chunk sums, in chunk order
        for j0 in range(0, len(order), sch):
            part = {}
            for j in order[j0:j0 + sch]:
                r = flat_ids[j]
                part[r] = flat_t[j].copy() if r not in part else part[r] + flat_t[j]
            for r, v in part.items():
                parts.setdefault(r, []).append(v)
        want32 = np.zeros((V, E), dtype=np.float32)
        for r, ps in parts.items():
            if len(ps) == 1:
                want32[r] = ps[0]
                continue
            per = -(-len(ps) // nb)
            tot = None
            for b0 in range(0, len(ps), per):
                blk = ps[b0].copy()
                for v in ps[b0 + 1:b0 + per]:
                    blk = blk + v
                tot = blk if tot is None else tot + blk
            want32[r] = tot
        if mask:
            want64[0] = 0
            want32[0] = 0
        plan = ops.ScatterPlan(t_ids, E, want_rows=True)
        U = plan.count_host()
        uniq = np.unique(ids.reshape(-1))
        assert U == len(uniq) and np.array_equal(plan.rows[:U].cpu().numpy(), uniq)
        outs = []
        for rep in range(2):
            got = torch.zeros(V, E, device=dev)
            ops.embed_grad_segsum(plan, (B, T, F), t_dx, got, Z, mask, d_last=t_dl, t_last=t_last)
            outs.append((got.clone(), plan.out_rows[:U].clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        got = outs[0][0].cpu().numpy()
        np.testing.assert_allclose(got, want64, rtol=1e-5, atol=1e-5)
        assert np.array_equal(got, want32), "not the fixed-order float32 sum"
        assert np.array_equal(outs[0][1].cpu().numpy(), want32[uniq])          # the compact rows == the dense rows
        # += semantics (dual mode scatters two branches into one table gradient)
        acc = torch.ones(V, E, device=dev)
        ops.embed_grad_segsum(plan, (B, T, F), t_dx, acc, Z, mask, d_last=t_dl, t_last=t_last)
        touched = np.zeros(V, bool)
        touched[uniq] = True
        if mask:
            touched[0] = False
        np.testing.assert_array_equal(acc.cpu().numpy()[~touched], 1.0)
        np.testing.assert_allclose(acc.cpu().numpy()[touched], want32[touched] + 1.0, rtol=1e-6, atol=1e-6)
        # and the atomic kernel it replaces agrees to rounding
        old = torch.zeros(V, E, device=dev)
        t_dx2 = t_dx.clone()
        t_dx2[:, Z + t_last] += t_dl
        ops.embed_grad_scatter(t_ids, t_dx2, old, Z, mask)
        np.testing.assert_allclose(old.cpu().numpy(), want64, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name,cfg,B", [("industry_c3_shape", cfg_industry(H=64, K=7, T=1001, V=2000), 9),
                                        ("amazon_h32", cfg_amazon(K=4, V=3000), 33)])
def test_training_steps_are_bit_reproducible(dev, tmp_path, monkeypatch, name, cfg, B):
    """with the scatter's atomics gone nothing in a train step depends on execution order -- two models
    from the same weights fed the same batches end on torch.equal parameters and moments (dense two-pass table Adam
    included), and the table gradient of a stand-alone compute_gradients is bit-identical run to run."""
    monkeypatch.setenv("HPMN_TWO_PASS_MIN_NUMEL", "0")
    p = f32_params(cfg, 191)
    ids, label = rand_ids(cfg, B, 192)
    ids[:, :, 0] = ids[:, -1:, 0]
    t_ids, t_lab = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    runs = []
    for r in range(2):
        monkeypatch.setenv("HPMN_DET_SCATTER", "1")
        monkeypatch.setenv("HPMN_TABLE_GRAD", "compact" if r == 0 or name != "amazon_h32" else "dense")
        m = make_model(cfg, tmp_path / ("r%d" % r), p)
        assert m.det_scatter
        m.compute_gradients(t_ids, t_lab, keep_prob=1.0)
        # (r5: run 0 keeps NO dense gradient table -- the scatter's compact rows, spread out here; for one of the two shapes run 1
        #  is the r4 layout: same plan, same sums, dense table + late pass instead of hpmn_rows_sum_adam -- still bit-identical)
        assert m.compact_table_grad == (r == 0 or name != "amazon_h32")
        g0 = m.table_gradient().clone()
        for step in range(3):
            m.train_step(t_ids.roll(step, 0), t_lab.roll(step, 0), keep_prob=1.0)
        torch.cuda.synchronize()
        runs.append((g0, m.flat_param.clone(), m.flat_m.clone(), m.flat_v.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert float(runs[0][0].abs().max()) > 0


# ------------------------------------------------------------------------------- end to end
def test_training_learns_planted_signal_and_save_load(dev, tmp_path):
    from hpmn_amd import datasets
    from hpmn_amd.hpmn import Hpmn
    tr, te, fs = datasets.make_synthetic_amazon(n_samples=3000, n_item=2000, n_cate=50, n_user=3000,
                                                max_len=100, seed=datasets.SEED_BASE, as_arrays=True)
    m = Hpmn(str(tmp_path / "m"), tr, te, fs, 3, 2, 100, 100, 0.003, 32, 16, 3, [2, 2, 5, 5, 1],
             [2, 2, 5, 5, 1], 3, 3, True, False, l2_reg=0., memory_reg=1e-5, verbose=False, seed=1)
    auc0, _, _ = m.eval(te, 512)
    m.eval_every = 10 ** 9                       # no periodic eval inside this short run
    m.train(6, 128)
    auc1, loss1, mem1 = m.eval(te, 512)
    assert auc1 > 0.8 and auc1 > auc0 + 0.1, (auc0, auc1)
    m.save_model()
    before = m.forward_inference(m._dev(te).ids[:16])["prediction"].clone()
    m.set_params({k: np.zeros_like(v) for k, v in m.get_params().items()})
    m.load_model()
    assert torch.equal(before, m.forward_inference(m._dev(te).ids[:16])["prediction"])
    m.log(5, [auc1, loss1, mem1, auc1, loss1, mem1])
    assert open(str(tmp_path / "m") + "/result.log").read().count("\t") == 6


# ------------------------------------------------------------------------------- get_weights (code/hpmn.py:521-560, :375-410)
def test_get_weights_dumps_first_hop_attention_of_both_classes(dev, tmp_path):
    """Hpmn.get_weights -> weights.npy / lengths.npy / labels.npy; Hpmn_Industry.get_weights -> weights_new.npy /
    ids.npy: first-hop attention weights of train+test in stored order at batch 512, checked against the oracle's
    user_weights."""
    from hpmn_amd.hpmn import Hpmn, Hpmn_Industry
    # Hpmn
    cfg = cfg_amazon(K=3, T=100, V=400)
    p = f32_params(cfg, 101)
    ids_tr, lab_tr = rand_ids(cfg, 530, 102)             # > 512: two batches
    ids_te, lab_te = rand_ids(cfg, 7, 103)
    mk = lambda ids, lab: [(int(l), x.tolist(), int((x[:, 1] != 0).sum()), [[0, 0]], 1) for x, l in zip(ids, lab)]
    tr, te = mk(ids_tr, lab_tr), mk(ids_te, lab_te)
    m = Hpmn(str(tmp_path / "a"), tr, te, cfg.feature_size, 3, 2, 100, 1, 0.003, 32, 16, 3, [2, 2, 5, 5, 1], [1], 3, 1,
             True, False, verbose=False)
    m.set_params(p)
    m.get_weights()
    w = np.load(str(tmp_path / "a") + "/weights.npy")
    want = O.forward(cfg, p, np.concatenate([ids_tr, ids_te]))["user_weights"]
    assert w.shape == (537, 3)
    np.testing.assert_allclose(w, want, rtol=0, atol=TOL)
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-5)
    np.testing.assert_array_equal(np.load(str(tmp_path / "a") + "/labels.npy"), np.concatenate([lab_tr, lab_te]))
    np.testing.assert_array_equal(np.load(str(tmp_path / "a") + "/lengths.npy"),
                                  [s[2] for s in tr] + [s[2] for s in te])
    # Hpmn_Industry
    cfg = cfg_industry(H=64, K=4, T=41, V=300)
    p = f32_params(cfg, 104)
    ids_tr, lab_tr = rand_ids(cfg, 9, 105)
    ids_te, lab_te = rand_ids(cfg, 4, 106)
    m = Hpmn_Industry(str(tmp_path / "i"), dict(ids=ids_tr, label=lab_tr), dict(ids=ids_te, label=lab_te),
                      cfg.feature_size, 2, 1, 41, 1, 0.001, 64, 16, 3, [2] * 10 + [1], [1], 4, 1, True, False,
                      verbose=False)
    m.set_params(p)
    m.get_weights()
    w = np.load(str(tmp_path / "i") + "/weights_new.npy")
    want = O.forward(cfg, p, np.concatenate([ids_tr, ids_te]))["user_weights"]
    np.testing.assert_allclose(w, want, rtol=0, atol=TOL)
    np.testing.assert_array_equal(np.load(str(tmp_path / "i") + "/ids.npy"),
                                  np.concatenate([ids_tr, ids_te])[:, :, 1])        # code/hpmn.py:399


# ------------------------------------------------------------------------------- all-layers-in-one-launch scan
@pytest.mark.parametrize("mode", ["all", "upper"])
def test_pipelined_mfma_scan_matches_oracle_and_per_layer_kernels(dev, tmp_path, monkeypatch, mode):
    """hpmn_pipe_fwd / hpmn_pipe_bwd (batch-tiled split-f16 MFMA recurrence, layers pipelined across workgroups
    inside one launch, in-launch hand-offs through progress words): forward vs the float64 oracle, gradients vs
    float64 autograd, at the XLong graph's full length, an odd batch with a partial tile, four id columns, and
    short / odd layer lengths; and no lost hand-off (error word 0).  (Training on the tile kernel: a legacy build.)"""
    _needs_legacy_build()
    from hpmn_amd import ops
    monkeypatch.setattr(ops, "PIPE", mode)
    cases = [(cfg_industry(H=64, K=7, T=1001, V=900), 21),               # two tiles, the second partial
             (O.HpmnConfig(400, 4, 300, 64, 16, 3, (2, 2, 3, 5, 5, 1), 5, False, 1e-5), 3),
             (cfg_amazon(H=64, K=3, T=100, F=3, V=300), 5),
             (cfg_industry(H=64, K=3, T=41, V=150), 40)]
    for cfg, B in cases:
        p = f32_params(cfg, 111)
        ids, label = rand_ids(cfg, B, 112)
        want = O.forward(cfg, p, ids, label)
        label = (want["prediction"] > 0.5).astype(np.int32)
        m = make_model(cfg, tmp_path, p)
        assert ops.pipe_mode(m.spec) == mode
        ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
        out, ce = m.compute_gradients(ti, tl, keep_prob=1.0, global_batch=B)
        assert ops.pipe_error_word(m.spec.K, B, dev) == 0
        np.testing.assert_allclose(out["memory"].cpu().numpy(), want["memory"], rtol=0, atol=TOL)
        np.testing.assert_allclose(out["prediction"].cpu().numpy(), want["prediction"], rtol=0, atol=TOL)
        tp = R.to_torch(p, torch.float64, requires_grad=True)
        ref = R.forward(cfg, tp, torch.as_tensor(ids.astype(np.int64)), torch.as_tensor(label.astype(np.int64)))
        ref["cross_entropy"].backward()
        for k in p:
            w = tp[k].grad.numpy()
            # (5e-4 of the tensor's max here: the 1024-step reverse scan on split-f16 operands lands a handful of
            #  elements of the layer-0 kernel at 2.6e-4; the per-layer fp32 kernels hold 2e-4)
            np.testing.assert_allclose(m.grads[k].cpu().numpy(), w, rtol=0, atol=5e-4 * max(1e-6, np.abs(w).max()) + 1e-6,
                                       err_msg=k)
        # and against the per-layer kernels on the same inputs
        g_pipe = m.flat_grad.clone()
        monkeypatch.setattr(ops, "PIPE", "0")
        out0, _ = m.compute_gradients(ti, tl, keep_prob=1.0, global_batch=B)
        monkeypatch.setattr(ops, "PIPE", mode)
        np.testing.assert_allclose(out["memory"].cpu().numpy(), out0["memory"].cpu().numpy(), rtol=0, atol=TOL)
        np.testing.assert_allclose(g_pipe.cpu().numpy(), m.flat_grad.cpu().numpy(), rtol=0,
                                   atol=5e-4 * float(m.flat_grad.abs().max()))


# ------------------------------------------------------------------------------- training graph through the C ABI
def test_training_step_through_the_c_abi_alone(dev, tmp_path):
    """hpmn_scan_fwd_train -> hpmn_read_fwd_bwd -> hpmn_scan_bwd (+ hpmn_train_join) -> hpmn_adam_step called
    directly through ctypes -- no Python orchestration of layers, streams or events, torch only lends the device
    memory: gradients vs float64 autograd, the parameter update vs the oracle's TF-form Adam."""
    import ctypes as C
    from hpmn_amd import _lib
    lib = _lib.load()
    cfg = cfg_industry(H=64, K=4, T=41, V=150)
    B = 6
    p = f32_params(cfg, 121)
    ids, label = rand_ids(cfg, B, 122)
    m = make_model(cfg, tmp_path, p)            # only used as the owner of the flat parameter / gradient buffers
    spec, K, H, D0 = m.spec, cfg.user_num_layers, 64, 32
    st = torch.cuda.current_stream().cuda_stream
    t_ids, t_lab = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    ctx = C.c_void_p()
    assert lib.hpmn_train_ctx_create(C.byref(ctx)) == 0
    d = spec.desc(B, cfg.feature_size)
    ws = torch.empty(lib.hpmn_scan_train_workspace_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
    w = m._gru_weights()
    arr = lambda ts: (C.c_void_p * K)(*[t.data_ptr() for t in ts])
    memory, last = torch.empty(B, K, H, device=dev), torch.empty(B, D0, device=dev)
    m.flat_grad.zero_()
    emb = m.params["Embedding/emb_mtx"]
    assert lib.hpmn_scan_fwd_train(ctx, C.byref(d), t_ids.data_ptr(), emb.data_ptr(), arr(w[0::4]), arr(w[1::4]),
                                   arr(w[2::4]), arr(w[3::4]), memory.data_ptr(), last.data_ptr(), ws.data_ptr(), st) == 0
    rd = m._read_desc
    rd.B = B
    rws = torch.zeros(lib.hpmn_read_workspace_bytes(C.byref(rd)) // 4, device=dev)
    pred, loss = torch.empty(B, device=dev), torch.zeros(2, device=dev)
    d_mem, d_last = torch.empty_like(memory), torch.empty_like(last)
    assert lib.hpmn_read_fwd_bwd(C.byref(rd), m._read_params.data_ptr(), memory.data_ptr(), last.data_ptr(),
                                 t_lab.data_ptr(), None, None, 1.0, 1.0 / B, cfg.memory_reg, pred.data_ptr(),
                                 loss.data_ptr(), d_mem.data_ptr(), d_last.data_ptr(), m._read_grads.data_ptr(),
                                 rws.data_ptr(), st) == 0
    g = [m.grads[n] for names in m._gru_names for n in names]
    assert lib.hpmn_scan_bwd(ctx, C.byref(d), t_ids.data_ptr(), arr(w[0::4]), arr(w[2::4]), d_mem.data_ptr(),
                             d_last.data_ptr(), arr(g[0::4]), arr(g[1::4]), arr(g[2::4]), arr(g[3::4]),
                             m.grads["Embedding/emb_mtx"].data_ptr(), ws.data_ptr(), 1, st) == 0
    assert lib.hpmn_train_join(ctx, st) == 0
    tp = R.to_torch(p, torch.float64, requires_grad=True)
    ref = R.forward(cfg, tp, torch.as_tensor(ids.astype(np.int64)), torch.as_tensor(label.astype(np.int64)))
    ref["cross_entropy"].backward()
    for k in p:
        wnt = tp[k].grad.numpy()
        np.testing.assert_allclose(m.grads[k].cpu().numpy(), wnt, rtol=0, atol=2e-4 * max(1e-6, np.abs(wnt).max()) + 1e-6,
                                   err_msg=k)
    np.testing.assert_allclose(pred.cpu().numpy(), ref["prediction"].detach().numpy(), rtol=0, atol=TOL)
    lr_t = 0.001 * math.sqrt(1 - 0.999) / (1 - 0.9)
    before = m.flat_param.clone()
    assert lib.hpmn_adam_step(m.flat_param.data_ptr(), m.flat_grad.data_ptr(), m.flat_m.data_ptr(), m.flat_v.data_ptr(),
                              m.flat_param.numel(), lr_t, 0.9, 0.999, 1e-8, 1.0, 1.0, st) == 0
    moved = (m.flat_param - before).abs()
    assert float(moved.max()) <= 0.001 * 1.01 and float(moved.max()) > 1e-4      # first Adam step: |dp| ~ lr
    # ---- r6 (code/hpmn.py:336: ONE sess.run per step): the same step behind ONE call, hpmn_train_step (ABI v14),
    #      on a second copy of the buffers -- the struct filled by hand, nothing but device memory from torch -- must land on
    #      the parameters, moments and loss of the call sequence above, and leave the flat gradient all-zero
    m2 = make_model(cfg, tmp_path / "one", p)
    ts = _lib.HpmnTrainStep()
    ts.scan = spec.desc(B, cfg.feature_size)
    C.memmove(C.byref(ts.read), C.byref(m2._read_desc), C.sizeof(_lib.HpmnReadDesc))
    ts.read.B = B
    ts.ids, ts.label = t_ids.data_ptr(), t_lab.data_ptr()
    ts.param, ts.grad, ts.m, ts.v = (m2.flat_param.data_ptr(), m2.flat_grad.data_ptr(), m2.flat_m.data_ptr(), m2.flat_v.data_ptr())
    ts.n_emb, ts.n_total = m2.params["Embedding/emb_mtx"].numel(), m2.flat_param.numel()
    for i, names in enumerate(m2._gru_names):
        for j, n in enumerate(names):
            ts.off_gru[i][j] = m2._offs[n]
    ts.off_read = m2._offs["User/dense/kernel"]
    memory2, last2, pred2 = torch.empty(B, K, H, device=dev), torch.empty(B, D0, device=dev), torch.empty(B, device=dev)
    d_mem2, d_last2 = torch.empty_like(memory2), torch.empty_like(last2)
    ws2 = torch.empty(lib.hpmn_scan_train_workspace_bytes(C.byref(ts.scan)), device=dev, dtype=torch.uint8)
    rws2 = torch.zeros(lib.hpmn_read_workspace_bytes(C.byref(ts.read)) // 4, device=dev)
    acc2, loss3 = torch.zeros(2, device=dev), torch.empty(3, device=dev)
    ts.memory, ts.last, ts.pred, ts.d_memory, ts.d_last = (memory2.data_ptr(), last2.data_ptr(), pred2.data_ptr(),
                                                           d_mem2.data_ptr(), d_last2.data_ptr())
    ts.scan_workspace, ts.read_workspace, ts.loss_acc, ts.loss3 = ws2.data_ptr(), rws2.data_ptr(), acc2.data_ptr(), loss3.data_ptr()
    ts.keep_prob, ts.inv_global_batch, ts.memory_reg = 1.0, 1.0 / B, cfg.memory_reg
    ts.lr_t, ts.beta1, ts.beta2, ts.eps, ts.clip = lr_t, 0.9, 0.999, 1e-8, 1.0
    m2.flat_grad.fill_(3.0)                      # dirty on purpose: clear_grad_first
    ts.clear_grad_first = 1
    assert lib.hpmn_train_step(None, C.byref(ts), st) == -1 and lib.hpmn_train_step(ctx, None, st) == -1
    assert lib.hpmn_train_step(ctx, C.byref(ts), st) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(pred2.cpu().numpy(), pred.cpu().numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(memory2.cpu().numpy(), memory.cpu().numpy(), rtol=0, atol=0)
    l3 = loss3.cpu().numpy()
    np.testing.assert_allclose(l3[:2], loss.cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(l3[2], float(ref["cross_entropy"]), rtol=2e-4, atol=1e-5)
    assert float(m2.flat_grad.abs().max()) == 0.0 and float(acc2.abs().max()) == 0.0      # consumed, cleared
    # (the table gradient's atomics and the weight-gradient slabs add in the same order in both: the updates agree to the last
    #  bits; an element whose gradient is rounding noise may take its +-lr step the other way -- none does at this seed)
    np.testing.assert_allclose(m2.flat_param.cpu().numpy(), m.flat_param.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(m2.flat_m.cpu().numpy(), m.flat_m.cpu().numpy(), rtol=0, atol=1e-7)
    lib.hpmn_train_ctx_destroy(ctx)


def test_scatter_with_the_lds_table_gives_the_same_table_gradient(dev, tmp_path):
    """r6 (HPMN_ID_HOT, on by default): the atomic scatter pre-reduces the equal ids of a wave's time segment in an LDS table
    (claimed per id with a compare-and-swap; a slot held by another id falls back to the atomic row add) and flushes one row
    add per entry.  Uniform ids (every lookup its own row: the table overflows and half of the runs take the fallback), a
    heavy-tailed law (most lookups on five rows, NOT adjacent in time -- the constant uid column is one run), the id-0 mask of
    the Hpmn class with ragged padding: the same table gradient as the plain kernel and as the sorted-segment reduction.  (C3 on Zipf(1.1) ids: 3.01 -> 2.40 ms/step; uniform 2.50 either way.)"""
    from hpmn_amd import ops
    rng = np.random.default_rng(12)
    for name, cfg, B in (("industry", cfg_industry(H=64, K=4, T=41, V=5000), 16), ("amazon", cfg_amazon(K=3, T=100, V=300), 9)):
        p = f32_params(cfg, 77)
        for law in ("uniform", "hot"):
            ids, label = rand_ids(cfg, B, 5)
            ids[:, :, 0] = ids[:, -1:, 0]
            if law == "hot":
                hot = rng.choice(np.array([7, 11, 13, 17, 19], dtype=ids.dtype), size=ids.shape[:2])
                ids[:, :, 1] = np.where(ids[:, :, 1] != 0, hot, 0)          # (padding stays padding)
            t_ids, t_lab = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
            got = {}
            for mode in ("table", "plain", "sorted"):
                m = make_model(cfg, tmp_path / (name + law + mode), p)
                object.__setattr__(m.spec, "hot_ids", mode == "table")
                m.det_scatter = mode == "sorted"
                m.compute_gradients(t_ids, t_lab, keep_prob=1.0)
                torch.cuda.synchronize()
                assert (m.last_scatter_plan is not None) == (mode == "sorted")
                got[mode] = m.grads["Embedding/emb_mtx"].cpu().numpy().copy()
                frac = m._probe_id_law(t_ids)
            if cfg.industry:                                   # (V = 5000: uniform ids really are distinct rows; the Amazon case has 300)
                assert (frac < 0.5) == (law == "hot"), (name, law, frac)
            scale = np.abs(got["plain"]).max()
            assert scale > 0
            for mode in ("table", "sorted"):
                np.testing.assert_allclose(got[mode], got["plain"], rtol=0, atol=2e-6 * scale, err_msg="%s %s %s" % (name, law, mode))
            if not cfg.industry:
                assert float(np.abs(got["table"][0]).max()) == 0.0                   # id 0 is masked


def test_one_call_step_is_the_plain_train_step(dev, tmp_path):
    """Hpmn.train_step takes the one-call form (hpmn_train_step) for the plain single-process step; switched off per instance
    (ONE_CALL_STEP = False) the same model issues the four calls + Python glue of rounds 2-5.  Five steps with dropout masks
    injected, ragged Amazon-shaped batches of two sizes (the cached descriptor is rebuilt when the batch shape changes): the
    loss of every step and the final parameters agree."""
    cfg = cfg_amazon(K=3, T=100, V=300)
    p = f32_params(cfg, 131)
    # (lr 3e-4: an element whose gradient is rounding noise takes its +-lr step either way in either form -- at the reference's
    #  3e-3 one such flip moved the fifth step's loss by 3.5e-5)
    a, b = make_model(cfg, tmp_path / "a", p, lr=3e-4), make_model(cfg, tmp_path / "b", p, lr=3e-4)
    b.ONE_CALL_STEP = False
    rng = np.random.default_rng(9)
    ces = [[], []]
    for step in range(5):
        B = 7 if step != 3 else 4
        ids, label = rand_ids(cfg, B, 200 + step)
        masks = tuple(torch.as_tensor((rng.random((B, n)) < 0.5).astype(np.float32) / 0.5).to(dev) for n in (200, 80))
        for k, mdl in enumerate((a, b)):
            assert mdl._one_call_ok(torch.as_tensor(ids).to(dev)) == (k == 0)
            out, ce = mdl.train_step(torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev), keep_prob=0.5, masks=masks)
            assert out["prediction"].shape == (B,)
            ces[k].append(float(ce))
    # (same kernels in the same order on the same streams; what differs from run to run -- in either form -- is the order of the
    #  scatter's and the loss sums' atomics: last bits of a gradient, which Adam turns into a +-lr step the other way for the
    #  rare element whose gradient IS rounding noise.  Measured: losses to 4e-6 relative at step 5)
    np.testing.assert_allclose(ces[0], ces[1], rtol=3e-5)
    assert a.adam_t == b.adam_t == 5
    for k in p:
        if k.endswith(("dense_3/bias", "dense_6/bias", "dense_9/bias")):
            continue          # (the bias in front of a softmax: its exact gradient is 0, what arrives is noise, Adam makes +-lr of it)
        d = np.abs(a.params[k].cpu().numpy() - b.params[k].cpu().numpy())
        assert int(np.sum(d > 1e-4)) <= max(4, int(2e-3 * d.size)) and float(np.median(d)) < 1e-6, (k, float(d.max()), int(np.sum(d > 1e-4)))


# ------------------------------------------------------------------------------- item branch / dual mode (code/hpmn.py:444-462, :297-317)
def _dual_case(industry, H):
    if industry:
        # the XLong literals: item side 184 steps + 8 zero steps = 192, periods [3,2,2,2,2,2,2,1], 8 layers
        u = cfg_industry(H=H, K=4, T=41, V=400)
        i = O.HpmnConfig(400, 1, 184, H, 16, 3, (3, 2, 2, 2, 2, 2, 2, 1), 8, True, 5e-5, scope="Item", front_zero=8, last_idx=-1)
    else:
        u = cfg_amazon(H=H, K=3, T=100, V=400)
        i = O.HpmnConfig(400, 2, 36, H, 16, 3, (2, 2, 3, 3, 1), 5, False, 1e-5, scope="item", last_idx=-1)   # Taobao's item side
    return u, i


@pytest.mark.parametrize("industry,H,user", [(False, 32, True), (True, 64, True), (False, 64, False), (True, 32, False)])
def test_item_branch_and_dual_mode_match_oracle(dev, tmp_path, industry, H, user):
    """item=True: repre = [user_repre, item_repre] (or item_repre alone), memory_loss = imloss + umloss; both scans
    and their BPTT on the HIP kernels (the item side brings period 3 at layer 0, D0 = 16 and 1-step layers), the
    joint read path under autograd on the device.  Forward and every gradient against float64 autograd."""
    from hpmn_amd.hpmn import Hpmn, Hpmn_Industry
    ucfg, icfg = _dual_case(industry, H)
    B = 5
    p = O.randomize_params(O.init_params(ucfg, seed=131, item_cfg=icfg, user=user), seed=132)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    rng = np.random.default_rng(133)
    uids, label = rand_ids(ucfg, B, 134)
    iids = rng.integers(0 if industry else 1, 400, size=(B, icfg.user_maxlen, icfg.user_dim)).astype(np.int32)
    if not industry:
        iids[1, :20] = 0
    cls = Hpmn_Industry if industry else Hpmn
    m = cls(str(tmp_path), [], [], 400, ucfg.user_dim, icfg.user_dim, ucfg.user_maxlen, icfg.user_maxlen, 0.001, H, 16, 3,
            list(ucfg.user_layers), list(icfg.user_layers), ucfg.user_num_layers, icfg.user_num_layers, user, True,
            memory_reg=ucfg.memory_reg, verbose=False)
    assert sorted(m.params) == sorted(p)
    m.set_params(p)
    tp = R.to_torch(p, torch.float64, requires_grad=True)
    ref = R.forward_dual(ucfg, icfg, tp, torch.as_tensor(uids.astype(np.int64)), torch.as_tensor(iids.astype(np.int64)),
                         torch.as_tensor(label.astype(np.int64)), user=user, item=True)
    ref["cross_entropy"].backward()
    tu, ti, tl = torch.as_tensor(uids).to(dev), torch.as_tensor(iids).to(dev), torch.as_tensor(label).to(dev)
    out = m.forward_inference(tu, item_ids=ti)
    np.testing.assert_allclose(out["logit"].cpu().numpy(), ref["logit"].detach().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(out["prediction"].cpu().numpy(), ref["prediction"].detach().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(out["item_weights"].cpu().numpy(), ref["item_weights"].detach().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(out["memories"][m.item_scope].cpu().numpy(), ref["item_memory"].detach().numpy(), rtol=0, atol=TOL)
    if user:
        np.testing.assert_allclose(out["user_weights"].cpu().numpy(), ref["user_weights"].detach().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(float(out["memory_loss"]), float(ref["memory_loss"]), rtol=1e-4, atol=1e-5)
    _, ce = m.compute_gradients(tu, tl, keep_prob=1.0, global_batch=B, item_ids=ti)
    np.testing.assert_allclose(float(ce), float(ref["cross_entropy"]), rtol=2e-4, atol=1e-5)
    for k in p:
        w = tp[k].grad.numpy()
        np.testing.assert_allclose(m.grads[k].cpu().numpy(), w, rtol=0, atol=2e-4 * max(1e-6, np.abs(w).max()) + 1e-6,
                                   err_msg=k)
    # a training step runs, eval through the harness stages the item side from the samples' 4th field
    m.train_step(tu, tl, keep_prob=0.5, item_ids=ti)
    samples = [(int(l), u.tolist(), 1, i.tolist(), 1) for l, u, i in zip(label, uids, iids)]
    if len(set(label.tolist())) == 2:
        auc, ll, ml = m.eval(samples, 3)
        assert 0.0 <= auc <= 1.0 and np.isfinite(ll)
    m.save_model()
    before = m.forward_inference(tu, item_ids=ti)["prediction"].clone()
    m.set_params({k: np.zeros_like(v) for k, v in m.get_params().items()})
    m.load_model()
    assert torch.equal(before, m.forward_inference(tu, item_ids=ti)["prediction"])


def test_lazy_table_adam_touches_only_the_batch_rows_and_matches_dense_on_them(dev, tmp_path):
    """lazy_table_adam=True (the labelled deviation for tables too large for the dense sweep): on the FIRST step
    (all moments zero) the rows a batch touches must end exactly where the dense update puts them, every other
    row must not move, and the dense variables must be identical; later steps differ from dense only on rows whose
    moments are non-zero but whose gradient is zero (that IS the deviation) -- checked as: untouched rows never move."""
    for cfg in (cfg_industry(H=64, K=3, T=41, V=600), cfg_amazon(K=3, T=100, V=600)):
        p = f32_params(cfg, 141)
        ids, label = rand_ids(cfg, 6, 142)
        ids = np.minimum(ids, 299)                                # rows 300.. are never touched
        md = make_model(cfg, tmp_path, p)
        from hpmn_amd.hpmn import Hpmn, Hpmn_Industry
        cls = Hpmn_Industry if cfg.industry else Hpmn
        ml = cls(str(tmp_path / "lazy"), [], [], cfg.feature_size, cfg.user_dim, 2, cfg.user_maxlen, 10, 0.003, cfg.hidden_size,
                 16, 3, list(cfg.user_layers), [2, 1], cfg.user_num_layers, 1, True, False, memory_reg=cfg.memory_reg,
                 verbose=False, lazy_table_adam=True)
        ml.set_params(p)
        assert ml.flat_grad.numel() == md.flat_grad.numel() - 600 * 16
        ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
        md.train_step(ti, tl, keep_prob=1.0)
        ml.train_step(ti, tl, keep_prob=1.0)
        for k in p:
            a, b = ml.params[k].cpu().numpy(), md.params[k].cpu().numpy()
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-6, err_msg=k)            # step 1: identical everywhere
        e0 = p["Embedding/emb_mtx"].astype(np.float32)
        for _ in range(3):
            ml.train_step(ti, tl, keep_prob=1.0)
        e = ml.params["Embedding/emb_mtx"].cpu().numpy()
        np.testing.assert_array_equal(e[300:], e0[300:])
        touched = np.unique(ids)
        touched = touched[touched != 0] if not cfg.industry else touched
        assert float(np.abs(e[touched] - e0[touched]).max()) > 1e-4
        if not cfg.industry:
            np.testing.assert_array_equal(e[0], e0[0])            # the masked row 0 never gets a gradient


def test_table_rows_beyond_2_gib_give_identical_results(dev, tmp_path, monkeypatch):
    """BASELINE configs[4] is about tables far larger than any cache: every kernel that indexes the table (gather in
    the projection / fused forward / inference chain, gradient scatter, dense and row-wise Adam) must do its row
    arithmetic in 64 bits.  A 40 M-row table (2.4 GiB per buffer) whose top 600 rows hold a small model's table, ids
    shifted to match: outputs, the gradient rows and the updated rows must be bit-identical to the small model's."""
    from hpmn_amd.hpmn import Hpmn_Industry
    cfg = cfg_industry(H=64, K=3, T=41, V=600)
    p = f32_params(cfg, 151)
    ids, label = rand_ids(cfg, 6, 152)
    base = 40_000_000 - 600
    small = make_model(cfg, tmp_path, p)
    monkeypatch.setenv("HPMN_TABLE_GRAD", "compact")     # (auto takes this form from 8 GiB per buffer; this table has 2.4)
    big = Hpmn_Industry(str(tmp_path / "big"), [], [], 40_000_000, 2, 1, 41, 1, 0.003, 64, 16, 3, [2] * 10 + [1], [1], 3, 1,
                        True, False, memory_reg=cfg.memory_reg, verbose=False)
    big.set_params({k: v for k, v in p.items() if k != "Embedding/emb_mtx"})
    with torch.no_grad():
        big.params["Embedding/emb_mtx"].zero_()
        big.params["Embedding/emb_mtx"][base:].copy_(small.params["Embedding/emb_mtx"])
    ts, tb = torch.as_tensor(ids).to(dev), torch.as_tensor(ids + base).to(dev)
    tl = torch.as_tensor(label).to(dev)
    a, b = small.forward_inference(ts), big.forward_inference(tb)
    assert torch.equal(a["logit"], b["logit"]) and torch.equal(a["memory"], b["memory"])
    small.compute_gradients(ts, tl, keep_prob=1.0)
    big.compute_gradients(tb, tl, keep_prob=1.0)
    assert big.compact_table_grad and big.grads["Embedding/emb_mtx"] is None     # (r5: a table this size keeps NO dense gradient)
    gs, gb = small.table_gradient(), big.table_gradient()
    np.testing.assert_allclose(gb[base:].cpu().numpy(), gs.cpu().numpy(), rtol=0, atol=1e-7)   # (atomics: last-bit order)
    assert float(gb[:base].abs().max()) == 0.0
    for k in p:
        if k != "Embedding/emb_mtx":
            np.testing.assert_allclose(big.grads[k].cpu().numpy(), small.grads[k].cpu().numpy(), rtol=0,
                                       atol=1e-6 * float(small.grads[k].abs().max()) + 1e-12, err_msg=k)
    small.train_step(ts, tl, keep_prob=1.0)
    big.train_step(tb, tl, keep_prob=1.0)
    np.testing.assert_allclose(big.params["Embedding/emb_mtx"][base:].cpu().numpy(),
                               small.params["Embedding/emb_mtx"].cpu().numpy(), rtol=0, atol=2e-6)
    assert float(big.params["Embedding/emb_mtx"][:base].abs().max()) == 0.0


INT64_CASES = [
    ("amazon_h32", cfg_amazon(K=3, T=100, V=300), 7),                       # all layers + gather in one launch (gru32_all)
    ("industry_h64", cfg_industry(H=64, K=4, T=41, V=500), 5),              # fused layer-0 forward + pair launches
    ("industry_h128", cfg_industry(H=128, K=3, T=41, V=500), 3),            # input_proj gather + four-wave scans
    ("taobao_f4", O.HpmnConfig(600, 4, 300, 64, 16, 3, (2, 2, 3, 5, 5, 1), 5, False, 1e-5), 4),   # D0 = 64, pairs from layer 0
    ("e4_f4", O.HpmnConfig(400, 4, 41, 64, 4, 3, (2,) * 10 + (1,), 3, True, 5e-5), 4),            # 16-byte rows (the big-table test's shape)
]


@pytest.mark.parametrize("name,cfg,B", INT64_CASES, ids=[c[0] for c in INT64_CASES])
def test_int64_ids_give_the_results_of_int32_ids(dev, tmp_path, name, cfg, B):
    """ABI v10: every entry point that takes ids takes them as int32 or int64 (HPMN_ID_I64 in the id-flags word).  Same ids in
    both widths through inference, the training forward, BPTT + scatter, the two-pass table Adam and the gather probes: the
    outputs must be IDENTICAL bit for bit (same kernels, same order; only the scatter's atomics may differ in the last bit)."""
    from hpmn_amd import ops
    p = f32_params(cfg, 171)
    ids, label = rand_ids(cfg, B, 172)
    m32 = make_model(cfg, tmp_path / "a", p)
    m64 = make_model(cfg, tmp_path / "b", p)
    for m in (m32, m64):
        m.TWO_PASS_MIN_NUMEL = 0
    t32, t64 = torch.as_tensor(ids).to(dev), torch.as_tensor(ids.astype(np.int64)).to(dev)
    tl = torch.as_tensor(label).to(dev)
    a, b = m32.forward_inference(t32), m64.forward_inference(t64)
    for k in ("logit", "memory", "user_weights"):
        assert torch.equal(a[k], b[k]), k
    oa, ca = m32.compute_gradients(t32, tl, keep_prob=1.0)
    ob, cb = m64.compute_gradients(t64, tl, keep_prob=1.0)
    assert torch.equal(oa["memory"], ob["memory"]) and abs(float(ca) - float(cb)) <= 1e-6 * abs(float(ca))   # (loss sums: atomics)
    for k in p:
        ga, gb = m32.grads[k], m64.grads[k]
        if k == "Embedding/emb_mtx":
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=0, atol=1e-6 * float(ga.abs().max()))
        else:
            assert torch.equal(ga, gb), k
    m32.train_step(t32, tl, keep_prob=1.0)
    m64.train_step(t64, tl, keep_prob=1.0)
    np.testing.assert_allclose(m64.flat_param.cpu().numpy(), m32.flat_param.cpu().numpy(), rtol=0, atol=2e-6)
    emb = m32.params["Embedding/emb_mtx"]
    if cfg.embedding_size == 16:
        assert torch.equal(ops.embed_gather_seq(t32, emb, 3, not cfg.industry), ops.embed_gather_seq(t64, emb, 3, not cfg.industry))
        # (the in-place probe combines its time slices with atomics: equal to rounding, not to the bit)
        np.testing.assert_allclose(ops.embed_gather_sum(t64, emb, not cfg.industry).cpu().numpy(),
                                   ops.embed_gather_sum(t32, emb, not cfg.industry).cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(ops.embed_gather(t32, emb, not cfg.industry), ops.embed_gather(t64, emb, not cfg.industry))


def test_a_table_of_more_rows_than_int32_holds(dev, tmp_path):
    """Row g of the scope table (BASELINE configs[4]: "embedding tables sized to 288 GB"): a table of 2.2 G rows -- more than
    an int32 id can name -- on the PRODUCT path: int64 ids end to end through the gather inside the forward kernels, the
    scatter, the row marking and the dense two-pass Adam.  Rows of 4 floats keep the four flat buffers at 35 GB each (with
    E = 16 param + grad + m + v of such a table would need 563 GB).  The top 400 rows hold a small model's table, ids shifted
    to match: logits, gradient rows and updated rows must equal the small model's; everything below stays untouched."""
    from hpmn_amd.hpmn import Hpmn_Industry
    free, _ = torch.cuda.mem_get_info()
    V = 2_200_000_000
    if free < 4.4 * V * 16:
        pytest.skip("needs ~155 GB of free HBM")
    cfg = O.HpmnConfig(400, 4, 41, 64, 4, 3, (2,) * 10 + (1,), 3, True, 5e-5)
    p = f32_params(cfg, 181)
    ids, label = rand_ids(cfg, 6, 182)
    base = V - 400
    small = make_model(cfg, tmp_path / "small", p)
    big = Hpmn_Industry(str(tmp_path / "big"), [], [], V, 4, 1, 41, 1, 0.003, 64, 4, 3, [2] * 10 + [1], [1], 3, 1,
                        True, False, memory_reg=cfg.memory_reg, verbose=False)
    assert big.params["Embedding/emb_mtx"].shape == (V, 4)
    big.set_params({k: v for k, v in p.items() if k != "Embedding/emb_mtx"})
    with torch.no_grad():
        big.params["Embedding/emb_mtx"].zero_()
        big.params["Embedding/emb_mtx"][base:].copy_(small.params["Embedding/emb_mtx"])
    ts = torch.as_tensor(ids).to(dev)
    tb = torch.as_tensor(ids.astype(np.int64) + base).to(dev)
    assert int(tb.max()) > 2 ** 31
    tl = torch.as_tensor(label).to(dev)
    a, b = small.forward_inference(ts), big.forward_inference(tb)
    assert torch.equal(a["logit"], b["logit"]) and torch.equal(a["memory"], b["memory"])
    small.compute_gradients(ts, tl, keep_prob=1.0)
    big.compute_gradients(tb, tl, keep_prob=1.0)
    assert big.compact_table_grad and big.grads["Embedding/emb_mtx"] is None     # (r5: param + m + v only, 105 GB)
    gs, gb = small.table_gradient(), big.table_gradient()                        # (spread out for this comparison: 35 GB)
    np.testing.assert_allclose(gb[base:].cpu().numpy(), gs.cpu().numpy(), rtol=0, atol=1e-7)
    assert float(gb[base - 1_000_000:base].abs().max()) == 0.0 and float(gb[:1_000_000].abs().max()) == 0.0
    assert float(gb[:base].abs().max()) == 0.0                   # (one reduction over 35 GB: nothing anywhere below the top rows)
    del gb
    small.train_step(ts, tl, keep_prob=1.0)
    big.train_step(tb, tl, keep_prob=1.0)                         # two-pass dense Adam over 2.2 G rows, int64 row marking
    torch.cuda.synchronize()
    np.testing.assert_allclose(big.params["Embedding/emb_mtx"][base:].cpu().numpy(),
                               small.params["Embedding/emb_mtx"].cpu().numpy(), rtol=0, atol=2e-6)
    assert float(big.params["Embedding/emb_mtx"][base - 1_000_000:base].abs().max()) == 0.0
    assert int(big._row_flags[base:].max()) == 0 and int(big._row_flags[::4099].max()) == 0    # the update cleared the flags it consumed
    a, b = small.forward_inference(ts), big.forward_inference(tb)
    np.testing.assert_allclose(b["logit"].cpu().numpy(), a["logit"].cpu().numpy(), rtol=0, atol=1e-5)
    del big
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_two_pass_table_adam_equals_the_dense_sweep(dev, tmp_path, monkeypatch):
    """train_step's dense table update split in two passes (rows the batch does not point at early, on the auxiliary
    stream; the batch's rows behind the scatter) must leave every buffer exactly where the one-sweep path leaves it:
    parameters and both moment buffers after several steps, gradient table and row flags all-zero in between, and a
    stand-alone compute_gradients in the middle must not confuse it.  r5: three forms -- the touched rows updated from the
    scatter's COMPACT rows with no dense gradient table at all (hpmn_rows_sum_adam, the default), the r4 two-pass step over
    a dense gradient table, the one sweep."""
    from hpmn_amd import hpmn as H
    cfg = O.HpmnConfig(4000, 2, 41, 64, 16, 3, (2, 2, 2), 3, True, 1e-5)
    p = f32_params(cfg, 71)
    batches = [rand_ids(cfg, 6, 100 + i, ragged=True) for i in range(4)]
    results = []
    for form in ("compact", "dense", "sweep"):
        monkeypatch.setattr(H.Hpmn_Basic, "TWO_PASS_TABLE_ADAM", form != "sweep")
        monkeypatch.setattr(H.Hpmn_Basic, "TWO_PASS_MIN_NUMEL", 0)
        monkeypatch.setenv("HPMN_TABLE_GRAD", "dense" if form == "dense" else ("compact" if form == "compact" else "auto"))
        m = make_model(cfg, tmp_path, p)
        assert m.compact_table_grad == (form == "compact")
        assert (m.grads["Embedding/emb_mtx"] is None) == (form == "compact")
        masks = (torch.ones(6, 200, device=dev), torch.ones(6, 80, device=dev))
        for i, (ids, label) in enumerate(batches):
            ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
            if i == 2:
                m.compute_gradients(ti, tl, keep_prob=1.0)            # leaves a table gradient behind
                if form == "compact":
                    g_compact = m.table_gradient().clone()
                elif form == "dense":
                    np.testing.assert_allclose(m.grads["Embedding/emb_mtx"].cpu().numpy(), g_compact.cpu().numpy(), rtol=1e-5,
                                               atol=1e-7)
            assert m._two_pass_table_adam(ti) == (form == "dense")
            m.train_step(ti, tl, keep_prob=1.0, masks=masks)
            if form == "dense":
                n_emb = m.params["Embedding/emb_mtx"].numel()
                assert float(m.flat_grad[:n_emb].abs().max()) == 0.0
            if form != "sweep":
                assert int(m._row_flags.max()) == 0
        torch.cuda.synchronize()
        results.append([b.clone() for b in (m.flat_param, m.flat_m, m.flat_v)])
    for other in (1, 2):
        for a, b, name in zip(results[0], results[other], ("param", "m", "v")):
            # (the atomic scatter of the dense forms may order differently run to run: last-bit differences in touched rows)
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,T,runs,H", [(6, 3, 41, 24, 64), (75, 3, 41, 24, 64), (9, 7, 1001, 9, 64), (3, 5, 297, 12, 64),
                                           (5, 3, 41, 16, 128), (4, 4, 297, 8, 128), (7, 3, 41, 16, 32)])
def test_scans_give_the_same_result_beside_an_unrelated_kernel(dev, tmp_path, B, K, T, runs, H):
    """The two-wave scan kernels hand data between waves through LDS behind progress counters; a missing condition
    there only shows when one wave is slowed down.  A 2.5 GB fill on another stream beside the step does that (it is
    how a stale projection tile in the fused forward was found): forward outputs, saved states and every gradient must
    not depend on it.  H = 128 runs the four-wave barrier kernels, H = 32 the one-wave kernels; the first-generation and
    helper-wave H = 64 kernels run this test in the subprocesses of test_fallback_kernel_paths_still_match_the_oracle."""
    from hpmn_amd import ops
    cfg = cfg_industry(H=H, K=K, T=T, V=600)           # (T + 23 zero steps: 64, 1024, 320 -> odd upper layers)
    p = f32_params(cfg, 151)
    ids, label = rand_ids(cfg, B, 152)
    m = make_model(cfg, tmp_path, p)
    ti, tl = torch.as_tensor(ids).to(dev), torch.as_tensor(label).to(dev)
    junk = torch.empty(640_000_000, device=dev)
    side = torch.cuda.Stream()

    def run(concurrent):
        if concurrent:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(1 if T < 100 else 8):           # (long enough to cover the step)
                    junk.zero_()
        memory, last, saved = ops.scan_forward_train(m.spec, ti, m.params["Embedding/emb_mtx"], m._gru_weights())
        states = [s[1].clone() for s in saved] + [s[2].clone() for s in saved]
        out, _ = m.compute_gradients(ti, tl, keep_prob=1.0)
        torch.cuda.synchronize()
        return [memory.clone(), last.clone()] + states, {k: v.clone() for k, v in m.grads.items()}

    ref_f, ref_g = run(False)
    for it in range(runs):
        f, g = run(it % 3 != 2)
        for i, (a, b) in enumerate(zip(ref_f, f)):
            assert torch.equal(a, b), "forward tensor %d differs in run %d" % (i, it)
        for k in ref_g:
            # (the scatter's fp32 atomics may order differently run to run)
            tol = 1e-6 if k == "Embedding/emb_mtx" else 0.0
            assert float((ref_g[k] - g[k]).abs().max()) <= tol, "%s differs in run %d" % (k, it)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"HPMN_BWD_HELPER": "1", "HPMN_FUSED_FWD_GEN": "1"},                       # the first-generation two-wave kernels
    {"HPMN_BWD_HELPER": "0", "HPMN_FUSED_FWD": "0", "HPMN_TWO_PASS_ADAM": "0"},  # one wave per sequence, two-kernel forward
    {"HPMN_BWD_DX_WAVE": "0", "HPMN_L0_SPLIT": "1"},                           # separate input-gradient launches, layer 0 split in time
    {"HPMN_PAIR_FWD": "0", "HPMN_PAIR_BWD": "0", "HPMN_PAIR_INFER": "0"},      # one launch per layer (no two-layer launches)
    {"HPMN_PAIR_FWD": "1", "HPMN_PAIR_BWD": "1", "HPMN_FUSED_SCATTER": "1"},   # the other pairing; scatter fused into layer 0's launch
    {"HPMN_FUSED_SCATTER": "2"},                                               # (r5) the scatter inside the LOOP of layer 0's reverse scan
    {"HPMN_PAIR_SINGLE": "0"},                                                 # (r6) two sequences per workgroup in the two-layer launches
                                                                               # whatever the batch (default up to 128 sequences: one)
], ids=["gen1", "one-wave", "dx-launches+split", "no-pairs", "pairs-alt", "scatter-in-loop", "pairs-two-per-wg"])
def test_fallback_kernel_paths_still_match_the_oracle(env):
    if env.get("HPMN_FUSED_FWD_GEN") == "1":
        _needs_legacy_build()
    """The switches of DESIGN_HISTORY.md 3.11 select kernels at library load, so each set runs a slice of this file in a
    process of its own: H = 64 forward/gradient parity at the tiny and odd lengths and at the XLong length."""
    import subprocess
    e = dict(os.environ)
    e.update(env)
    here = os.path.dirname(os.path.abspath(__file__))
    # (the in-loop scatter only exists at layer 0 of D <= 32 graphs, and its steps take twice as long: the XLong-shaped cases)
    pick = ("xlong_c3_shape or c_abi_alone" if env.get("HPMN_FUSED_SCATTER") == "2" else
            "(tiny_and_odd and 64) or xlong_c3_shape or c_abi_alone or (beside_an_unrelated and 6-3-41)"
            " or (beside_an_unrelated and 3-5-297)")
    if "HPMN_PAIR_SINGLE" in env:
        pick += " or xlong_c3_b66 or two_layers_in_one_launch"
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", pick],
                       env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_retired_kernel_generations_in_a_legacy_build(tmp_path):
    """Opt-in (HPMN_TEST_LEGACY=1: it compiles a second library, minutes): build with -DHPMN_LEGACY_KERNELS and run the tests
    of the retired generations -- first-generation fused forward + helper-wave reverse scan, training on the tile kernel --
    against it in a process of its own (HPMN_LIB_PATH)."""
    if os.environ.get("HPMN_TEST_LEGACY") != "1":
        pytest.skip("opt-in: HPMN_TEST_LEGACY=1")
    import subprocess
    from hpmn_amd import build
    lib = build.build_library(out=str(tmp_path / "libhpmn_hip_legacy.so"), flags=["-DHPMN_LEGACY_KERNELS"])
    e = dict(os.environ, HPMN_LIB_PATH=lib)
    e.pop("HPMN_TEST_LEGACY", None)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "pipelined_mfma_scan or (fallback_kernel_paths and gen1)"],
                       env=e, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_eight_wave_h128_scans_match_the_oracle():
    """HPMN_SCAN128_WAVES=8 (gru_scan128.hip: eight waves per sequence, measured slower and therefore off by default): H = 128
    forward / gradient parity at the tiny and odd lengths, the K = 3 case and the wide batch, in a process of its own."""
    import subprocess
    e = dict(os.environ, HPMN_SCAN128_WAVES="8")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "(tiny_and_odd and 128) or industry_h128"],
                       env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_h128_alternatives_of_round_4_match_the_oracle():
    """The other side of the third part's switches, in a process of its own: H = 128 scans capped at one workgroup per CU
    (HPMN_SCAN128_SOLO=3, the LDS padding), the row-per-lane input gradient (HPMN_DX_LDS=0), the weight gradients' column groups
    in grid order (HPMN_WGRAD_XCD=0) and layer 0's whole-CU weight gradient cut into three time pieces (HPMN_WGRAD_TSPLIT=3,
    H = 64): H = 128 forward / gradient parity, the input-gradient launch, and the H = 64 XLong gradient case."""
    import subprocess
    # (r5: + the H = 128 slab reductions on the second helper stream, HPMN_WGRAD_REDUCE_ASIDE=1 -- built, measured neutral, off by
    #  default; the slice is trimmed to the cases that reach these switches, the suite has to stay under eight minutes)
    e = dict(os.environ, HPMN_SCAN128_SOLO="3", HPMN_DX_LDS="0", HPMN_WGRAD_XCD="0", HPMN_WGRAD_TSPLIT="3",
             HPMN_WGRAD_REDUCE_ASIDE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "(tiny_and_odd and 128) or industry_h128 or input_gradient_launch_h128 or xlong_c3_shape"],
                       env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------- two layers per launch
PAIR_CASES = [
    # name, B, T_lo, D_lo (gather F = D/16 or x rows), period_lo, period_up
    ("xlong-l0l1", 9, 1024, "g32", 2, 2),
    ("xlong-l2l3", 5, 256, "x64", 2, 2),
    ("top", 4, 32, "x64", 2, 2),
    ("taobao-l0l1", 7, 300, "g64", 2, 2),
    ("taobao-l2l3", 3, 75, "x64", 3, 5),
    ("period5", 2, 400, "g32", 5, 1),
    ("one-block", 3, 16, "g32", 2, 1),
    ("odd-rows", 6, 46, "x64", 2, 1),
    ("wide", 301, 64, "g32", 2, 2),
]


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("name,B,T,src,p_lo,p_up", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_two_layers_in_one_launch_equal_two_launches(dev, name, B, T, src, p_lo, p_up, train):
    """hpmn_gru_pair_fwd against two hpmn_gru_fused_fwd calls: the same arithmetic in the same order, so every output
    -- final states, subsampled outputs, saved states and gates, the materialised gather -- must be BIT-identical;
    with either SIMD assignment of the upper layer's waves, beside an unrelated kernel too."""
    from hpmn_amd import ops
    H, E, V = 64, 16, 700
    g = torch.Generator(device="cpu").manual_seed(hash(name) % 1000)
    D = int(src[1:])
    gather = src[0] == "g"
    Tu = T // p_lo
    assert T % p_lo == 0 and Tu % p_up == 0

    def w(*shape, scale=0.3):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    wl = dict(wg=w(D + H, 2 * H), bg=w(2 * H) + 1, wc=w(D + H, H), bc=w(H))
    wu = dict(wg=w(2 * H, 2 * H), bg=w(2 * H) + 1, wc=w(2 * H, H), bc=w(H))
    inp = {}
    front = 0
    if gather:
        front = 23 if T > 64 else 0
        ids = torch.randint(0, V, (B, T - front, D // E), generator=g, dtype=torch.int32)
        ids[: B // 2, : (T - front) // 3] = 0
        inp = dict(ids=ids.to(dev), emb=w(V, E, scale=1.0), front_zero=front, mask_id0=(name != "xlong-l0l1"))
    else:
        inp = dict(x=w(B, T, D, scale=1.0))

    def outs():
        mem = torch.full((B, 2, H), 7.0, device=dev)
        def layer(Tl, Dl, p, want_y):
            y = torch.full((B, Tl // p, H), 7.0, device=dev) if want_y else None
            hs = torch.full((B, Tl + 1, H), 7.0, device=dev) if train else None
            ga = torch.full((B, Tl, 3 * H), 7.0, device=dev) if train else None
            return [y, hs, ga, None]
        lo, up = layer(T, D, p_lo, True), layer(Tu, H, p_up, p_up > 0 and name != "top")
        if gather and train:
            lo[3] = torch.full((B, T, D), 7.0, device=dev)
        return mem, lo, up

    mem_a, lo_a, up_a = outs()
    ops.gru_fused_fwd(**inp, **wl, H=H, T=T, h_last=mem_a[:, 0], period=p_lo, out=tuple(lo_a))
    ops.gru_fused_fwd(x=lo_a[0], **wu, H=H, T=Tu, h_last=mem_a[:, 1], period=p_up, out=tuple(up_a))
    junk = torch.empty(200_000_000, device=dev)
    side = torch.cuda.Stream()
    for flags in (0, 1, 0, 1):
        mem_b, lo_b, up_b = outs()
        if flags:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                junk.zero_()
        ops.gru_pair_fwd(dict(**inp, **wl, H=H, T=T, h_last=mem_b[:, 0], period=p_lo, out=tuple(lo_b)),
                         dict(**wu, H=H, T=Tu, h_last=mem_b[:, 1], period=p_up, out=tuple(up_b)), flags=flags)
        torch.cuda.synchronize()
        assert torch.equal(mem_a, mem_b), "final states (flags %d)" % flags
        for i, (a, b) in enumerate(zip(lo_a + up_a, lo_b + up_b)):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b), "output %d differs (flags %d)" % (i, flags)


PAIR_BWD_CASES = [
    # name, B, T_lo, D_lo, period_lo, period_up (0: the upper layer is the top layer, no d_y)
    ("xlong-l1l0", 9, 1024, 32, 2, 2),
    ("xlong-l3l2", 5, 256, 64, 2, 2),
    ("top", 4, 32, 64, 2, 0),
    ("taobao-l3l2", 3, 75, 64, 3, 5),
    ("period5", 2, 400, 32, 5, 1),
    ("one-block", 3, 32, 16, 2, 2),
    ("odd-rows", 6, 46, 64, 2, 0),
    ("odd-upper", 3, 27, 32, 3, 3),
    ("wide", 301, 64, 32, 2, 2),
]


@pytest.mark.parametrize("name,B,T,D,p_lo,p_up", PAIR_BWD_CASES, ids=[c[0] for c in PAIR_BWD_CASES])
def test_two_reverse_scans_in_one_launch_equal_two_launches(dev, name, B, T, D, p_lo, p_up):
    """hpmn_gru_pair_bwd against two hpmn_gru_scan_bwd calls (the upper with its input gradient as the epilogue, the lower
    reading it back as d_y): the same arithmetic (the compiler contracts a few multiply-adds of the coefficient
    arithmetic differently in the two kernels, so last-bit differences are allowed: 5e-6 of each tensor's max after up to
    1024 steps) -- with either SIMD assignment, beside an unrelated kernel too."""
    from hpmn_amd import ops
    H = 64
    g = torch.Generator(device="cpu").manual_seed(len(name) * 7 + B)
    Tu = T // p_lo
    assert T % p_lo == 0 and (p_up == 0 or Tu % p_up == 0)

    def w(*shape, scale=0.3):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    def layer(Tl, Dl):
        gates = torch.rand(B, Tl, 3 * H, generator=g)
        gates[..., 2 * H:] = gates[..., 2 * H:] * 2 - 1                # r, u in (0,1), c in (-1,1)
        return dict(wg=w(Dl + H, 2 * H), wc=w(Dl + H, H), D=Dl, hs=w(B, Tl + 1, H, scale=0.7), gates=gates.to(dev))
    lo, up = layer(T, D), layer(Tu, H)
    d_mem = w(B, 2, H, scale=0.1)
    d_y_up = w(B, Tu // p_up, H, scale=0.1) if p_up else None
    per_up = p_up if p_up else 1

    # reference: two launches
    dact_up = torch.full((B, Tu, 3 * H), 7.0, device=dev)
    dx_up = torch.full((B, Tu, H), 7.0, device=dev)
    ops.gru_scan_bwd(up["wg"], up["wc"], H, up["hs"], up["gates"], d_mem[:, 1], d_y_up, per_up, out=dact_up, d_x=dx_up)
    dact_lo = torch.full((B, T, 3 * H), 7.0, device=dev)
    dx_lo = torch.full((B, T, D), 7.0, device=dev)
    ops.gru_scan_bwd(lo["wg"], lo["wc"], D, lo["hs"], lo["gates"], d_mem[:, 0], dx_up, p_lo, out=dact_lo, d_x=dx_lo)
    junk = torch.empty(200_000_000, device=dev)
    side = torch.cuda.Stream()
    for flags in (0, 1, 0, 1):
        a_up = torch.full((B, Tu, 3 * H), 7.0, device=dev)
        a_lo = torch.full((B, T, 3 * H), 7.0, device=dev)
        x_lo = torch.full((B, T, D), 7.0, device=dev)
        if flags:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                junk.zero_()
        ops.gru_pair_bwd(dict(**lo, d_h_last=d_mem[:, 0], period=p_lo, d_act=a_lo, d_x=x_lo),
                         dict(**up, d_h_last=d_mem[:, 1], period=per_up, d_act=a_up, d_y=d_y_up), flags=flags)
        torch.cuda.synchronize()
        for what, got, want in (("upper d_act", a_up, dact_up), ("lower d_act", a_lo, dact_lo), ("lower d_x", x_lo, dx_lo)):
            err = float((got - want).abs().max()) / float(want.abs().max())
            # (D <= 32: the single-layer launch forms d_x on the bf16 matrix pipe with split operands, the pair's epilogue in fp32)
            tol = 2e-5 if what == "lower d_x" and D <= 32 else 5e-6
            assert err <= tol, "%s (flags %d): %g of the tensor's max" % (what, flags, err)


@pytest.mark.parametrize("D", [32, 128])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 33), (7, 250), (64, 257)])
def test_input_gradient_launch_h128(dev, B, T, D):
    """hpmn_gru_input_grad at H = 128 (bf16 matrix pipe, split operands, rows streamed through a register ring across tile
    borders: input_proj.hip) against float64: <= 2e-5 of the tensor's max; partial last tile, one row, a time chunk."""
    from hpmn_amd import ops
    H = 128
    g = torch.Generator(device="cpu").manual_seed(B * 100 + T + D)
    d_act = (torch.randn(B, T, 3 * H, generator=g) * torch.rand(B, T, 1, generator=g)).to(dev)
    wg = (torch.randn(D + H, 2 * H, generator=g) * 0.3).to(dev)
    wc = (torch.randn(D + H, H, generator=g) * 0.3).to(dev)
    want = d_act.double() @ torch.cat([wg[:D].double(), wc[:D].double()], dim=1).t()
    scale = float(want.abs().max())
    got = ops.gru_input_grad(d_act, wg, wc, D)
    torch.cuda.synchronize()
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale
    if T >= 33:
        out = torch.full((B, T, D), 7.0, device=dev)
        ops.gru_input_grad(d_act, wg, wc, D, out=out, t_range=(5, T - 9))
        torch.cuda.synchronize()
        assert float((out[:, 5:T - 4].double() - want[:, 5:T - 4]).abs().max()) <= 2e-5 * scale
        assert bool((out[:, :5] == 7.0).all()) and bool((out[:, T - 4:] == 7.0).all())


@pytest.mark.parametrize("D", [16, 32])
@pytest.mark.parametrize("T", [1, 2, 15, 16, 17, 18, 31, 33, 34, 47, 50, 65, 200, 1001])
def test_input_gradient_formed_inside_the_reverse_scan(dev, D, T):
    """The layer's input gradient out of the reverse-scan launch (HpmnGruBwd.d_x; for D <= 32 formed in the loop on the bf16
    matrix pipe with split operands, gru_scan_bwd_feed.hip) against the fp32 launch of its own (hpmn_gru_input_grad) and
    float64, over the block structure's edge lengths (16-iteration blocks, the 16-iteration unrolled loop starting at
    iteration 2, odd lengths), an odd batch, and a launch cut in two time chunks; d_act must not depend on who forms d_x."""
    from hpmn_amd import ops
    H, B = 64, 5
    if not ops.scan_bwd_fuses_dx(H, B):
        pytest.skip("the scan launch does not produce d_x in this configuration")
    g = torch.Generator(device="cpu").manual_seed(T * 10 + D)

    def w(*shape, scale=0.3):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    gates = torch.rand(B, T, 3 * H, generator=g)
    gates[..., 2 * H:] = gates[..., 2 * H:] * 2 - 1
    gates = gates.to(dev)
    wg, wc, hs = w(D + H, 2 * H, scale=0.12), w(D + H, H, scale=0.12), w(B, T + 1, H, scale=0.7)
    d_last = w(B, H, scale=0.1)
    period = 1
    d_y = w(B, T, H, scale=0.1)
    plain = ops.gru_scan_bwd(wg, wc, D, hs, gates, d_last, d_y, period)                  # no d_x: the scan alone
    dx = torch.full((B, T, D), 7.0, device=dev)
    d_act = ops.gru_scan_bwd(wg, wc, D, hs, gates, d_last, d_y, period, d_x=dx)
    torch.cuda.synchronize()
    # (the two kernel variants contract a few multiply-adds of the coefficient arithmetic differently: last bits)
    assert float((plain - d_act).abs().max()) <= 2e-6 * float(plain.abs().max())
    want32 = ops.gru_input_grad(d_act, wg, wc, D)
    w64 = torch.cat([wg[:D].double(), wc[:D].double()], dim=1)                           # [D, 3H]
    want64 = d_act.double() @ w64.t()
    scale = float(want64.abs().max())
    assert float((want32.double() - want64).abs().max()) <= 2e-6 * scale
    err = float((dx.double() - want64).abs().max()) / scale
    assert err <= 2e-5, "in-launch d_x: %g of the tensor's max" % err
    if T >= 34:
        cut = (T // 2) // 2 * 2
        carry = torch.zeros(B, H, device=dev)
        dx2 = torch.full((B, T, D), 7.0, device=dev)
        da2 = torch.full((B, T, 3 * H), 7.0, device=dev)
        ops.gru_scan_bwd(wg, wc, D, hs, gates, d_last, d_y, period, out=da2, t_range=(cut, T), dh_carry=carry, d_x=dx2)
        ops.gru_scan_bwd(wg, wc, D, hs, gates, d_last, d_y, period, out=da2, t_range=(0, cut), dh_carry=carry, d_x=dx2)
        torch.cuda.synchronize()
        assert float((da2 - d_act).abs().max()) <= 5e-6 * float(d_act.abs().max())
        assert float((dx2.double() - want64).abs().max()) / scale <= 2e-5


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("B,T,D", [(5, 200, 32), (3, 46, 64), (130, 64, 32)])
def test_saved_gates_without_the_candidate(dev, pair, B, T, D):
    """ABI v11 (HPMN_FWD_NO_CANDIDATE / HPMN_BWD_CANDIDATE_FROM_HS, include/hpmn_hip.h): the forward leaves the candidate third of
    `gates` UNWRITTEN (it stays NaN here) and everything else bit-identical; the reverse scan never reads it and recovers the
    two coefficients the candidate enters from the saved states -- d_act and d_x within 2e-6 of each tensor's max of the
    stored-candidate launch, with update gates saturated both ways (u == 1.0f exactly on some units: the candidate cannot be
    recovered there and both coefficients are 0; u ~ 1e-8 on others) -- single-layer launches and the two-layer launches."""
    from hpmn_amd import ops
    H = 64
    if not ops.candidate_elision(H, B):
        pytest.skip("candidate elision is off in this build / environment")
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    p_lo, p_up = 2, 1
    Tu = T // p_lo

    def w(*shape, scale=0.3):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    def weights(Dl):
        bg = w(2 * H) + 1
        bg[H + 3] = 30.0; bg[H + 17] = 19.0; bg[H + 40] = -18.0; bg[H + 41] = 12.0      # saturated / nearly saturated update gates
        return dict(wg=w(Dl + H, 2 * H), bg=bg, wc=w(Dl + H, H), bc=w(H))
    wl, wu = weights(D), weights(H)
    if D == 32:                                       # (the two-layer launch takes a 32-wide lower layer in gather form only)
        inp = dict(ids=torch.randint(0, 500, (B, T, 2), generator=g, dtype=torch.int32).to(dev), emb=w(500, 16, scale=1.0))
    else:
        inp = dict(x=w(B, T, D, scale=1.0))

    def forward(no_c):
        mem = torch.zeros(B, 2, H, device=dev)
        def bufs(Tl, p):
            ga = torch.full((B, Tl, 3 * H), float("nan"), device=dev)
            return [torch.zeros(B, Tl // p, H, device=dev), torch.zeros(B, Tl + 1, H, device=dev), ga, None]
        lo, up = bufs(T, p_lo), bufs(Tu, p_up)
        if pair:
            ops.gru_pair_fwd(dict(**inp, **wl, H=H, T=T, h_last=mem[:, 0], period=p_lo, out=tuple(lo), no_candidate=no_c),
                             dict(**wu, H=H, T=Tu, h_last=mem[:, 1], period=p_up, out=tuple(up), no_candidate=no_c))
        else:
            ops.gru_fused_fwd(**inp, **wl, H=H, T=T, h_last=mem[:, 0], period=p_lo, out=tuple(lo), no_candidate=no_c)
            ops.gru_fused_fwd(x=lo[0], **wu, H=H, T=Tu, h_last=mem[:, 1], period=p_up, out=tuple(up), no_candidate=no_c)
        return mem, lo, up
    mem_a, lo_a, up_a = forward(False)
    mem_b, lo_b, up_b = forward(True)
    assert torch.equal(mem_a, mem_b)
    for a, b in ((lo_a, lo_b), (up_a, up_b)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                       # y, hs
        assert torch.equal(a[2][..., :2 * H], b[2][..., :2 * H])                        # r, u
        assert not torch.isnan(a[2]).any() and torch.isnan(b[2][..., 2 * H:]).all()     # the candidate: written / untouched
    u_lo = lo_a[2][..., H:2 * H]
    assert float((u_lo == 1.0).float().mean()) > 0.01 and float((u_lo < 1e-6).float().mean()) > 0.005   # both saturations occur

    d_mem = w(B, 2, H, scale=0.1)

    def backward(lo, up, from_hs):
        a_up = torch.full((B, Tu, 3 * H), 7.0, device=dev)
        a_lo = torch.full((B, T, 3 * H), 7.0, device=dev)
        x_lo = torch.full((B, T, D), 7.0, device=dev)
        kl = dict(wg=wl["wg"], wc=wl["wc"], D=D, hs=lo[1], gates=lo[2])
        ku = dict(wg=wu["wg"], wc=wu["wc"], D=H, hs=up[1], gates=up[2])
        if pair:
            ops.gru_pair_bwd(dict(**kl, d_h_last=d_mem[:, 0], period=p_lo, d_act=a_lo, d_x=x_lo, candidate_from_hs=from_hs),
                             dict(**ku, d_h_last=d_mem[:, 1], period=p_up, d_act=a_up, d_y=None, candidate_from_hs=from_hs))
        else:
            x_up = torch.full((B, Tu, H), 7.0, device=dev)
            ops.gru_scan_bwd(ku["wg"], ku["wc"], H, ku["hs"], ku["gates"], d_mem[:, 1], None, p_up, out=a_up, d_x=x_up,
                             candidate_from_hs=from_hs)
            ops.gru_scan_bwd(kl["wg"], kl["wc"], D, kl["hs"], kl["gates"], d_mem[:, 0], x_up, p_lo, out=a_lo, d_x=x_lo,
                             candidate_from_hs=from_hs)
        torch.cuda.synchronize()
        return a_up, a_lo, x_lo
    want = backward(lo_a, up_a, False)
    got = backward(lo_b, up_b, True)                  # (gates with a NaN candidate third: reading it would poison everything)
    for what, gt, wt in zip(("upper d_act", "lower d_act", "lower d_x"), got, want):
        assert torch.isfinite(gt).all(), what
        err = float((gt - wt).abs().max()) / float(wt.abs().max())
        # (a 32-wide d_x of the single-layer launch comes off the bf16 matrix pipe: each side is within ~5e-6 of the exact product)
        tol = 2e-5 if what == "lower d_x" and D <= 32 and not pair else 2e-6
        assert err <= tol, "%s: %g of the tensor's max" % (what, err)


@pytest.mark.parametrize("mask", [True, False])
@pytest.mark.parametrize("F", [1, 2, 3, 4])
def test_gather_consumed_in_place_equals_the_gathered_rows_summed(dev, mask, F):
    """hpmn_embed_gather_sum (the roofline probe of the in-place gather) against hpmn_embed_gather + a sum over time."""
    from hpmn_amd import ops
    g = torch.Generator().manual_seed(5 + F)
    V, E, B, T = 1000, 16, 37, 211
    emb = torch.randn(V, E, generator=g).to(dev)
    ids = torch.randint(0, V, (B, T, F), generator=g, dtype=torch.int32)
    ids[:, :50] = 0
    ids = ids.to(dev)
    got = ops.embed_gather_sum(ids, emb, mask)
    want = ops.embed_gather(ids.view(B * T, F), emb, mask).view(B, T, F * E).double().sum(1)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-4)


# ------------------------------------------------------------------------------- read path: weight gradients from the tape
@pytest.mark.parametrize("industry,H,K", [(False, 32, 3), (True, 64, 4), (True, 64, 12)])
def test_read_path_weight_gradients_deferred_equal_immediate_and_survive_batch_changes(dev, tmp_path, industry, H, K):
    """hpmn_read_fwd_bwd with d_params == NULL + hpmn_read_param_grads (the tape -> product launch -> 16-slab sum, what the
    training step runs on its auxiliary stream) gives bit-identical gradients to the one-call form, and the workspace can be
    reused across batch sizes (its tape part moves with B, its slab part must not pick up stale bytes)."""
    from hpmn_amd import ops
    cfg = cfg_industry(H=H, K=K, T=64 if K == 12 else 41) if industry else cfg_amazon(H=H, K=K, T=50)
    if K == 12:
        cfg = O.HpmnConfig(feature_size=500, user_dim=2, user_maxlen=4096 - 23, hidden_size=H, embedding_size=16, hop=3,
                           user_layers=(2,) * 11 + (1,), user_num_layers=K, industry=True, memory_reg=5e-5)
    m = make_model(cfg, tmp_path, f32_params(cfg, 5))
    D0 = cfg.user_dim * cfg.embedding_size
    g = torch.Generator(device=dev).manual_seed(3)
    results = {}
    for B in (7, 3, 7, 16):
        memory = torch.randn(B, K, H, device=dev, generator=g) * 0.4
        last = torch.randn(B, D0, device=dev, generator=g) * 0.4
        label = torch.randint(0, 2, (B,), device=dev, dtype=torch.int32, generator=g)
        grads = []
        for defer in (False, True):
            m._read_grads.zero_()
            out = ops.read_fwd_bwd(m._read_desc, m._read_params, m._read_grads, memory, last, label, None, 1.0, 1.0 / B,
                                   cfg.memory_reg, defer_param_grads=defer)
            if defer:
                assert float(m._read_grads.abs().max()) == 0.0        # nothing yet: the training launch alone ran
                out.pop("reduce_param_grads")()
            torch.cuda.synchronize()
            grads.append(m._read_grads.clone())
        assert torch.equal(grads[0], grads[1])
        assert float(grads[0].abs().max()) > 0.0
        results.setdefault(B, []).append((memory, last, label, grads[0]))
    # the same inputs through a FRESH workspace give the same bits (B = 7 ran before and after B = 3)
    ops._read_ws.clear()
    memory, last, label, want = results[7][1]
    m._read_grads.zero_()
    ops.read_fwd_bwd(m._read_desc, m._read_params, m._read_grads, memory, last, label, None, 1.0, 1.0 / 7, cfg.memory_reg)
    torch.cuda.synchronize()
    assert torch.equal(m._read_grads, want)


def test_inference_read_launch_with_four_samples_per_workgroup_equals_two(dev, tmp_path):
    """hpmn_read_fwd takes four samples per workgroup once the batch reaches 1024 rows (DESIGN_HISTORY.md 3.15): the same batch in
    slices of 500 (two per workgroup) must give the same predictions, first-hop weights and memory loss -- per-sample
    arithmetic does not depend on the tile a sample sits in -- incl. a batch size that leaves a partial last tile."""
    from hpmn_amd import ops
    for cfg in (cfg_industry(H=64, K=7, T=41, V=400), cfg_amazon(H=32, K=4, T=50, V=300)):
        m = make_model(cfg, tmp_path, f32_params(cfg, 9))
        K, H, D0 = cfg.user_num_layers, cfg.hidden_size, cfg.user_dim * cfg.embedding_size
        B = 1027
        g = torch.Generator(device=dev).manual_seed(11)
        memory = torch.randn(B, K, H, device=dev, generator=g) * 0.5
        last = torch.randn(B, D0, device=dev, generator=g) * 0.5
        big = ops.read_fwd(m._read_desc, m._read_params, memory, last, want_logit=True, want_att=True)
        preds, logits, atts, ml = [], [], [], 0.0
        for lo in range(0, B, 500):
            o = ops.read_fwd(m._read_desc, m._read_params, memory[lo:lo + 500].contiguous(), last[lo:lo + 500].contiguous(),
                             want_logit=True, want_att=True)
            preds.append(o["prediction"]); logits.append(o["logit"]); atts.append(o["user_weights"])
            ml += float(o["memory_loss"])
        assert torch.equal(big["logit"], torch.cat(logits))
        assert torch.equal(big["prediction"], torch.cat(preds))
        assert torch.equal(big["user_weights"], torch.cat(atts))
        np.testing.assert_allclose(float(big["memory_loss"]), ml, rtol=1e-5)


# ------------------------------------------------------------------------------- r5: the default evaluation path at its own length
def test_tile_eval_path_at_the_xlong_shape_matches_the_oracle(dev, tmp_path):
    """VERDICT r4 weak #1b: Hpmn.eval from 1 536 rows runs gru_pipe_fwd_kernel -- 16-sequence tiles on split-f16 operands,
    three products, fp32 accumulate -- and until r5 its only oracle check in the DEFAULT build was 128 steps long.  Here at
    the length the 1.3 M sequences/s are quoted on: T = 1001 (1024 recurrent steps), K = 7, H = 64.
    (a) ops.tiled_forward_inference on B = 37 (two full tiles and a partial one) against the float64 oracle: memory and, through
        the read path, logit / prediction / first-hop weights at the north_star tolerance 1e-4;
    (b) one evaluation-sized pass of 2 000 rows through Hpmn.forward_inference -- which must take the tile path by itself --
        against the oracle on 96 of its rows (first tile, a middle tile, the last, partial, tile) and against the
        per-sequence fp32 kernels on all rows."""
    from hpmn_amd import ops
    cfg = cfg_industry(H=64, K=7, T=1001, V=5000)
    p = f32_params(cfg, 301)
    m = make_model(cfg, tmp_path, p)
    emb, w = m.params["Embedding/emb_mtx"], m._gru_weights()
    # (a)
    ids, label = rand_ids(cfg, 37, 302)
    want = O.forward(cfg, p, ids, label)
    t = torch.as_tensor(ids).to(dev)
    mem, last = ops.tiled_forward_inference(m.spec, t, emb, w, group=1)
    assert ops.pipe_error_word(m.spec.K, 37, dev) == 0
    np.testing.assert_allclose(mem.cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    out = ops.read_fwd(m._read_desc, m._read_params, mem, last, True, True)
    for k in ("logit", "prediction", "user_weights"):
        np.testing.assert_allclose(out[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)
    # (b)
    n = 2000
    assert m._tiled_inference(n)
    big, _ = rand_ids(cfg, n, 303)
    tb = torch.as_tensor(big).to(dev)
    got = m.forward_inference(tb)
    pick = np.r_[0:32, 992:1024, n - 32:n]
    wantb = O.forward(cfg, p, big[pick])
    for k in ("memory", "logit", "prediction"):
        np.testing.assert_allclose(got[k][pick].cpu().numpy(), wantb[k], rtol=0, atol=TOL, err_msg=k)
    m.TILED_EVAL_MIN_ROWS = 0                                    # the per-sequence fp32 kernels on the same rows
    ref = m.forward_inference(tb)
    assert float((got["memory"] - ref["memory"]).abs().max()) <= 5e-5
    assert float((got["logit"] - ref["logit"]).abs().max()) <= 1e-4


def test_split_gradient_kernels_track_the_fp32_kernels_over_200_steps(tmp_path):
    """VERDICT r4 #2b: the GRU weight gradients and layer 0's input gradient of the timed region are three bf16 products on
    split operands (fp32 accumulate; 5e-6 of max|grad| per step against 6e-7 for the fp32 kernels).  200 training steps from the
    same weights on the same batches, once on the default kernels and once with every product on fp32 kernels
    (bench.ALL_FP32_ENV; the switches are read once per process, hence two processes): the loss curves must agree to 1e-3
    relative at every step and the trained models' predictions on four of the batches to 1e-3.  Deterministic scatter in both, so
    that the kernels' precision is the ONLY difference (two fp32 runs then differ by 3e-7: the loss sums' atomics).
    Learning rate 1e-4 (loss 0.693 -> 0.546 over the 200 steps; measured deviation 1.8e-5).  At the reference's 1e-3 this problem
    -- 25 batches the model ends up memorising -- is chaotic in the plain sense: the curves agree to 1e-3 for 65 steps, then
    the step at which the loss collapses shifts and the runs decorrelate (tools/traj_cmp.py: at 3e-4 the first excess is at
    step 140); that says nothing about either gradient, so the test stays where a trajectory comparison means something."""
    import subprocess
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    outs = []
    for tag, extra in (("split", {}), ("fp32", bench.ALL_FP32_ENV)):
        env = dict(os.environ)
        env.update(extra)
        env["HPMN_DET_SCATTER"] = "1"
        dst = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "traj_worker.py"), dst], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(np.load(dst))
    a, b = outs
    la, lb = a["loss"], b["loss"]
    assert len(la) == 200 and np.isfinite(la).all() and la[-20:].mean() < 0.9 * la[:20].mean()       # (it trains)
    np.testing.assert_allclose(la, lb, rtol=1e-3, atol=0)
    np.testing.assert_allclose(a["pred_final"], b["pred_final"], rtol=0, atol=1e-3)      # the two trained models agree


def test_split_gradient_kernels_are_fp32_equivalent(tmp_path):
    """r6 (the reference is fp32 throughout, code/hpmn.py:119-120,209-214): the GRU weight gradients, layer 0's
    in-loop input gradient and the H = 128 projection / input-gradient kernels run on the bf16 matrix pipe with THREE planes per
    operand and the six products of order <= 2 (what is dropped is below 2^-24 of a product) -- fp32-equivalent, like the read
    path's training launch since r5; rounds 4/5 ran two planes / three products (~5e-6 of max|grad|).  One compute_gradients
    per mode and shape (tests/grad_planes_worker.py; the switches are read once per process):
    * BACKWARD: every GRU variable's and the table's gradient on the default kernels against the fp32 kernels for the same
      products (weight gradients, input gradients; the forward and the read path identical in both runs, so nothing but the
      kernels under test differs), in units of the gradient's largest element: <= 1e-6 at H = 64 (measured 3e-7) -- and the two-plane
      arithmetic (HPMN_WGRAD_PLANES=2, HPMN_DX_PLANES=2) must be measurably further away, or the test looks at nothing;
    * FORWARD (H = 128): the memory slots with the three-plane projection against the fp32 projection kernel: <= 2e-5
      after 233 recurrent steps x 4 layers (measured 5e-6; the two-plane projection: 1e-4)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    modes = {"p3": {}, "p2_bwd": {"HPMN_WGRAD_PLANES": "2", "HPMN_DX_PLANES": "2"},
             "fp32_bwd": {"HPMN_WGRAD_BF16": "0", "HPMN_BWD_DX_INLOOP": "0", "HPMN_DX_BF16": "0"},
             "p2_proj": {"HPMN_PROJ_PLANES": "2"}, "fp32_proj": {"HPMN_PROJ_BF16": "0"}}
    for tag, extra in modes.items():
        env = dict(os.environ)
        env.update(extra)
        env["HPMN_DET_SCATTER"] = "1"
        dst = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "grad_planes_worker.py"), dst], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[tag] = dict(np.load(dst))
    worst = {"p3": {}, "p2_bwd": {}}
    for k, ref in res["fp32_bwd"].items():
        if not ("GRU" in k or "emb_mtx" in k):
            continue
        scale = float(np.abs(ref).max())
        assert scale > 0 and np.isfinite(ref).all(), k
        for tag in worst:
            e = float(np.abs(res[tag][k] - ref).max()) / scale
            cfg = k.split("/")[0]
            worst[tag][cfg] = max(worst[tag].get(cfg, 0.0), e)
    fwd = {tag: float(np.abs(res[tag]["h128/memory"] - res["fp32_proj"]["h128/memory"]).max()) for tag in ("p3", "p2_proj")}
    print("max |grad - fp32 kernels' grad| / max|grad|:", worst, " H = 128 max |memory - fp32 projection's|:", fwd)
    # (H = 128: every layer's input gradient is one of the kernels under test and feeds the reverse scan of the layer below, so
    #  a kernel's 3e-7 arrives at layer 0 amplified by the recurrences in between -- measured 2.4e-6 with three planes, 1.4e-5
    #  with two; H = 64: only layer 0's is, 3.0e-7 / 9.7e-7)
    bar = {"h64": 1e-6, "h128": 5e-6}
    for cfg in ("h64", "h128"):
        np.testing.assert_array_equal(res["p3"][cfg + "/memory"], res["fp32_bwd"][cfg + "/memory"])     # same forward
        assert worst["p3"][cfg] <= bar[cfg], (cfg, worst)
        assert worst["p2_bwd"][cfg] >= 2.0 * worst["p3"][cfg], (cfg, worst)
    assert fwd["p3"] <= 2e-5 and fwd["p2_proj"] >= 2.0 * fwd["p3"], fwd       # (measured 5.3e-6 / 9.9e-5 after 233 recurrent steps x 4 layers)


def test_read_training_launch_on_bf16_fragments_tracks_the_fp32_launch(tmp_path):
    """r5: the read path's TRAINING launch runs its dense layers on v_mfma_f32_16x16x32_bf16 with split
    operands -- three bf16 planes per fp32 operand and the six products of order <= 2, forward and input-gradient products alike
    (fp32-equivalent: what is dropped is below 2^-24 of a product) -- from per-step weight fragment images (csrc/read_path.hip: read_wimg_kernel, dense_bf); the r4
    launch (fp32 matrix instructions, HPMN_READ_BF16=0) is the same arithmetic in fp32.  One compute_gradients each, same seeded
    weights and batch (two processes: the switch is read once): predictions to 4e-6, loss to 2e-6 relative, every dense
    variable's gradient to 5e-6 of the gradient's largest element (measured: 6.3e-7 / 2.2e-7 / 6.8e-7; with TWO planes in the
    input-gradient products the gradients were at 8.5e-6 and test_three_training_steps_track_the_restatement lost an element
    to Adam's normalisation -- three planes cost nothing measurable), at the XLong slot count
    (14 of 16 tile rows, an odd batch) and at the Amazon one (H = 32)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, val in (("bf16", "1"), ("fp32", "0")):
        env = dict(os.environ)
        env["HPMN_READ_BF16"] = val
        env["HPMN_DET_SCATTER"] = "1"
        dst = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "read_bf_worker.py"), dst], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(np.load(dst))
    a, b = outs
    for tag in ("xlong", "small"):
        np.testing.assert_allclose(a[tag + "_pred"], b[tag + "_pred"], rtol=0, atol=4e-6, err_msg=tag)
        np.testing.assert_allclose(a[tag + "_ce"], b[tag + "_ce"], rtol=2e-6, atol=0, err_msg=tag)
        ga, gb = a[tag + "_grad"], b[tag + "_grad"]
        assert np.isfinite(ga).all() and float(np.abs(gb).max()) > 0
        assert float(np.abs(ga - gb).max()) <= 5e-6 * float(np.abs(gb).max()), (tag, float(np.abs(ga - gb).max()), float(np.abs(gb).max()))
        np.testing.assert_allclose(a[tag + "_table_grad_abs"], b[tag + "_table_grad_abs"], rtol=5e-6)


@pytest.mark.parametrize("xp_rows", [False, True])
@pytest.mark.parametrize("T,K,B", [(105, 4, 37), (1001, 7, 21)])
def test_tile_kernel_h128_matches_the_oracle(dev, tmp_path, monkeypatch, T, K, B, xp_rows):
    """r5 (VERDICT r4 missing #4): evaluation at H = 128 on the matrix cores -- hpmn_tile_fwd, 16-sequence tiles, four waves
    each holding a quarter of the units of all three gates, split-f16 operands; layer 0 projects in the kernel, the layers above
    read hpmn_gru_input_proj's rows.  ops.tiled_forward_inference against the float64 oracle at 1e-4 (memory; logit / prediction
    through the read path), a partial last tile, the Industry zero prefix, configs[4]'s length; and against the per-sequence kernels."""
    from hpmn_amd import ops
    # (xp_rows: the upper layers on rows hpmn_gru_input_proj projected -- the kernel's mode 1; default: projected in the kernel,
    #  mode 2 -- layer 0 is mode 0 in both)
    monkeypatch.setattr(ops, "TILE128_XP", xp_rows)
    cfg = cfg_industry(H=128, K=K, T=T, V=900)
    p = f32_params(cfg, 501)
    ids, label = rand_ids(cfg, B, 502)
    want = O.forward(cfg, p, ids, label)
    m = make_model(cfg, tmp_path, p)
    t = torch.as_tensor(ids).to(dev)
    assert ops.tile_kernel_supported(128, m.spec.D0)
    mem, last = ops.tiled_forward_inference(m.spec, t, m.params["Embedding/emb_mtx"], m._gru_weights())
    np.testing.assert_allclose(mem.cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    out = ops.read_fwd(m._read_desc, m._read_params, mem, last, True, True)
    for k in ("logit", "prediction"):
        np.testing.assert_allclose(out[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)
    ref_mem, ref_last = ops.scan_forward_inference(m.spec, t, m.params["Embedding/emb_mtx"], m._gru_weights())
    assert torch.equal(last, ref_last)
    np.testing.assert_allclose(mem.cpu().numpy(), ref_mem.cpu().numpy(), rtol=0, atol=5e-5)
    # an evaluation-sized pass takes the tile path by itself and agrees with the small one row for row
    big = torch.as_tensor(np.concatenate([ids] * (1600 // B + 1))[:1600]).to(dev)
    assert m._tiled_inference(1600)
    got = m.forward_inference(big)
    assert torch.equal(got["memory"][:B], mem)


@pytest.mark.parametrize("name,cfg,B", [("xlong", cfg_industry(H=64, K=7, T=1001, V=900), 37),
                                        ("taobao", O.HpmnConfig(700, 4, 300, 64, 16, 3, (2, 2, 3, 5, 5, 1), 5, False, 1e-5), 21),
                                        ("amazon-like-d48", cfg_amazon(H=64, K=3, T=100, F=3, V=300), 5)])
def test_four_wave_tile_kernel_h64_matches_the_oracle_and_the_first_generation(dev, tmp_path, monkeypatch, name, cfg, B):
    """r5: the second-generation H = 64 evaluation kernel (gru_tile64.hip through hpmn_tile_fwd: four waves per tile, each with
    a quarter of the units of all three gates, one layer per launch) against the float64 oracle at 1e-4 -- XLong's length, odd
    periods (3, 5), D = 32 / 48 / 64, the id-0 mask, partial tiles -- and against the twelve-wave kernel it replaces in
    ops.tiled_forward_inference (same split-f16 arithmetic, different summation trees: 2e-5)."""
    from hpmn_amd import ops
    p = f32_params(cfg, 601)
    ids, label = rand_ids(cfg, B, 602)
    want = O.forward(cfg, p, ids, label)
    m = make_model(cfg, tmp_path, p)
    t = torch.as_tensor(ids).to(dev)
    emb, w = m.params["Embedding/emb_mtx"], m._gru_weights()
    assert ops.TILE64
    monkeypatch.setattr(ops, "_cu_count", lambda dev: 0)            # (the four-wave kernel whatever the tile count)
    mem, last = ops.tiled_forward_inference(m.spec, t, emb, w)
    np.testing.assert_allclose(mem.cpu().numpy(), want["memory"], rtol=0, atol=TOL)
    out = ops.read_fwd(m._read_desc, m._read_params, mem, last, True, True)
    for k in ("logit", "prediction"):
        np.testing.assert_allclose(out[k].cpu().numpy(), want[k], rtol=0, atol=TOL, err_msg=k)
    monkeypatch.setattr(ops, "TILE64", False)
    mem1, last1 = ops.tiled_forward_inference(m.spec, t, emb, w)
    assert torch.equal(last, last1)
    np.testing.assert_allclose(mem.cpu().numpy(), mem1.cpu().numpy(), rtol=0, atol=2e-5)
