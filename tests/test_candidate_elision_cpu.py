"""ABI v11, saved gates without the candidate (include/hpmn_hip.h, hpmn_amd/csrc/common.h: gru_coeff_from_states): the float32
arithmetic the reverse-scan feeders run, restated in NumPy float32 operation by operation, against the float64 coefficients
of BPTT through  h = u h_prev + (1 - u) c  (code/util.py:95-109) -- the error bound DESIGN_HISTORY.md 3.19 states, on CPU."""
import numpy as np


def _fma32(a, b, c):
    # one rounding, like v_fma_f32
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _coeff_from_states(h_new, h_prev, u):
    omu = (np.float32(1.0) - u).astype(np.float32)
    q = _fma32(-u, h_prev, h_new)
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(omu > 0, (q * (np.float32(1.0) / omu).astype(np.float32)).astype(np.float32), np.float32(0.0))
    k1 = _fma32(-q, c, omu)
    k2 = (u * _fma32(h_prev, omu, -q)).astype(np.float32)
    return k1, k2


def _cases(n, rng):
    # pre-activations spanning both saturations of the update gate, candidates up to |c| -> 1, states in (-1, 1)
    a_u = rng.uniform(-20.0, 20.0, size=n)
    a_u[: n // 8] = rng.uniform(14.0, 40.0, size=n // 8)            # u rounds to 1.0f exactly for a >~ 17
    a_u[n // 8: n // 4] = rng.uniform(-40.0, -14.0, size=n // 8)
    a_c = rng.uniform(-9.0, 9.0, size=n)
    h_prev = np.tanh(rng.uniform(-3.0, 3.0, size=n))
    return a_u, a_c, h_prev


def test_coefficients_from_the_saved_states_match_the_stored_candidate_form():
    rng = np.random.default_rng(19)
    n = 400_000
    a_u, a_c, h_prev64 = _cases(n, rng)
    u = (1.0 / (1.0 + np.exp(-a_u))).astype(np.float32)               # what the forward saves (float32)
    c = np.tanh(a_c).astype(np.float32)                               # what it no longer saves
    h_prev = h_prev64.astype(np.float32)
    # the forward's own arithmetic: h = fma(u, h_prev - c, c)
    h_new = _fma32(u, (h_prev - c).astype(np.float32), c)
    k1, k2 = _coeff_from_states(h_new, h_prev, u)
    u64, c64, hp64 = u.astype(np.float64), c.astype(np.float64), h_prev.astype(np.float64)
    k1_true = (1.0 - u64) * (1.0 - c64 * c64)
    k2_true = (hp64 - c64) * u64 * (1.0 - u64)
    assert np.isfinite(k1).all() and np.isfinite(k2).all()
    sat = u == np.float32(1.0)
    assert sat.sum() > 1000 and (u < 1e-6).sum() > 1000               # both saturations are in the sample
    assert (k1[sat] == 0).all() and np.abs(k2[sat]).max() <= 1.2e-7   # u == 1: the true coefficients are 0
    # ABSOLUTE errors (the coefficients are O(1)); the worst case is the corner u = 1 - 2^-24, where q = (1 - u) c is smaller
    # than the forward's own rounding of h: measured 3.2e-7 there, 2.3e-7 elsewhere (2 |c| times the two roundings of h)
    assert np.abs(k1 - k1_true).max() <= 5e-7
    assert np.abs(k2 - k2_true).max() <= 1.5e-7
    inner = (u <= np.float32(1.0) - np.float32(2.0 ** -20))
    assert np.abs(k1 - k1_true)[inner].max() <= 2.5e-7
    # ... which is the stored-candidate form's own float32 error class
    omu = (np.float32(1.0) - u).astype(np.float32)
    k1_old = (omu * (np.float32(1.0) - c * c).astype(np.float32)).astype(np.float32)
    k2_old = ((h_prev - c).astype(np.float32) * u * omu).astype(np.float32)
    assert np.abs(k1_old - k1_true).max() <= 1.5e-7 and np.abs(k2_old - k2_true).max() <= 1.5e-7


def test_a_whole_reverse_pass_is_unchanged_at_the_gradient_tolerance():
    """64 units, 300 steps of the recurrence's diagonal part: gradients through the recovered coefficients against float64."""
    rng = np.random.default_rng(23)
    T, H = 300, 64
    a_u, a_c, _ = _cases(T * H, rng)
    u = (1.0 / (1.0 + np.exp(-a_u.reshape(T, H) * 0.2))).astype(np.float32)
    c = np.tanh(a_c.reshape(T, H)).astype(np.float32)
    h = np.zeros((T + 1, H), np.float32)
    for t in range(T):
        h[t + 1] = _fma32(u[t], (h[t] - c[t]).astype(np.float32), c[t])
    dh = rng.normal(size=H).astype(np.float32)
    dh64 = dh.astype(np.float64)
    g_c = np.zeros((T, H), np.float32)
    g_c64 = np.zeros((T, H))
    for t in range(T - 1, -1, -1):
        k1, _ = _coeff_from_states(h[t + 1], h[t], u[t])
        g_c[t] = (dh * k1).astype(np.float32)                          # d loss / d (candidate pre-activation)
        dh = (dh * u[t]).astype(np.float32)
        u64, c64 = u[t].astype(np.float64), c[t].astype(np.float64)
        g_c64[t] = dh64 * (1.0 - u64) * (1.0 - c64 * c64)
        dh64 = dh64 * u64
    assert np.abs(g_c - g_c64).max() <= 2e-6 * np.abs(g_c64).max()     # (the suite's gradient bar is 2e-4 of the max)
