"""Custom-proposal initialize / update (reference src/initialize.jl:46-62, src/update.jl:79-96,
src/translate.jl:86-105, test/update.jl:47-80): native locally optimal proposal of the linear-Gaussian model."""
import math

import numpy as np
import pytest


def test_oracle_proposal_weights_are_analytic(g, o):
    """weight = model_score_diff - fwd_proposal_score; for the exact conditional it equals log p(y_t | x_{t-1}),
    independent of the sampled x (the reference's tests check such analytic weights, test/update.jl:57-63)."""
    m = g.models.lgssm2(); ys = g.models.simulate(m, 3); N = 400
    A, sq, sr, s0 = m.info["A"], m.info["sq"], m.info["sr"], m.info["s0"]
    f = o.OracleFilter(m.model_id, m.params, N, 5).initialize(ys[0], proposal=True)
    want = sum(-0.5 * ys[0][k] ** 2 / (s0 ** 2 + sr ** 2) - 0.5 * math.log(2 * math.pi * (s0 ** 2 + sr ** 2)) for k in range(2))
    np.testing.assert_allclose(f.lw, want, rtol=1e-11, atol=1e-11)
    # the proposed x really follows the conditional: mean = gain * y, sd = sv
    gain = s0 ** 2 / (s0 ** 2 + sr ** 2)
    assert np.abs(f.rows[:, :2].mean(axis=0) - gain * ys[0]).max() < 0.1
    xp, lw0 = f.rows.copy(), f.lw.copy()
    f.update(ys[1], proposal=True)
    mu = xp[:, :2] @ A.T
    want = sum(-0.5 * (ys[1][k] - mu[:, k]) ** 2 / (sq ** 2 + sr ** 2) - 0.5 * math.log(2 * math.pi * (sq ** 2 + sr ** 2)) for k in range(2))
    np.testing.assert_allclose(f.lw - lw0, want, rtol=1e-9, atol=1e-9)


def test_oracle_proposal_filter_matches_kalman_with_higher_ess(g, o):
    m = g.models.lgssm2(); ys = g.models.simulate(m, 30); N = 4000
    exact = g.models.kalman_loglik(m, ys)
    ess = {}
    for prop in (False, True):
        f = o.OracleFilter(m.model_id, m.params, N, 9).initialize(ys[0], proposal=prop)
        tot = 0.0
        for t in range(1, 30):
            f.resample("stratified", sort_particles=False, check=False)
            f.update(ys[t], proposal=prop)
            tot += f.effective_sample_size()
        ess[prop] = tot / 29
        assert abs(f.log_ml_estimate() - exact) < (0.15 if prop else 0.5), (prop, f.log_ml_estimate(), exact)   # Monte-Carlo error
    assert ess[True] > ess[False]                  # the optimal proposal wastes fewer particles


@pytest.mark.gpu
@pytest.mark.parametrize("keep_prev", [False, True])
def test_hip_proposal_bitexact(g, o, keep_prev):
    m = g.models.lgssm2(); ys = g.models.simulate(m, 6); N = 20_000
    st = g.pf_initialize(m, (1,), ys[0], g.locally_optimal, (), N, seed=3, keep_prev=keep_prev)
    orc = o.OracleFilter(m.model_id, m.params, N, 3, keep_prev=keep_prev).initialize(ys[0], proposal=True)
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    for t in range(1, 6):
        if t % 2:
            g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
        g.pf_update(st, (t + 1,), (None,), ys[t], g.locally_optimal, ()); orc.update(ys[t], proposal=True)
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    with pytest.raises(g.ErrorException):
        g.pf_update(g.pf_initialize(g.models.sv1(), (1,), [0.1], 64), (2,), (None,), [0.1], g.locally_optimal, ())
