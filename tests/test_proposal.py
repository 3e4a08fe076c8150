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


def test_oracle_move_reweight_with_the_locally_optimal_proposal(g, o):
    """move_reweight(trace, proposal, proposal_args) (src/rejuvenate.jl:134-148): rel_weight = weight - fwd_score + bwd_score.  With
    q = p(x_t | x_{t-1}, y_t) the three terms cancel analytically -- a Gibbs move: every particle gets a fresh x_t from the exact
    conditional, the log-weights move by rounding error only, and the filter's estimates stay where they were."""
    m = g.models.lgssm2(); ys = g.models.simulate(m, 4); N = 2000
    f = o.OracleFilter(m.model_id, m.params, N, 5, keep_prev=True).initialize(ys[0])
    f.update(ys[1]); f.resample("multinomial", check=False); f.update(ys[2])
    rows0, lw0, lml0 = f.rows.copy(), f.lw.copy(), f.log_ml_estimate()
    f.rejuvenate("reweight", 2, proposal=())
    assert np.all(f.rows[:, :2] != rows0[:, :2]) and np.array_equal(f.rows[:, 2:4], rows0[:, 2:4])     # x_t moved, x_{t-1} kept
    assert np.abs(f.lw - lw0).max() < 1e-9 and abs(f.log_ml_estimate() - lml0) < 1e-9
    # the moved x_t follows the conditional: mean = mu + gain (y - mu)
    A, sq, sr = m.info["A"], m.info["sq"], m.info["sr"]
    mu = rows0[:, 2:4] @ A.T
    gain = sq ** 2 / (sq ** 2 + sr ** 2)
    assert np.abs((f.rows[:, :2] - (mu + gain * (ys[2] - mu))).mean(axis=0)).max() < 0.01


@pytest.mark.gpu
def test_hip_move_reweight_proposal_bitexact(g, o):
    m = g.models.lgssm2(); ys = g.models.simulate(m, 5); N = 30_000
    st = g.pf_initialize(m, (1,), ys[0], N, seed=3, keep_prev=True)
    orc = o.OracleFilter(m.model_id, m.params, N, 3, keep_prev=True).initialize(ys[0])
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
        # right after a resample: the pending gather rides on the move kernel
        g.pf_rejuvenate(st, g.move_reweight, (g.locally_optimal_move, ()), t, method="reweight"); orc.rejuvenate("reweight", t, proposal=())
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    with pytest.raises(g.ErrorException):                                                 # a model without a native move proposal
        s2 = g.pf_initialize(g.models.sv1(), (1,), [0.1], 64, keep_prev=True)
        g.pf_rejuvenate(s2, g.move_reweight, (g.locally_optimal_move, ()), 1, method="reweight")
    with pytest.raises(g.ErrorException):                                                 # the wrong proposal for the model
        g.pf_rejuvenate(st, g.move_reweight, (g.outlier_propose(0.9), (1,)), 1, method="reweight")


def test_oracle_mh_with_the_locally_optimal_proposal(g, o):
    """pf_move_accept!(state, mh, (proposal, proposal_args)) -- Gen.mh(trace, proposal, proposal_args) through src/rejuvenate.jl:40-53.  With
    q = p(x_t | x_{t-1}, y_t) alpha = weight - fwd_score + bwd_score is 0 up to rounding: an independence sampler from the exact conditional
    that accepts whenever log(rand()) < alpha ~ +-1e-15 -- every proposal whose alpha rounds to >= 0 plus almost all others (log u < -1e-15
    nearly surely): the acceptance count is ~ N per sweep, the weights never change."""
    m = g.models.lgssm2(); ys = g.models.simulate(m, 4); N = 4000
    f = o.OracleFilter(m.model_id, m.params, N, 5, keep_prev=True).initialize(ys[0])
    f.update(ys[1]); f.update(ys[2])
    rows0, lw0 = f.rows.copy(), f.lw.copy()
    f.rejuvenate("move", 2, proposal=())
    assert np.array_equal(f.lw, lw0) and np.array_equal(f.rows[:, 2:4], rows0[:, 2:4])
    assert f.n_accepted >= 2 * N - 5
    A, sq, sr = m.info["A"], m.info["sq"], m.info["sr"]
    mu = rows0[:, 2:4] @ A.T
    gain = sq ** 2 / (sq ** 2 + sr ** 2)
    assert np.abs((f.rows[:, :2] - (mu + gain * (ys[2] - mu))).mean(axis=0)).max() < 0.01


@pytest.mark.gpu
def test_hip_mh_proposal_bitexact(g, o):
    """k_move<..., REWEIGHT = false, PROP = true> against the oracle: states, weights, acceptance counts; with a pending resample gather
    riding on the move; models without a native move proposal refuse"""
    m = g.models.lgssm2(); ys = g.models.simulate(m, 5); N = 30_001
    st = g.pf_initialize(m, (1,), ys[0], N, seed=3, keep_prev=True)
    orc = o.OracleFilter(m.model_id, m.params, N, 3, keep_prev=True).initialize(ys[0])
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        if t != 2:
            g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
        g.pf_rejuvenate(st, g.mh, (g.locally_optimal_move, ()), t, method="move", count=(t == 3)); orc.rejuvenate("move", t, proposal=())
        if t == 3:
            assert st.n_accepted == orc.n_accepted
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    with pytest.raises(g.ErrorException):
        s2 = g.pf_initialize(g.models.sv1(), (1,), [0.1], 64, keep_prev=True)
        g.pf_rejuvenate(s2, g.mh, (g.locally_optimal_move, ()), 1, method="move")
