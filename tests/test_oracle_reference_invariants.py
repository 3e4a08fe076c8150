"""Pins the oracle's deterministic arithmetic to what the REFERENCE's own tests assert
(/root/reference/test/*.jl, v0.2.3; cited per test).  The reference has no golden vectors or seeds
(SURVEY.md §8c), so these invariants -- not random streams -- are what can be pinned."""
import math

import numpy as np
import pytest

METHODS = ["multinomial", "residual", "stratified"]


def lgssm(g, o, N=100, seed=3, T=4, keep_prev=False):
    m = g.models.lgssm2()
    ys = g.models.simulate(m, T)
    f = o.OracleFilter(m.model_id, m.params, N, seed, keep_prev=keep_prev).initialize(ys[0])
    return m, ys, f


def lse(v):
    m = np.max(v)
    return m + math.log(np.sum(np.exp(v - m)))


def test_initialize_weights_are_analytic(g, o):
    """test/initialize.jl:6-10, test/update.jl:5-10: weights equal the analytic logpdf of the observation."""
    m, ys, f = lgssm(g, o)
    sr = m.info["sr"]
    want = sum(-0.5 * ((ys[0][k] - f.rows[:, k]) / sr) ** 2 - math.log(sr) - 0.5 * math.log(2 * math.pi) for k in range(2))
    np.testing.assert_allclose(f.lw, want, rtol=1e-12, atol=1e-12)
    old = f.lw.copy(); xp = f.rows.copy()
    f.update(ys[1])
    want = sum(-0.5 * ((ys[1][k] - f.rows[:, k]) / sr) ** 2 - math.log(sr) - 0.5 * math.log(2 * math.pi) for k in range(2))
    np.testing.assert_allclose(f.lw - old, want, rtol=1e-10, atol=1e-12)
    # the transition really is x' = A x + sq z: residual of the mean has the right scale
    resid = f.rows[:, :2] - xp[:, :2] @ m.info["A"].T
    assert abs(resid.std() - m.info["sq"]) < 0.03


def test_utils_identities(g, o):
    """test/utils.jl:6-10: sum(exp(lognorm)) == 1, sum(norm weights) == 1, ESS == (sum w)^2 / sum w^2."""
    m, ys, f = lgssm(g, o)
    assert abs(np.exp(f.log_norm_weights()).sum() - 1.0) < 1e-12
    w = f.norm_weights()
    assert abs(w.sum() - 1.0) < 1e-12
    assert abs(f.effective_sample_size() - w.sum() ** 2 / (w ** 2).sum()) < 1e-9 * 100
    # against the literal Float64 restatement of utils.jl
    assert abs(f.effective_sample_size() - o.lib().lit_ess(f.lw, f.n)) < 1e-9 * 100
    assert abs(f.summary().lse - o.lib().lit_logsumexp(f.lw, f.n)) < 1e-12 * 10


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_resample_invariants(g, o, method, alpha):
    """test/resample.jl:5-23,43-70,90-119: new_traces == old_traces[parents]; log-ML unchanged by resampling;
    residual copies >= floor(N w); stratified max-weight particle gets >= floor(N w_max) copies."""
    m, ys, f = lgssm(g, o, N=100)
    old_rows, lw = f.rows.copy(), f.lw.copy()
    old_lml = lse(lw) - math.log(100)
    lp = lw if alpha is None else alpha * lw
    w = np.exp(lp - lse(lp))
    f.resample(method, priority_alpha=alpha)
    par = f.parents - 1
    assert np.array_equal(f.rows, old_rows[par])
    assert abs(f.log_ml_estimate() - old_lml) < 1e-9
    copies = np.bincount(par, minlength=100)
    if method == "residual":
        assert np.all(copies >= np.floor(100 * w - 1e-9).astype(int))
    if method == "stratified":
        assert copies[np.argmax(w)] >= math.floor(100 * w.max() - 1e-9)
    if alpha is None:
        assert np.all(f.lw == 0.0)                                    # resample.jl:195
    else:
        assert abs(lse(f.lw) - math.log(100)) < 1e-9                   # resample.jl:192,200


@pytest.mark.parametrize("method", ["residual", "stratified"])
def test_uniform_weights_are_identity(g, o, method):
    """test/resample.jl:36-40,83-87"""
    m, ys, f = lgssm(g, o, N=100)
    f.lw[:] = 0.0
    rows = f.rows.copy()
    f.resample(method)
    assert np.array_equal(f.parents, np.arange(1, 101)) and np.array_equal(f.rows, rows)


@pytest.mark.parametrize("method", METHODS)
def test_invalid_weights(g, o, method):
    """test/resample.jl:26-31,73-78,122-127: all -Inf: check=true throws, check=false ends with all-zero log-weights."""
    m, ys, f = lgssm(g, o, N=100)
    f.lw[:] = -np.inf
    with pytest.raises(o.OracleError):
        f.resample(method, check=True)
    assert f.resample(method, check=False) is True
    assert np.all(f.lw == 0.0)
    f.lw[:] = np.nan
    with pytest.raises(o.OracleError):
        f.resample(method, check="warn")
    with pytest.raises(o.OracleError):
        f.resample("systematic")                                      # resample.jl:28


def test_move_reweight_is_likelihood_ratio(g, o):
    """test/rejuvenate.jl:3-28,52-71: new log-weight = old + log p(y|x*) - log p(y|x); move-accept keeps weights."""
    m, ys, f = lgssm(g, o, N=200, keep_prev=True)
    f.update(ys[1])
    sr = m.info["sr"]
    ll = lambda rows: sum(-0.5 * ((ys[1][k] - rows[:, k]) / sr) ** 2 for k in range(2))
    lw0, r0 = f.lw.copy(), f.rows.copy()
    f.rejuvenate("reweight", 1)
    np.testing.assert_allclose(f.lw - lw0, ll(f.rows) - ll(r0), rtol=1e-9, atol=1e-10)
    assert np.array_equal(f.rows[:, 2:4], r0[:, 2:4])                 # x_{t-1} untouched
    lw1, r1 = f.lw.copy(), f.rows.copy()
    f.rejuvenate("move", 3)
    assert np.array_equal(f.lw, lw1)                                  # rejuvenate.jl:40-53: weights unchanged
    moved = np.any(f.rows[:, :2] != r1[:, :2], axis=1)
    assert 0 < moved.sum() < 200 and f.n_accepted >= moved.sum()
    with pytest.raises(o.OracleError):
        f.rejuvenate("gibbs")                                         # rejuvenate.jl:25


def test_statistics_degenerate_and_weighted(g, o):
    """test/statistics.jl:10-18 (degenerate values: mean exact, var ~ 0) + definition of weighted mean/var."""
    m, ys, f = lgssm(g, o, N=500)
    f.rows[:, 0] = 1.0
    assert abs(f.mean(0) - 1.0) < 1e-12 and abs(f.var(0)) < 1e-6
    w = f.norm_weights()
    mu = (w * f.rows[:, 1]).sum()
    assert abs(f.mean(1) - mu) < 1e-12
    assert abs(f.var(1) - (w * (f.rows[:, 1] - mu) ** 2).sum()) < 1e-12


def test_log_ml_matches_kalman(g, o):
    """known answer: exact Kalman log-likelihood of the LG-SSM; particle estimate within Monte-Carlo error."""
    m = g.models.lgssm2()
    ys = g.models.simulate(m, 30)
    exact = g.models.kalman_loglik(m, ys)
    est = []
    for seed in range(1, 5):
        f = o.OracleFilter(m.model_id, m.params, 20000, seed).initialize(ys[0])
        for t in range(1, 30):
            f.resample("stratified", sort_particles=False, check=False)
            f.update(ys[t])
        est.append(f.log_ml_estimate())
    assert abs(np.mean(est) - exact) < 0.08, (est, exact)


def test_object_motion_readme_example(g, o):
    """README.md:60-107 (BASELINE config 1): N=100, T=10, residual resampling + MH when ESS < N/2; the filter
    must infer still for t<=5 and moving for t>=6 (README.md:97-107: 0.07 at t=5, 0.95 at t=6)."""
    m = g.models.object_motion()
    ys = g.models.simulate(m, 10)
    p_t = np.zeros((8, 10))
    for seed in range(8):
        f = o.OracleFilter(m.model_id, m.params, 100, seed + 1, keep_prev=True).initialize(ys[0])
        p_t[seed, 0] = f.mean(0)
        for t in range(1, 10):
            if f.effective_sample_size() < 0.5 * 100:                 # README.md:68
                f.resample("residual")
                f.rejuvenate("move", 1)
            f.update(ys[t])
            p_t[seed, t] = f.mean(0)                                  # filtering estimate of moving_t
    p = p_t.mean(axis=0)
    assert p[:5].max() < 0.35 and p[6:].min() > 0.65, p
