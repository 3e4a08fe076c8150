"""Stratified initialisation / update over the model's discrete latent (reference src/initialize.jl:64-109,
src/update.jl:165-210, stratified_map! src/utils.jl:29-55; tests test/initialize.jl:39-64, test/update.jl:13-40):
oracle against the reference's invariants and closed-form weights on CPU, HIP against the oracle bit for bit on the GPU."""
import math

import numpy as np
import pytest

HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


def logpdf_normal(x, mu, sd):
    return -0.5 * ((x - mu) / sd) ** 2 - math.log(sd) - HALF_LOG_2PI


def setup(g):
    m = g.models.object_motion()
    ys = g.models.simulate(m, 4)
    return m, ys


def blocks(n, K, layout):
    """index sets of stratified_map! (utils.jl:36-43) and the remainder (:45-52)"""
    B = n // K
    idx = [np.arange(k * B, (k + 1) * B) if layout == "contiguous" else np.arange(k, K * B, K) for k in range(K)]
    return idx, np.arange(K * B, n)


@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
@pytest.mark.parametrize("n", [100, 101])
def test_oracle_initialize_with_stratification(g, o, layout, n):
    """test/initialize.jl:39-64: every block carries its stratum's value; weights = log p(stratum choice) + log K + log p(obs | x)"""
    m, ys = setup(g)
    strata = [0.0, 1.0]
    f = o.OracleFilter(m.model_id, m.params, n, 5).initialize(ys[0], strata=strata, layout=layout)
    idx, rem = blocks(n, 2, layout)
    for k, val in enumerate(strata):
        assert np.all(f.rows[idx[k], 0] == val)
    assert np.all(np.isin(f.rows[rem, 0], strata)) and rem.size == n % 2
    p_start, sobs = m.info["p_start"], m.info["sobs"]
    for i in range(n):
        mv, y = f.rows[i]
        expect = math.log(p_start if mv else 1 - p_start) + math.log(2) + logpdf_normal(ys[0][0], y, sobs)
        assert abs(f.lw[i] - expect) < 1e-9
    # a single stratum = plain constrained generation: log K = 0
    f1 = o.OracleFilter(m.model_id, m.params, n, 5).initialize(ys[0], strata=[1.0], layout=layout)
    assert np.all(f1.rows[:, 0] == 1.0)


@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_oracle_update_with_stratification(g, o, layout):
    """test/update.jl:13-40: blocks carry the stratum's value; increments = logpdf(bernoulli, val, p) + log(2) + logpdf(normal, obs, y, std)"""
    m, ys = setup(g)
    n = 100
    f = o.OracleFilter(m.model_id, m.params, n, 9).initialize(ys[0])
    prev_rows, prev_lw = f.rows.copy(), f.lw.copy()
    f.update(ys[1], strata=[0.0, 1.0], layout=layout)
    idx, _ = blocks(n, 2, layout)
    p_stay, p_start, sobs = m.info["p_stay"], m.info["p_start"], m.info["sobs"]
    for k, val in enumerate([0.0, 1.0]):
        assert np.all(f.rows[idx[k], 0] == val)
        for i in idx[k]:
            p = p_stay if prev_rows[i, 0] else p_start
            inc = math.log(p if val else 1 - p) + math.log(2) + logpdf_normal(ys[1][0], f.rows[i, 1], sobs)
            assert abs(f.lw[i] - (prev_lw[i] + inc)) < 1e-9


def test_oracle_stratified_estimate_is_unbiased(g, o):
    """the log K correction makes the stratified filter estimate the same marginal likelihood as the plain one"""
    m, ys = setup(g)
    lml_s, lml_p = [], []
    for seed in range(40):
        a = o.OracleFilter(m.model_id, m.params, 2000, seed).initialize(ys[0], strata=[0.0, 1.0])
        b = o.OracleFilter(m.model_id, m.params, 2000, 1000 + seed).initialize(ys[0])
        a.update(ys[1], strata=[0.0, 1.0]); b.update(ys[1])
        lml_s.append(a.log_ml_estimate()); lml_p.append(b.log_ml_estimate())
    ms, mp = np.log(np.mean(np.exp(lml_s))), np.log(np.mean(np.exp(lml_p)))
    assert abs(ms - mp) < 0.05
    assert np.std(lml_s) < np.std(lml_p)                      # and it does so with less variance


def test_choiceproduct(g):
    """src/utils.jl:57-98"""
    assert g.choiceproduct(("moving", [False, True])) == [{"moving": False}, {"moving": True}]
    assert len(g.choiceproduct(("a", [1, 2]), ("b", [3]))) == 2
    assert g.choiceproduct({"a": [1, 2]}) == [{"a": 1}, {"a": 2}]


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
@pytest.mark.parametrize("n", [100, 101, 50_001])
@pytest.mark.parametrize("keep_prev", [False, True])
def test_hip_strata_bitexact(g, o, layout, n, keep_prev):
    m, ys = setup(g)
    strata = g.choiceproduct(("moving", [False, True]))
    st = g.pf_initialize(m, (0,), ys[0], strata, n, seed=21, keep_prev=keep_prev, layout=layout)
    orc = o.OracleFilter(m.model_id, m.params, n, 21, keep_prev=keep_prev).initialize(ys[0], strata=[0.0, 1.0], layout=layout)
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    g.pf_update(st, (1,), (None,), ys[1], strata, layout=layout); orc.update(ys[1], strata=[0.0, 1.0], layout=layout)
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
    g.pf_update(st, (2,), (None,), ys[2], [1.0, 0.0, 1.0]); orc.update(ys[2], strata=[1.0, 0.0, 1.0])   # default :interleaved, K = 3
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    if keep_prev:
        g.pf_rejuvenate(st, g.mh, (), 1); orc.rejuvenate("move", 1)
        assert np.array_equal(st.traces, orc.rows)
    # sub-state: strata inside the view (update.jl:193-210 takes a ParticleFilterView)
    g.pf_update(st[10:60], (3,), (None,), ys[3], strata, layout="contiguous"); orc[10:60].update(ys[3], strata=[0.0, 1.0], layout="contiguous")
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)


@pytest.mark.gpu
def test_hip_strata_errors(g, o):
    m = g.models.lgssm2(); ys = g.models.simulate(m, 2)
    with pytest.raises(g.ErrorException):
        g.pf_initialize(m, (0,), ys[0], [0.0, 1.0], 100)                     # no discrete latent
    mo, yo = setup(g)
    with pytest.raises(g.ErrorException):
        g.pf_initialize(mo, (0,), yo[0], [{"y": 0.0}], 100)                  # not the stratifiable address
