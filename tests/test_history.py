"""Trajectory store: statistics of PAST choices along the surviving ancestry, i.e. what
mean(state, 5 => :moving) does on Gen's persistent traces (reference README.md:97-107, src/statistics.jl:13-14)."""
import numpy as np
import pytest


def readme_filter(make, N, seed, ys, ess_thresh=0.5):
    """README.md:60-79: resample (:residual) + mh rejuvenation when ESS < N/2, then extend to t."""
    f = make(seed)
    for t in range(1, len(ys)):
        if f["ess"]() < ess_thresh * N:
            f["resample"]()
            f["rejuvenate"]()
        f["update"](ys[t])
    return f


def oracle_ops(g, o, model, ys, N):
    def make(seed):
        flt = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=True, history=True).initialize(ys[0])
        return dict(obj=flt, ess=flt.effective_sample_size, resample=lambda: flt.resample("residual"),
                    rejuvenate=lambda: flt.rejuvenate("move", 1), update=flt.update)
    return make


def exact_object_motion(model, ys):
    """Known answer for BASELINE config 1: enumerate the 2^T `moving` sequences of README.md:43-55 (y is a random walk
    with sigma_y = 0.01, integrated out as extra observation variance) -> exact smoothed P(moving_t | y_1:T), log p(y)."""
    import itertools, math
    T, yobs = ys.shape[0], ys[:, 0]
    p_stay, p_start, sy, sobs = model.params[0], model.params[1], model.info["sy"], model.info["sobs"]
    var = sobs ** 2 + np.arange(1, T + 1) * sy ** 2
    post, Z = np.zeros(T), 0.0
    for seq in itertools.product([0, 1], repeat=T):
        p, prev = 1.0, 0
        for m in seq:
            pm = p_stay if prev else p_start
            p *= pm if m else 1.0 - pm
            prev = m
        y = np.cumsum([m * ys[t, 1] for t, m in enumerate(seq)])          # ys[:,1] = sin(t)
        w = p * math.exp(np.sum(-0.5 * (yobs - y) ** 2 / var - 0.5 * np.log(2 * np.pi * var)))
        Z += w
        post += w * np.array(seq)
    return post / Z, math.log(Z)


def test_oracle_readme_example_vs_exact_posterior(g, o):
    """README.md:60-107 / BASELINE.md §1 (config 1: N = 100, T = 10, residual resample + mh when ESS < N/2): the
    statistics the README prints, mean(state, t => :moving), against the EXACT smoothed posterior of this data set
    (the README's own numbers, 0.07 / 0.95, belong to its unseeded random data set and cannot be reproduced)."""
    model = g.models.object_motion()
    ys = g.models.simulate(model, 10)
    exact, exact_lml = exact_object_motion(model, ys)
    assert exact[4] < 0.05 and exact[6] > 0.95                      # still at t = 5, certainly moving at t = 7
    N = 100
    est, lml = [], []
    for seed in range(1, 33):
        f = readme_filter(oracle_ops(g, o, model, ys, N), N, seed, ys)["obj"]
        est.append([f.history_mean(t, 0) for t in range(1, 11)]); lml.append(f.log_ml_estimate())
        assert np.array_equal(f.history_column(10, 1), f.rows[:, 1])   # the current step's history = the current columns
        p = f.history_mean(5, 0)
        assert abs(f.history_var(5, 0) - p * (1 - p)) < 1e-12          # variance of a 0/1 choice
    assert np.abs(np.mean(est, axis=0) - exact).max() < 0.08, (np.mean(est, axis=0), exact)
    assert abs(np.mean(lml) - exact_lml) < 0.1
    # and with many particles the filter converges to the exact answer
    N = 20000
    f = readme_filter(oracle_ops(g, o, model, ys, N), N, 5, ys)["obj"]
    assert np.abs(np.array([f.history_mean(t, 0) for t in range(1, 11)]) - exact).max() < 0.03
    assert abs(f.log_ml_estimate() - exact_lml) < 0.05


def test_oracle_history_is_ancestral(g, o):
    """every current particle's past value is the value its ancestor had at that step"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 6); N = 200
    f = o.OracleFilter(model.model_id, model.params, N, 3, history=True).initialize(ys[0])
    snaps, anc_chain = [f.rows[:, 0].copy()], []
    for t in range(1, 6):
        f.resample("multinomial"); anc_chain.append(f.parents - 1)
        f.update(ys[t]); snaps.append(f.rows[:, 0].copy())
    for step in range(1, 7):
        idx = np.arange(N)
        for a in reversed(anc_chain[step - 1:]):
            idx = a[idx]
        assert np.array_equal(f.history_column(step, 0), snaps[step - 1][idx])


@pytest.mark.gpu
@pytest.mark.parametrize("name,N", [("object_motion", 100), ("object_motion", 5000), ("lgssm2", 20000)])
def test_hip_history_matches_oracle(g, o, name, N):
    model = g.models.by_name(name)
    ys = g.models.simulate(model, 10)
    orc = o.OracleFilter(model.model_id, model.params, N, 7, keep_prev=True, history=True).initialize(ys[0])
    st = g.pf_initialize(model, (1,), ys[0], N, seed=7, keep_prev=True, history=16)
    for t in range(1, 10):
        ess = g.get_ess(st)
        assert ess == orc.effective_sample_size()
        if ess < 0.5 * N or t % 4 == 0:
            g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
            if t % 2 == 0:                               # two resamples inside one step compose
                g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
            g.pf_rejuvenate(st, g.mh, (), 1); orc.rejuvenate("move", 1)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        for step in (1, max(1, t - 1), t + 1):
            for c in range(model.dim):
                assert np.array_equal(st.history_column(step, c), orc.history_column(step, c)), (t, step, c)
    for step in (5, 6):
        assert g.mean(st, (step, 0)) == orc.history_mean(step, 0) and g.var(st, (step, 0)) == orc.history_var(step, 0)
    with pytest.raises(g.ErrorException):
        st.history_column(11, 0)
    with pytest.raises(g.ErrorException):
        g.pf_resize(st, 50)                              # resizing with a trajectory store is refused


@pytest.mark.gpu
def test_hip_readme_example(g, o):
    """BASELINE.json configs[0] on the device: object_motion, T = 10, N = 100, residual resample + mh rejuvenate, the
    README's driver loop verbatim; smoothed P(moving_t) against the exact posterior and against the oracle bit for bit."""
    model = g.models.object_motion()
    ys = g.models.simulate(model, 10)
    exact, exact_lml = exact_object_motion(model, ys)
    N, est = 100, []
    for seed in range(1, 17):
        st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=True, history=10)
        orc = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=True, history=True).initialize(ys[0])
        for t in range(1, 10):
            if g.effective_sample_size(st) < 0.5 * N:                       # README.md:68
                g.pf_resample(st, "residual"); orc.resample("residual")
                g.pf_rejuvenate(st, g.mh, ()); orc.rejuvenate("move", 1)
            g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        est.append([g.mean(st, (t, 0)) for t in range(1, 11)])
        pm = g.proportionmap(st, (6, 0))                                  # statistics.jl:91-101 on a past 0/1 choice
        assert set(pm) <= {0.0, 1.0} and abs(sum(pm.values()) - 1.0) < 1e-12 and abs(pm.get(1.0, 0.0) - est[-1][5]) < 1e-12
        assert abs(sum(g.proportionmap(st, 0).values()) - 1.0) < 1e-12
        assert est[-1] == [orc.history_mean(t, 0) for t in range(1, 11)]
        assert g.get_lml_est(st) == orc.log_ml_estimate()
    assert np.abs(np.mean(est, axis=0) - exact).max() < 0.1, (np.mean(est, axis=0), exact)
