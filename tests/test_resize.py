"""Resize family (reference src/resize.jl, test/resize.jl): oracle against the reference's own
invariants on CPU, HIP against the oracle bit for bit on the GPU."""
import math

import numpy as np
import pytest


def lse(v):
    m = np.max(v)
    return m + math.log(np.sum(np.exp(v - m)))


def oracle_filter(g, o, N=100, seed=3, keep_prev=False):
    m = g.models.lgssm2()
    ys = g.models.simulate(m, 3)
    return m, ys, o.OracleFilter(m.model_id, m.params, N, seed, keep_prev=keep_prev).initialize(ys[0])


@pytest.mark.parametrize("method", ["multinomial", "residual"])
@pytest.mark.parametrize("n_new", [50, 150])
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_oracle_resize_invariants(g, o, method, n_new, alpha):
    """test/resize.jl:3-84: length, new_traces == old_traces[parents], log-ML preserved, residual minimum copies"""
    m, ys, f = oracle_filter(g, o)
    old_rows, lw = f.rows.copy(), f.lw.copy()
    old_lml = f.log_ml_estimate()
    lp = lw if alpha is None else alpha * lw
    w = np.exp(lp - lse(lp))
    f.resize(n_new, method, priority_alpha=alpha)
    assert f.n == n_new and f.rows.shape[0] == n_new and f.lw.size == n_new
    assert np.array_equal(f.rows, old_rows[f.parents - 1])
    assert abs(f.log_ml_estimate() - old_lml) < 1e-9
    if method == "residual":
        assert np.all(np.bincount(f.parents - 1, minlength=100) >= np.floor(n_new * w - 1e-9).astype(int))
    if alpha is None:
        assert np.all(f.lw == 0.0)


def test_oracle_resize_invalid_and_unknown(g, o):
    """test/resize.jl:31-37,79-84"""
    for method in ("multinomial", "residual"):
        m, ys, f = oracle_filter(g, o)
        f.lw[:] = -np.inf
        with pytest.raises(o.OracleError):
            f.resize(50, method, check=True)
        f.resize(50, method, check=False)
        assert f.n == 50 and np.all(f.lw == 0.0)
    m, ys, f = oracle_filter(g, o)
    with pytest.raises(o.OracleError):
        f.resize(50, "no-such-method")


@pytest.mark.parametrize("n_new", [25, 50, 99, 100])
def test_oracle_optimal_resize_invariants(g, o, n_new):
    """test/resize.jl:86-118 (pf_optimal_resize!): length, new_traces == old_traces[parents], kept particles keep
    log_weight + log(n/N), log-ML estimate preserved within rtol 1e-3; plus uniqueness of the parents (resize.jl:131-133)"""
    m, ys, f = oracle_filter(g, o)
    old_rows, lw, old_lml = f.rows.copy(), f.lw.copy(), f.log_ml_estimate()
    f.resize(n_new, "optimal")
    assert f.n == n_new and f.rows.shape[0] == n_new and f.lw.size == n_new
    assert np.array_equal(f.rows, old_rows[f.parents - 1])
    k = f.n_keep
    np.testing.assert_allclose(f.lw[:k], lw[f.parents[:k] - 1] + (math.log(n_new) - math.log(100)), rtol=0, atol=1e-12)
    assert np.unique(f.parents).size == n_new
    assert np.all(np.diff(f.parents[:k]) > 0) and np.all(np.diff(f.parents[k:]) > 0)
    assert -1e-12 <= f.log_ml_estimate() - old_lml <= 0.05      # test/resize.jl:109; see test_hip_optimal_resize_bitexact
    if n_new == 100:
        assert k == 100 and np.array_equal(f.parents, np.arange(1, 101))


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("n_new", [10, 40, 77])
def test_oracle_optimal_resize_vs_literal(g, o, seed, n_new):
    """the fixed-point threshold / keep set / systematic picks against the literal Float64 loops of resize.jl:149-219"""
    m, ys, f = oracle_filter(g, o, N=120, seed=seed)
    f.update(ys[1])
    lw = f.lw.copy()
    w = np.exp(lw - lse(lw))
    c = o.lib().lit_inv_w_threshold(np.ascontiguousarray(w), w.size, n_new)
    keep = c * w >= 1
    # the uniform the oracle will draw, as the Float64 the reference's rand() stands for
    from oracle.oracle import targets_multinomial
    U = int(targets_multinomial(f.seed, f.epoch, 0, 1, 1 << 62)[0]) / float(1 << 62)
    f.resize(n_new, "optimal")
    k = f.n_keep
    assert np.array_equal(f.parents[:k] - 1, np.flatnonzero(keep))
    strat = np.flatnonzero(~keep)
    wn = w[strat] / w[strat].sum()
    picks = np.empty(n_new - k, np.int64)
    cnt = o.lib().lit_systematic(np.ascontiguousarray(wn), strat.size, n_new - k, U, picks)
    assert cnt == n_new - k                                                              # @assert, resize.jl:181
    assert np.array_equal(f.parents[k:] - 1, strat[picks])
    rw = lse(lw) - math.log(c) + (math.log(n_new) - math.log(w.size))
    np.testing.assert_allclose(f.lw[k:], rw, rtol=0, atol=1e-9)


def test_oracle_optimal_resize_invalid(g, o):
    """test/resize.jl:111-117: check=true throws, check=false leaves all log-weights at -Inf"""
    m, ys, f = oracle_filter(g, o)
    f.lw[:] = -np.inf
    with pytest.raises(o.OracleError):
        f.resize(50, "optimal", check=True)
    f.resize(50, "optimal", check=False)
    assert f.n == 50 and np.all(f.lw == -np.inf)
    m, ys, f = oracle_filter(g, o)
    with pytest.raises(o.OracleError):
        f.resize(101, "optimal")


@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_oracle_replicate_dereplicate(g, o, layout):
    """test/resize.jl (replication / dereplication testsets): layouts, keepfirst inverts replicate, sample keeps block mass"""
    m, ys, f = oracle_filter(g, o, N=20)
    rows, lw, lml = f.rows.copy(), f.lw.copy(), f.log_ml_estimate()
    f.replicate(3, layout)
    assert f.n == 60 and abs(f.log_ml_estimate() - lml) < 1e-12
    want = np.repeat(np.arange(20), 3) if layout == "contiguous" else np.tile(np.arange(20), 3)
    assert np.array_equal(f.parents - 1, want) and np.array_equal(f.rows, rows[want]) and np.array_equal(f.lw, lw[want])
    f.dereplicate(3, layout, "keepfirst")
    assert f.n == 20 and np.array_equal(f.rows, rows) and np.array_equal(f.lw, lw)
    # :sample on blocks with different weights: new weight = logsumexp(block) - log k; total mass preserved
    f.replicate(3, layout)
    f.lw = f.lw + np.random.default_rng(1).normal(0, 1, 60)
    before = f.lw.copy(); lml = f.log_ml_estimate()
    f.dereplicate(3, layout, "sample")
    blocks = before.reshape(20, 3) if layout == "contiguous" else before.reshape(3, 20).T
    np.testing.assert_allclose(f.lw, [lse(b) - math.log(3) for b in blocks], rtol=1e-12)
    assert abs(f.log_ml_estimate() - lml) < 1e-9
    sel = f.parents - 1
    assert np.all((sel // 3 == np.arange(20)) if layout == "contiguous" else (sel % 20 == np.arange(20)))


# ------------------------------------------------------------------------------------------ GPU parity
def pair(g, o, N, seed=4, keep_prev=False, name="lgssm2"):
    model = g.models.by_name(name)
    ys = g.models.simulate(model, 4)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=keep_prev)
    orc = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=keep_prev).initialize(ys[0])
    return model, ys, st, orc


def same(st, orc):
    assert st.n_particles == orc.n
    assert np.array_equal(st.parents, orc.parents)
    assert np.array_equal(st.traces, orc.rows)
    assert np.array_equal(st.log_weights, orc.lw)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["multinomial", "residual"])
@pytest.mark.parametrize("n_old,n_new", [(100, 50), (100, 150), (5000, 20000), (30000, 7000)])
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_hip_resize_bitexact(g, o, method, n_old, n_new, alpha):
    model, ys, st, orc = pair(g, o, n_old)
    g.pf_resize(st, n_new, method, priority_fn=None if alpha is None else g.Tempering(alpha), check=False)
    orc.resize(n_new, method, priority_alpha=alpha, check=False)
    same(st, orc)
    np.testing.assert_allclose(g.get_lml_est(st), orc.log_ml_estimate(), rtol=1e-12)
    # the resized filter keeps working: update, resample, ESS at the new size
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    g.pf_resample(st, "stratified", check=False); orc.resample("stratified", check=False)
    g.pf_update(st, (3,), (None,), ys[2]); orc.update(ys[2])
    same(st, orc)
    assert g.get_ess(st) == orc.effective_sample_size() and g.get_lml_est(st) == orc.log_ml_estimate()


@pytest.mark.gpu
@pytest.mark.parametrize("n_old,n_new", [(100, 25), (100, 50), (100, 100), (5000, 1234), (30000, 7000), (200000, 199000)])
@pytest.mark.parametrize("steps", [1, 2])
def test_hip_optimal_resize_bitexact(g, o, n_old, n_new, steps):
    """pf_optimal_resize! (resize.jl:149-219): keep set, systematic picks, weights bit for bit against the oracle"""
    model, ys, st, orc = pair(g, o, n_old)
    if steps == 2:
        g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    lml = g.get_lml_est(st)
    g.pf_resize(st, n_new, "optimal", check=False)
    orc.resize(n_new, "optimal", check=False)
    same(st, orc)
    assert g.get_lml_est(st) == orc.log_ml_estimate()
    # test/resize.jl:109 (rtol 1e-3 on a log-ML of O(100) there): the estimate moves by log(1 + kappa - B/(n-A)) >= 0,
    # the reference's threshold particle being kept although B counts it (resize.jl:156,215)
    assert 0 <= g.get_lml_est(st) - lml <= 0.05
    g.pf_update(st, (3,), (None,), ys[2]); orc.update(ys[2])
    g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
    same(st, orc)
    assert g.get_ess(st) == orc.effective_sample_size() and g.get_lml_est(st) == orc.log_ml_estimate()


@pytest.mark.gpu
def test_hip_optimal_resize_degenerate(g, o):
    """all -Inf (test/resize.jl:111-117) and a single dominant particle (the others underflow to weight 0)"""
    model, ys, st, orc = pair(g, o, 100)
    st.log_weights = np.full(100, -np.inf); orc.lw[:] = -np.inf
    with pytest.raises(g.ErrorException):
        g.pf_optimal_resize(st, 50, check=True)
    assert st.n_particles == 100
    g.pf_optimal_resize(st, 50, check=False); orc.resize(50, "optimal", check=False)
    same(st, orc)
    assert np.all(st.log_weights == -np.inf)
    model, ys, st, orc = pair(g, o, 1000)
    lw = np.full(1000, -500.0); lw[37] = 0.0; lw[400] = -1.0
    st.log_weights = lw; orc.lw[:] = lw
    g.pf_optimal_resize(st, 10, check=False); orc.resize(10, "optimal", check=False)
    same(st, orc)
    assert list(st.parents[:2]) == [38, 401]


@pytest.mark.gpu
def test_hip_resize_errors(g, o):
    model, ys, st, orc = pair(g, o, 100)
    with pytest.raises(g.ErrorException):
        g.pf_resize(st, 50, "no-such-method")
    with pytest.raises(Exception):
        g.pf_resize(st, 101, "optimal")
    st.log_weights = np.full(100, -np.inf)
    with pytest.raises(g.ErrorException):
        g.pf_multinomial_resize(st, 50, check=True)
    assert st.n_particles == 100                       # a failed resize leaves the filter untouched
    g.pf_multinomial_resize(st, 50, check=False)
    assert st.n_particles == 50 and np.all(st.log_weights == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
@pytest.mark.parametrize("name", ["lgssm2", "bearings4"])
def test_hip_replicate_dereplicate_bitexact(g, o, layout, name):
    model, ys, st, orc = pair(g, o, 3000, keep_prev=True, name=name)
    g.pf_replicate(st, 4, layout=layout); orc.replicate(4, layout)
    same(st, orc)
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    same(st, orc)
    g.pf_dereplicate(st, 4, layout=layout, method="sample"); orc.dereplicate(4, layout, "sample")
    same(st, orc)
    g.pf_replicate(st, 2, layout=layout); orc.replicate(2, layout)
    g.pf_dereplicate(st, 2, layout=layout, method="keepfirst"); orc.dereplicate(2, layout, "keepfirst")
    same(st, orc)
    assert g.get_lml_est(st) == orc.log_ml_estimate()
