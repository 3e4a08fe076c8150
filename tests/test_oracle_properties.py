"""Property tests (hypothesis) of the exact-integer resampling spec against the literal Float64 restatement of
src/resample.jl / src/utils.jl, on adversarial weight vectors: ragged sizes, ties, huge dynamic range, -Inf entries."""
import math

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

SET = dict(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])

weights_strategy = st.lists(
    st.one_of(st.floats(-30, 5), st.just(0.0), st.floats(-700, -600), st.just(-math.inf)),
    min_size=1, max_size=300)


def make(g, o, lw, seed=11):
    m = g.models.lgssm2()
    f = o.OracleFilter(m.model_id, m.params, len(lw), seed)
    f.lw = np.array(lw, np.float64)
    f.rows = np.arange(len(lw) * f.W, dtype=np.float64).reshape(len(lw), f.W)
    return f


@given(lw=weights_strategy, method=st.sampled_from(["multinomial", "residual", "stratified"]), sort=st.booleans())
@settings(**SET)
def test_resample_properties(g, o, lw, method, sort):
    f = make(g, o, lw)
    N = f.n
    rows0, lw0 = f.rows.copy(), f.lw.copy()
    all_neg = bool(np.all(np.isneginf(lw0)))
    s0 = f.summary()
    lml0 = f.log_ml_estimate()
    invalid = f.resample(method, sort_particles=sort, check=False)
    par = f.parents - 1
    assert invalid == all_neg
    assert par.min() >= 0 and par.max() < N
    assert np.array_equal(f.rows, rows0[par])                         # test/resample.jl:11
    assert np.all(f.lw == 0.0)                                        # resample.jl:195
    if not all_neg:
        assert abs(f.log_ml_estimate() - lml0) <= 1e-9 * max(1.0, abs(lml0))   # test/resample.jl:12
        q = s0.q
        assert np.all(q[par] > 0)                                     # zero-mass particles are never selected
        counts = np.bincount(par, minlength=N)
        if method == "residual":
            assert np.all(counts >= (N * q.astype(object)) // s0.S)   # test/resample.jl:47-52, exact floor
        if method == "stratified" and not sort:
            assert np.all(np.diff(par) >= 0)
            # every particle gets floor or ceil of its expected copies, +-1 (stratified property)
            exp = N * q.astype(np.float64) / float(s0.S)
            assert np.all(np.abs(counts - exp) < 2.0)
    else:
        assert math.isinf(f.lml_est) and f.lml_est < 0


@given(lw=st.lists(st.floats(-20, 3), min_size=2, max_size=200), seed=st.integers(1, 1000))
@settings(**SET)
def test_spec_equals_literal_float64(g, o, lw, seed):
    """identical ancestors from the fixed-point spec and from the literal Float64 arithmetic (same indexed uniforms)"""
    lw = np.array(lw)
    N = lw.size
    L = o.lib()
    w = np.empty(N)
    assert L.lit_safe_softmax(lw, N, w) == 0
    u = np.array([L.o_resample_u52_d(seed, j, 0) for j in range(N)])
    f = make(g, o, lw, seed)
    f.resample("multinomial")
    par = np.empty(N, np.int64); L.lit_multinomial(w, N, u, par)
    agree = np.mean(f.parents - 1 == par)
    assert agree == 1.0 or agree > 0.98        # a uniform within ~2^-40 of a CDF edge may fall either side
    f = make(g, o, lw, seed)
    f.resample("stratified", sort_particles=False)
    L.lit_stratified(w, np.arange(N, dtype=np.int64), N, u, par)
    agree = np.mean(f.parents - 1 == par)
    assert agree == 1.0 or agree > 0.98
    # update_weights! without priorities (resample.jl:195): every log-weight 0.0, so logsumexp = log N
    assert (f.lw == 0.0).all() and abs(f.summary().lse - np.log(N)) < 1e-12
    np.testing.assert_allclose(make(g, o, lw).summary().lse, L.lit_logsumexp(lw, N), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(make(g, o, lw).effective_sample_size(), L.lit_ess(lw, N), rtol=1e-6)


@given(lw=st.lists(st.floats(-10, 3), min_size=4, max_size=120), n_new=st.integers(1, 300),
       method=st.sampled_from(["multinomial", "residual"]))
@settings(**SET)
def test_resize_properties(g, o, lw, n_new, method):
    f = make(g, o, lw)
    rows0 = f.rows.copy(); lml0 = f.log_ml_estimate()
    f.resize(n_new, method, check=False)
    assert f.n == n_new and f.rows.shape[0] == n_new
    assert np.array_equal(f.rows, rows0[f.parents - 1])               # test/resize.jl:13
    assert abs(f.log_ml_estimate() - lml0) <= 1e-9 * max(1.0, abs(lml0))   # test/resize.jl:14


def test_weights_64_below_the_maximum_have_fixed_point_weight_zero(o):
    """the bucket sort behind sort_particles=true leaves keys more than 2^6 below the maximum unranked ("dead" keys, gpf_k_sort.hpp
    SORT_COARSE_DEAD): that is only right if their fixed-point weight q = trunc(exp(d) 2^K + 1/2) is 0 for every
    admissible K <= 52 -- and it leaves a margin: the last nonzero weight is 53 ln 2 = 36.7 below the maximum"""
    L = o.lib()
    for K in range(1, 53):
        for d in (-64.0, -64.0 - 2.0 ** -40, -100.0, -512.0, -707.9, -1e300, -np.inf):
            assert L.o_exp_fix_d(d, K) == 0, (K, d)
    assert L.o_exp_fix_d(-36.0, 52) > 0 and L.o_exp_fix_d(-37.5, 52) == 0
