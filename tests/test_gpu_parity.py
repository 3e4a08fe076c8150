"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): ancestor indices bit-exact; log-weights / log-ML within 1e-6
relative (the spec is deterministic, so in practice they are bit-identical and asserted as such
where the arithmetic is element-wise)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-6          # north_star tolerance for floating-point outputs

MODELS = ["lgssm2", "bearings4", "sv1", "object_motion"]


def make_pair(g, o, name, N, seed, keep_prev, T=6):
    model = g.models.by_name(name)
    ys = g.models.simulate(model, T)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=keep_prev)
    orc = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=keep_prev).initialize(ys[0])
    return model, ys, st, orc


def assert_state_equal(st, orc, exact=True):
    rows = st.traces
    if exact:
        assert np.array_equal(rows, orc.rows), "particle rows differ"
        assert np.array_equal(st.log_weights, orc.lw), "log-weights differ"
    else:
        np.testing.assert_allclose(rows, orc.rows, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(st.log_weights, orc.lw, rtol=RTOL, atol=1e-9)


# ------------------------------------------------------------------ math spec, bit for bit
@pytest.mark.parametrize("which,lo,hi", [(0, -745.0, 20.0), (1, 1e-300, 1e300), (2, 0.0, 1.0), (3, -4.0, 4.0),
                                         (4, 0.0, 1e10), (5, -1e3, 1e3)])
def test_device_math_bitwise(g, o, which, lo, hi):
    rng = np.random.default_rng(which)
    n = 200_000
    a = rng.uniform(lo, hi, n)
    b = rng.uniform(lo, hi, n)
    if which == 1:
        a = np.exp(rng.uniform(-700, 700, n))
    if which == 5:
        b[b == 0] = 1.0
    model = g.models.lgssm2()
    st = g.DeviceParticleFilterState(model, 16)
    d1, d2 = st.debug_math(which, a, b)
    o1, o2 = np.empty(n), np.zeros(n)
    o.lib().o_math_vec(which, a, b, n, o1, o2)
    assert np.array_equal(d1.view(np.uint64), o1.view(np.uint64))
    if which == 2:
        assert np.array_equal(d2.view(np.uint64), o2.view(np.uint64))


# ------------------------------------------------------------------ initialize / update (initialize.jl:31-44, update.jl:12-25)
@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("keep_prev", [False, True])
def test_initialize_update_bitwise(g, o, name, keep_prev):
    model, ys, st, orc = make_pair(g, o, name, 5000, 11, keep_prev)
    assert_state_equal(st, orc)
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t])
        orc.update(ys[t])
        assert_state_equal(st, orc)
    assert g.get_ess(st) == orc.effective_sample_size()
    assert g.get_lml_est(st) == orc.log_ml_estimate()
    np.testing.assert_allclose(g.get_norm_weights(st), orc.norm_weights(), rtol=1e-12)
    np.testing.assert_allclose(g.get_log_norm_weights(st), orc.log_norm_weights(), rtol=1e-12, atol=1e-12)
    assert np.array_equal(st.parents, np.arange(1, 5001))


# ------------------------------------------------------------------ resampling (resample.jl:48-175)
@pytest.mark.parametrize("method", ["multinomial", "residual", "stratified"])
@pytest.mark.parametrize("N", [1, 7, 100, 2048, 2049, 50_000])
def test_resample_ancestors_bitexact(g, o, method, N):
    model, ys, st, orc = make_pair(g, o, "lgssm2", N, 5, False)
    for t in range(1, 4):
        kw = dict(sort_particles=(t % 2 == 0)) if method == "stratified" else {}
        g.pf_resample(st, method, check=False, **kw)
        orc.resample(method, check=False, **kw)
        assert np.array_equal(st.parents, orc.parents), f"ancestors differ at t={t}"
        assert_state_equal(st, orc)
        assert g.get_lml_est(st) == orc.log_ml_estimate()
        g.pf_update(st, (t + 1,), (None,), ys[t])
        orc.update(ys[t])
        assert_state_equal(st, orc)


@pytest.mark.parametrize("method", ["multinomial", "residual", "stratified"])
def test_resample_priorities(g, o, method):
    """priority_fn = w -> w/2 (test/resample.jl:15-23,56-70,104-119): ancestors exact, new weights and log-ML
    within tolerance, and logsumexp(lw) == log N afterwards."""
    N = 10_000
    model, ys, st, orc = make_pair(g, o, "lgssm2", N, 9, False)
    old = g.get_lml_est(st)
    g.pf_resample(st, method, priority_fn=g.Tempering(0.5), check=False)
    orc.resample(method, priority_alpha=0.5, check=False)
    assert np.array_equal(st.parents, orc.parents)
    np.testing.assert_allclose(st.log_weights, orc.lw, rtol=RTOL, atol=1e-9)
    assert abs(g.get_lml_est(st) - old) <= 1e-9 * abs(old)
    # arbitrary closure -> host-evaluated priorities, same result
    model, ys, st2, _ = make_pair(g, o, "lgssm2", N, 9, False)
    g.pf_resample(st2, method, priority_fn=lambda w: w / 2, check=False)
    assert np.array_equal(st2.parents, orc.parents)
    np.testing.assert_allclose(st2.log_weights, orc.lw, rtol=RTOL, atol=1e-9)


@pytest.mark.parametrize("method", ["multinomial", "residual", "stratified"])
def test_resample_invalid_weights(g, o, method):
    """test/resample.jl:26-31,73-78,122-127: all -Inf weights: check=true throws, check=false leaves all zeros."""
    N = 100
    model, ys, st, orc = make_pair(g, o, "lgssm2", N, 2, False)
    st.log_weights = np.full(N, -np.inf)
    with pytest.raises(g.ErrorException):
        g.pf_resample(st, method, check=True)
    g.pf_resample(st, method, check=False)
    assert np.all(st.log_weights == 0.0)
    orc.lw[:] = -np.inf
    orc.resample(method, check=False)
    assert np.array_equal(st.parents, orc.parents)
    with pytest.warns(UserWarning):
        st.log_weights = np.full(N, -np.inf)
        g.pf_resample(st, method, check="warn")
    st.log_weights = np.full(N, np.nan)
    with pytest.raises(g.ErrorException):
        g.pf_resample(st, method, check="warn")
    with pytest.raises(g.ErrorException):
        g.pf_resample(st, "systematic")


@pytest.mark.parametrize("method", ["residual", "stratified"])
def test_uniform_weights_identity(g, o, method):
    """test/resample.jl:36-40,83-87: equal weights => residual and stratified resampling are the identity."""
    N = 100
    model, ys, st, orc = make_pair(g, o, "lgssm2", N, 2, False)
    st.log_weights = np.zeros(N)
    before = st.traces
    g.pf_resample(st, method)
    assert np.array_equal(st.parents, np.arange(1, N + 1))
    assert np.array_equal(st.traces, before)


# ------------------------------------------------------------------ rejuvenation (rejuvenate.jl:40-90)
@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("method", ["move", "reweight"])
def test_rejuvenate_bitwise(g, o, name, method):
    model, ys, st, orc = make_pair(g, o, name, 4000, 21, True)
    # right after initialize (no previous step): proposals come from the prior
    g.pf_rejuvenate(st, None, (), 2, method=method, count=True)
    orc.rejuvenate(method, 2)
    assert st.n_accepted == orc.n_accepted
    assert_state_equal(st, orc)
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
        g.pf_rejuvenate(st, None, (), 1, method=method, count=True); orc.rejuvenate(method, 1)
        assert st.n_accepted == orc.n_accepted
        assert np.array_equal(st.parents, orc.parents)
        assert_state_equal(st, orc)
    with pytest.raises(g.ErrorException):
        g.pf_rejuvenate(st, None, (), 1, method="gibbs")


# ------------------------------------------------------------------ statistics (statistics.jl:13-14,48-50)
@pytest.mark.parametrize("name", MODELS)
def test_mean_var(g, o, name):
    model, ys, st, orc = make_pair(g, o, name, 30_000, 4, False)
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    for c in range(model.dim):
        # the summation order is part of the spec (binary tree over the indices): bit-identical, like everything else
        assert g.mean(st, c) == orc.mean(c) and g.var(st, c) == orc.var(c)
        assert np.array_equal(st.column(c), orc.column(c))


@pytest.mark.parametrize("N", [1, 7, 2048, 2049, 100_003, 4_200_000])
def test_mean_var_tree_sum_any_size(g, o, N):
    """one chunk, a ragged chunk, two levels (N > 2048) and three levels (N > 2048^2) of the tree, weights that differ by 300 nats"""
    model = g.models.sv1(); ys = g.models.simulate(model, 2)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=3)
    orc = o.OracleFilter(model.model_id, model.params, N, 3).initialize(ys[0])
    lw = -300.0 * np.random.default_rng(N).random(N)
    st.log_weights = lw; orc.lw = lw.copy()
    assert g.mean(st, 0) == orc.mean(0) and g.var(st, 0) == orc.var(0)
    st.close()


# ------------------------------------------------------------------ Gen.sample_unweighted_traces (utils.jl:189-194)
@pytest.mark.parametrize("n_samples", [10, 5000, 40_000])
def test_sample_unweighted_traces(g, o, n_samples):
    model, ys, st, orc = make_pair(g, o, "lgssm2", 20_000, 8, False)
    before = (st.traces, st.log_weights, g.get_lml_est(st))
    rows, idx = g.sample_unweighted_traces(st, n_samples, return_indices=True)
    orows, oidx = orc.sample_unweighted(n_samples)
    assert np.array_equal(idx, oidx) and np.array_equal(rows, orows)
    assert np.array_equal(rows, before[0][idx - 1])
    assert np.array_equal(st.traces, before[0]) and np.array_equal(st.log_weights, before[1]) and g.get_lml_est(st) == before[2]
    if n_samples >= 5000:                                  # high-weight particles are drawn more often
        cnt = np.bincount(idx - 1, minlength=20_000)
        assert np.corrcoef(cnt, g.get_norm_weights(st))[0, 1] > 0.5


# ------------------------------------------------------------------ adversarial weight vectors
def _weight_patterns(N, rng):
    i = np.arange(N, dtype=np.float64)
    pats = {
        "one_dominant": np.where(i == N // 3, 0.0, -700.0),                    # every other weight underflows to q = 0
        "two_clusters": np.where(i % 2 == 0, 0.0, -30.0),
        "geometric": -0.01 * i,
        "mostly_neginf": np.where(rng.random(N) < 0.01, rng.normal(size=N), -np.inf),
        "underflow_edge": -745.0 + 45.0 * rng.random(N),                       # exp(lw) spans the subnormal edge
        "block_ties": np.floor(i / 37.0) * -0.5,
        "tiny_differences": 1e-15 * rng.integers(0, 4, N),
        "huge_offset": 1e6 + rng.normal(size=N),                               # only differences matter
        "first_and_last": np.where((i == 0) | (i == N - 1), 0.0, -50.0),
    }
    pats["mostly_neginf"][N - 1] = 0.0                                         # at least one finite weight
    return pats


@pytest.mark.parametrize("N", [4097, 70_001])
def test_adversarial_weights_all_resamplers(g, o, N):
    """weight vectors chosen to stress the fixed-point normalisation, the CDF levels and the search boundaries:
    ancestors, weights, ESS and log-ML must still equal the oracle bit for bit, for every resampler and the optimal resize"""
    rng = np.random.default_rng(1234 + N)
    for name, lw in _weight_patterns(N, rng).items():
        for method, kw in (("multinomial", {}), ("residual", {}), ("stratified", {"sort_particles": True}),
                           ("stratified", {"sort_particles": False})):
            model, ys, st, orc = make_pair(g, o, "lgssm2", N, 17, False, T=3)
            st.log_weights = lw; orc.lw[:] = lw
            assert g.get_ess(st) == orc.effective_sample_size() or (np.isnan(g.get_ess(st)) and np.isnan(orc.effective_sample_size())), name
            assert g.get_lml_est(st) == orc.log_ml_estimate(), name
            g.pf_resample(st, method, check=False, **kw); orc.resample(method, check=False, **kw)
            assert np.array_equal(st.parents, orc.parents), f"{name} / {method} {kw}"
            assert_state_equal(st, orc)
            g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
            assert_state_equal(st, orc)
        model, ys, st, orc = make_pair(g, o, "lgssm2", N, 17, False, T=3)
        st.log_weights = lw; orc.lw[:] = lw
        g.pf_resize(st, N // 3, "optimal", check=False); orc.resize(N // 3, "optimal", check=False)
        assert np.array_equal(st.parents, orc.parents), f"{name} / optimal resize"
        assert_state_equal(st, orc)


def test_invalid_arguments_fail_loudly(g):
    """wrong observation length (an empty choicemap() has no device meaning), zero particles, update before initialize"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 2)
    with pytest.raises(Exception):
        g.pf_initialize(model, (1,), np.zeros(0), 100)
    with pytest.raises(Exception):
        g.pf_initialize(model, (1,), np.zeros(3), 100)
    with pytest.raises(Exception):
        g.pf_initialize(model, (1,), ys[0], 0)
    st = g.pf_initialize(model, (1,), ys[0], 100)
    with pytest.raises(Exception):
        g.pf_update(st, (2,), (None,), np.zeros(1))
    g.pf_update(st, (2,), (None,), ys[1])                      # the handle is still usable after a rejected call
    assert np.isfinite(g.get_lml_est(st))


def test_no_device_memory_leak_over_handle_lifetimes(g):
    """create / use every code path that allocates lazily / destroy, many times: free device memory must come back"""
    import torch
    from gpf_amd import sharded
    model = g.models.bearings4(); ys = g.models.simulate(model, 4)

    def cycle():
        st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3, keep_prev=True, history=8)
        g.pf_resample(st, "stratified", check=False, sort_particles=True)          # sort buffers
        g.pf_rejuvenate(st, g.mh, (), 1)
        g.pf_update(st, (2,), (None,), ys[1])
        g.pf_resample(st, "residual", check=False)                                  # residual channels
        g.mean(st, (1, 0)); g.get_ess(st)
        st.close()
        st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3)
        g.pf_update(st[1000:5000], (3,), (None,), ys[2])                            # a view
        g.pf_resize(st, 50_000, "optimal", check=False); g.pf_replicate(st, 3); g.sample_unweighted_traces(st, 1000)
        st.close()
        sh = sharded.pf_initialize(model, (1,), ys[0], 200_000, seed=3)             # staging list, counters, pinned mirror, event
        sharded.pf_resample(sh, "multinomial", check=False); sharded.pf_update(sh, (2,), (None,), ys[1])
        sharded.get_lml_est(sh)
        sh.local.close()

    import gc

    def free_after(k):
        for _ in range(k):
            cycle()
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()      # torch's own cache is not the library's
        return torch.cuda.mem_get_info()[0]

    free_after(3)
    free0 = free_after(12)              # the HIP runtime's own pools have settled by now
    free1 = free_after(25)
    # a leak grows with the number of lifetimes (one 200 000-particle column per lifetime = 40 MB here); pools do not
    assert free0 - free1 < 32 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 25 lifetimes"


def test_mean_var_functional_forms(g, o):
    """mean(f, state, addrs...) / var(f, state, addrs...) (src/statistics.jl:28-38, 65-82; test/statistics.jl): host closures
    over one or several addresses, vectorised or scalar; consistent with the plain forms"""
    model, ys, st, orc = make_pair(g, o, "lgssm2", 5000, 4, False)
    g.pf_update(st, (2,), (None,), ys[1])
    m0, v0 = g.mean(st, 0), g.var(st, 0)
    assert abs(g.mean(lambda x: x, st, 0) - m0) < 1e-12 and abs(g.var(lambda x: x, st, 0) - v0) < 1e-12
    w = g.get_norm_weights(st); X = st.traces
    np.testing.assert_allclose(g.mean(lambda x, y: x * y, st, 0, 1), np.sum(w * X[:, 0] * X[:, 1]), rtol=1e-12)
    np.testing.assert_allclose(g.mean(lambda x: float(x) ** 2 if x > 0 else 0.0, st, 1),          # scalar closure with a branch
                               np.sum(w * np.where(X[:, 1] > 0, X[:, 1] ** 2, 0.0)), rtol=1e-12)
    fv = np.hypot(X[:, 0], X[:, 1]); mu = np.sum(w * fv)
    np.testing.assert_allclose(g.var(np.hypot, st, 0, 1), np.sum(w * (fv - mu) ** 2), rtol=1e-10)


@pytest.mark.parametrize("N", [1024, 2048, 4096, 65536, 1 << 20])
def test_all_weights_equal_power_of_two(g, o, N):
    """S = N 2^K = 2^62 EXACTLY when N >= 1024 is a power of two and every weight equals the maximum -- e.g. a second resample right
    after a resample (all log-weights 0).  The 4-byte keys (prefix >> 30) and the descriptor words must cope with that one
    value (found by tests/test_gpu_fuzz.py: the last key overflowed to 0 and every slot got the same ancestor)."""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=44)
    orc = o.OracleFilter(model.model_id, model.params, N, 44).initialize(ys[0])
    for method, kw in (("multinomial", {}), ("multinomial", {}), ("residual", {}), ("stratified", {"sort_particles": True}),
                       ("stratified", {"sort_particles": False}), ("multinomial", {})):
        g.pf_resample(st, method, check=False, **kw); orc.resample(method, check=False, **kw)
        assert np.array_equal(st.parents, orc.parents), (method, kw)
        assert g.get_ess(st) == orc.effective_sample_size() == N
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    st.close()


def test_three_filters_on_three_streams_nothing_synchronises(g, o):
    """handles own their streams: three filters stepped in turn with check = false (nothing ever waits) run concurrently on the
    device -- the inter-workgroup protocols of scan and sort (tickets, bounded waits) must hold with other kernels resident"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 20)
    N, methods = 300_000, ["multinomial", "stratified", "residual"]
    sts = [g.pf_initialize(model, (1,), ys[0], N, seed=s) for s in (1, 2, 3)]
    for t in range(1, 16):
        for k, st in enumerate(sts):
            m = methods[(t + k) % 3]
            g.pf_resample(st, m, check=False, **({"sort_particles": bool(t % 2)} if m == "stratified" else {}))
            g.pf_update(st, (t + 1,), (None,), ys[t])
    for k, st in enumerate(sts):
        orc = o.OracleFilter(model.model_id, model.params, N, k + 1).initialize(ys[0])
        for t in range(1, 16):
            m = methods[(t + k) % 3]
            orc.resample(m, check=False, **({"sort_particles": bool(t % 2)} if m == "stratified" else {})); orc.update(ys[t])
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw) and np.array_equal(st.parents, orc.parents)
        assert g.get_lml_est(st) == orc.log_ml_estimate()
        st.close()


@pytest.mark.gpu
def test_norm_weights_with_a_nan_among_finite_weights(g, o):
    """softmax of a weight vector with ONE NaN (or +Inf) is NaN everywhere (utils.jl:103-107: logsumexp is NaN), also where the weight
    itself is finite (found by the random sequences: a block-wise tempered resample leaves NaN weights in all -Inf blocks only)"""
    m = g.models.lgssm2(); ys = g.models.simulate(m, 2); N = 5000
    st = g.pf_initialize(m, (1,), ys[0], N, seed=2)
    orc = o.OracleFilter(m.model_id, m.params, N, 2).initialize(ys[0])
    for bad in (np.nan, np.inf):
        lw = st.log_weights.copy(); lw[17] = bad
        st.log_weights = lw; orc.lw = lw.copy()
        with np.errstate(invalid="ignore", divide="ignore"):
            nw, lnw = g.get_norm_weights(st), g.get_log_norm_weights(st)
            assert np.isnan(nw).all() and np.array_equal(nw, orc.norm_weights(), equal_nan=True)
            assert np.isnan(lnw).all() and np.array_equal(lnw, orc.log_norm_weights(), equal_nan=True)
        assert np.isnan(g.get_ess(st)) and np.isnan(g.get_lml_est(st))
    # all -Inf: get_norm_weights is the PLAIN softmax (utils.jl:103-107): maximum = -Inf, vs .- maximum = NaN -> NaN everywhere; only
    # safe_softmax inside the resamplers falls back to uniform weights (utils.jl:123-126)
    lw = np.full(N, -np.inf)
    st.log_weights = lw; orc.lw = lw.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        nw, lnw = g.get_norm_weights(st), g.get_log_norm_weights(st)
        assert np.isnan(nw).all() and np.array_equal(nw, orc.norm_weights(), equal_nan=True)
        assert np.isnan(lnw).all() and np.array_equal(lnw, orc.log_norm_weights(), equal_nan=True)
    assert np.isnan(g.get_ess(st))
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)   # the resampler: uniform fallback
    assert np.array_equal(st.parents, orc.parents) and (st.log_weights == 0).all()
    st.close()
