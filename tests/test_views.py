"""Sub-state views (reference src/view.jl, test/resample.jl:130-162, test/update.jl:179-189,
test/rejuvenate.jl:73-103): block-wise operations on contiguous ranges of one filter."""
import numpy as np
import pytest

METHODS = ["multinomial", "residual", "stratified"]


def lgssm(g, o, N=100, seed=3, keep_prev=True, T=4):
    m = g.models.lgssm2(); ys = g.models.simulate(m, T)
    return m, ys, o.OracleFilter(m.model_id, m.params, N, seed, keep_prev=keep_prev).initialize(ys[0])


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_oracle_blockwise_resampling(g, o, method, alpha):
    """test/resample.jl:130-162: resample two 50-particle views independently; per view new == old[parents] and the
    view's log-ML estimate is unchanged; globally the traces follow the concatenated parents and the log-ML is unchanged."""
    m, ys, f = lgssm(g, o)
    old_full, lml_full, parents_full = f.rows.copy(), f.log_ml_estimate(), []
    for a in (0, 50):
        v = f[a:a + 50]
        old, lml = v.rows.copy(), v.log_ml_estimate()
        v.resample(method, priority_alpha=alpha)
        assert np.array_equal(v.rows, old[v.parents - 1])            # :151
        assert abs(v.log_ml_estimate() - lml) < 1e-9                 # :152
        parents_full += list(a + v.parents - 1)                       # :153
    assert np.array_equal(f.rows, old_full[np.array(parents_full)])  # :158
    assert abs(f.log_ml_estimate() - lml_full) < 1e-9                # :160
    assert f.lml_est == 0.0                                           # sub-states never touch the running estimate


def test_oracle_per_view_update_and_rejuvenation(g, o):
    """test/update.jl:179-189 (each half updated with its own observation) and test/rejuvenate.jl:73-103 (move on one
    view, reweight on the other: only the touched half changes)."""
    m, ys, f = lgssm(g, o)
    lw0, rows0 = f.lw.copy(), f.rows.copy()
    f[0:50].update(ys[1]); f[50:100].update(ys[2])
    assert np.all(f.lw != lw0) and np.all(f.rows[:, :2] != rows0[:, :2])
    sr = m.info["sr"]
    for sl, y in ((slice(0, 50), ys[1]), (slice(50, 100), ys[2])):
        want = sum(-0.5 * ((y[k] - f.rows[sl, k]) / sr) ** 2 - np.log(sr) - 0.5 * np.log(2 * np.pi) for k in range(2))
        np.testing.assert_allclose(f.lw[sl] - lw0[sl], want, rtol=1e-10, atol=1e-10)
    lw1, rows1 = f.lw.copy(), f.rows.copy()
    a = f[0:50]; a.last_obs = ys[1]; a.rejuvenate("move", 1)
    b = f[50:100]; b.last_obs = ys[2]; b.rejuvenate("reweight", 1)
    assert np.array_equal(f.lw[:50], lw1[:50])                        # move-accept leaves weights alone
    assert np.all(f.lw[50:] != lw1[50:]) and np.all(f.rows[50:, :2] != rows1[50:, :2])
    assert np.array_equal(f.rows[:50, 2:4], rows1[:50, 2:4])           # x_{t-1} untouched


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_oracle_blockwise_resampling_interleaved(g, o, method, alpha):
    """the block-wise resampling test of test/resample.jl:130-162 on INTERLEAVED blocks state[k:5:100] (the index sets of
    test/initialize.jl:60, test/update.jl:33, test/resize.jl:138): per view new == old[parents], its log-ML estimate is
    unchanged, the other blocks are untouched, and the whole filter's estimate is unchanged."""
    m, ys, f = lgssm(g, o)
    lml_full = f.log_ml_estimate()
    for k in range(5):
        v = f[k:100:5]
        assert v.n == 20
        before = f.rows.copy()
        old, lml = v.rows.copy(), v.log_ml_estimate()
        v.resample(method, priority_alpha=alpha)
        assert np.array_equal(v.rows, old[v.parents - 1]) and abs(v.log_ml_estimate() - lml) < 1e-9
        others = np.ones(100, bool); others[k::5] = False
        assert np.array_equal(f.rows[others], before[others])
        assert np.all((f.parents[k::5] >= 1) & (f.parents[k::5] <= 20))   # local to the view
    assert abs(f.log_ml_estimate() - lml_full) < 1e-9 and f.lml_est == 0.0


def test_oracle_strided_view_keeps_global_rng_ids(g, o):
    """pf_update!(state[k:5:100]) for every k with the same observation == ... the particles of view k carry the RNG counters
    k, k+5, ...: two filters updated through different partitions into strided views agree particle by particle when the
    epochs agree (one view per filter here), and a strided view differs from a contiguous one over other particles"""
    m, ys, f = lgssm(g, o)
    h = o.OracleFilter(m.model_id, m.params, 100, 3, keep_prev=True).initialize(ys[0])
    f[3:100:5].update(ys[1]); h[3:100:5].update(ys[1])
    assert np.array_equal(f.rows, h.rows) and np.array_equal(f.lw, h.lw)
    k = o.OracleFilter(m.model_id, m.params, 100, 3, keep_prev=True).initialize(ys[0])
    k.update(ys[1])                                                       # whole filter, same epoch: same per-particle streams
    assert np.array_equal(f.rows[3::5], k.rows[3::5]) and np.array_equal(f.lw[3::5], k.lw[3::5])
    assert not np.array_equal(f.rows[4::5], k.rows[4::5])                 # untouched by the view


# ------------------------------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [None, 0.5])
@pytest.mark.parametrize("N,step", [(100, 5), (100, 2), (10_007, 3)])
def test_hip_strided_views_bitexact(g, o, method, alpha, N, step):
    """state[k:step:N] for every k (src/view.jl:35-48; test/initialize.jl:60, test/update.jl:33): update, getters, resample and
    rejuvenation through strided views, against the oracle bit for bit, then the source as a whole"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 5)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=6, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, 6, keep_prev=True).initialize(ys[0])
    pf = None if alpha is None else g.Tempering(alpha)
    kw = dict(sort_particles=True) if method == "stratified" else {}
    for k in range(step):
        sv, ov = st[k:N:step], orc[k:N:step]
        assert sv.n_particles == ov.n == len(range(k, N, step))
        y = ys[1 + k % 2]
        g.pf_update(sv, (2,), (None,), y); ov.update(y)
        assert np.array_equal(sv.traces, ov.rows) and np.array_equal(sv.log_weights, ov.lw)
        assert g.get_ess(sv) == ov.effective_sample_size()
        lml_v = g.get_lml_est(sv)
        assert lml_v == ov.log_ml_estimate()
        g.pf_resample(sv, method, priority_fn=pf, check=False, **kw); ov.resample(method, priority_alpha=alpha, check=False, **kw)
        assert np.array_equal(sv.parents, ov.parents)                 # local to the view
        np.testing.assert_allclose(g.get_lml_est(sv), lml_v, rtol=1e-9)
        g.pf_rejuvenate(sv, g.mh, (), 1); ov.rejuvenate("move", 1)
        assert np.array_equal(sv.traces, ov.rows) and np.array_equal(sv.log_weights, ov.lw)
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert np.array_equal(st.parents, orc.parents)
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
    g.pf_update(st, (3,), (None,), ys[3]); orc.update(ys[3])
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.parents, orc.parents)


@pytest.mark.gpu
def test_hip_strided_view_set_and_strata(g, o):
    """writes through a strided view land in the source (gpf_set_log_weights / gpf_set_rows), and a stratified update of a
    strided view of the discrete-latent model equals the oracle"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    st = g.pf_initialize(model, (1,), ys[0], 60, seed=2)
    v = st[1:60:4]
    lw = np.linspace(-3.0, 0.0, v.n_particles)
    v.log_weights = lw
    assert np.array_equal(st.log_weights[1:60:4], lw) and np.array_equal(v.log_weights, lw)
    m = g.models.line_model()
    s2 = g.pf_initialize(m, (0,), g.models.line_obs(0, 0.0), 100, seed=4)
    o2 = o.OracleFilter(m.model_id, m.params, 100, 4).initialize(g.models.line_obs(0, 0.0))
    for k in range(2):
        g.pf_update(s2[k:100:2], (1,), (None,), g.models.line_obs(1, 0.0), [{"outlier": 0.0}, {"outlier": 1.0}], layout="interleaved")
        o2[k:100:2].update(g.models.line_obs(1, 0.0), strata=[0.0, 1.0], layout="interleaved")
    assert np.array_equal(s2.traces, o2.rows) and np.array_equal(s2.log_weights, o2.lw)


@pytest.mark.gpu
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [None, 0.5])
@pytest.mark.parametrize("N,cut", [(100, 50), (10_000, 3_000)])
def test_hip_views_bitexact(g, o, method, alpha, N, cut):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 5)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=6, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, 6, keep_prev=True).initialize(ys[0])
    pf = None if alpha is None else g.Tempering(alpha)
    kw = dict(sort_particles=True) if method == "stratified" else {}
    lml_full = g.get_lml_est(st)
    for (a, b), y in (((0, cut), ys[1]), ((cut, N), ys[2])):
        sv, ov = st[a:b], orc[a:b]
        assert sv.n_particles == b - a
        g.pf_update(sv, (2,), (None,), y); ov.update(y)
        assert g.get_ess(sv) == ov.effective_sample_size()
        lml_v = g.get_lml_est(sv)
        assert lml_v == ov.log_ml_estimate()
        g.pf_resample(sv, method, priority_fn=pf, check=False, **kw); ov.resample(method, priority_alpha=alpha, check=False, **kw)
        assert np.array_equal(sv.parents, ov.parents)                 # local to the view
        np.testing.assert_allclose(g.get_lml_est(sv), lml_v, rtol=1e-9)
        g.pf_rejuvenate(sv, g.mh, (), 1); ov.rejuvenate("move", 1)
        assert np.array_equal(sv.traces, ov.rows) and np.array_equal(sv.log_weights, ov.lw)
    # the source sees everything the views did
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert np.array_equal(st.parents, orc.parents)
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    # and keeps working as a whole
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
    g.pf_update(st, (3,), (None,), ys[3]); orc.update(ys[3])
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.parents, orc.parents)


@pytest.mark.gpu
def test_hip_view_errors(g, o):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    st = g.pf_initialize(model, (1,), ys[0], 100, seed=1)
    v = st[10:60]
    with pytest.raises(g.ErrorException):
        st[50:200][0:1]                                   # out of bounds / view of a view
    with pytest.raises(g.ErrorException):
        g.pf_resize(v, 10)
    g.pf_resize(st, 80)
    with pytest.raises(g.ErrorException):
        g.get_ess(v)                                      # stale after the parent was resized


@pytest.mark.gpu
@pytest.mark.parametrize("method", METHODS)
def test_hip_views_are_live(g, o, method):
    """A view held across changes made through the source or through an OVERLAPPING view answers for the weights as they
    are now (the reference's sub-states are SubArray views, src/view.jl:35-48), and odd view starts (the lane's slot run
    starts on the odd half of a Philox block) resample bit for bit."""
    N = 4000
    model = g.models.lgssm2(); ys = g.models.simulate(model, 6)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=9, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, 9, keep_prev=True).initialize(ys[0])
    kw = dict(sort_particles=False) if method == "stratified" else {}
    v, ov = st[501:2604], orc[501:2604]                      # odd start, odd length
    w, ow = st[2000:3500], orc[2000:3500]                    # overlaps v
    assert g.get_ess(v) == ov.effective_sample_size() and g.get_lml_est(v) == ov.log_ml_estimate()
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])                    # through the source
    assert g.get_ess(v) == ov.effective_sample_size() and g.get_lml_est(v) == ov.log_ml_estimate()
    g.pf_update(w, (3,), (None,), ys[2]); ow.update(ys[2])                      # through the overlapping view
    assert g.get_ess(v) == ov.effective_sample_size() and g.get_lml_est(v) == ov.log_ml_estimate()
    g.pf_update(v, (3,), (None,), ys[3]); ov.update(ys[3])
    g.pf_update(st, (4,), (None,), ys[4]); orc.update(ys[4])
    g.pf_resample(v, method, check=False, **kw); ov.resample(method, check=False, **kw)      # stale producer maxima would break this
    assert np.array_equal(v.parents, ov.parents)
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    assert g.get_ess(w) == ow.effective_sample_size()
    assert g.get_ess(st) == orc.effective_sample_size() and g.get_lml_est(st) == orc.log_ml_estimate()
    assert np.array_equal(v.parents, st.parents[501:2604])   # parents of a view alias the source's


@pytest.mark.gpu
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("follow", ["update", "rejuvenate", "getters", "resample"])
def test_hip_whole_view_resample_is_the_library_local_resample(g, o, monkeypatch, method, follow):
    """pf_resample!(state[1:end], method): gpf_resample_local (deferred gather, the kept log-weight as one device constant)
    must equal the view handle's eager path AND the oracle's sub-state resample, whatever consumes the result next"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 5); N = 20_000
    kw = dict(sort_particles=True) if method == "stratified" else {}

    def run(eager):
        if eager:
            monkeypatch.setenv("GPF_VIEW_RESAMPLE", "eager")
        else:
            monkeypatch.delenv("GPF_VIEW_RESAMPLE", raising=False)
        st = g.pf_initialize(model, (1,), ys[0], N, seed=9, keep_prev=True)
        g.pf_update(st, (2,), (None,), ys[1])
        lml0 = g.get_lml_est(st)
        g.pf_resample(st[0:N], method, check="warn", **kw)
        if follow == "update":
            g.pf_update(st, (3,), (None,), ys[2])
        elif follow == "rejuvenate":
            g.pf_rejuvenate(st, g.mh, (), 1)
        elif follow == "resample":
            g.pf_resample(st, "multinomial", check=False)
            g.pf_update(st, (3,), (None,), ys[2])
        out = (st.parents.copy(), st.traces.copy(), st.log_weights.copy(), g.get_lml_est(st), g.get_ess(st))
        if follow == "getters":
            np.testing.assert_allclose(out[3], lml0, rtol=1e-12)       # a sub-state resample keeps the estimate (resample.jl:185-187)
        return out

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    orc = o.OracleFilter(model.model_id, model.params, N, 9, keep_prev=True).initialize(ys[0])
    orc.update(ys[1])
    orc[0:N].resample(method, check=False, **kw)
    if follow == "update":
        orc.update(ys[2])
    elif follow == "rejuvenate":
        orc.rejuvenate("move", 1)
    elif follow == "resample":
        orc.resample("multinomial", check=False); orc.update(ys[2])
    assert np.array_equal(b[0], orc.parents) and np.array_equal(b[1], orc.rows) and np.array_equal(b[2], orc.lw)
    assert b[3] == orc.log_ml_estimate() and b[4] == orc.effective_sample_size()


# ----------------------------------------------------------------------------------------------- state[idxs], any index vector
def _index_sets(N, rng):
    """a random partition of 0..N-1 into three index vectors (unsorted), plus a sorted non-contiguous one"""
    perm = rng.permutation(N)
    a, b = N // 3, (2 * N) // 3
    return [perm[:a], perm[a:b], perm[b:]], np.sort(rng.choice(N, N // 2, replace=False))


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_oracle_index_vector_views(g, o, method, alpha):
    """src/view.jl:35-48 takes ANY index vector.  The block-wise resampling invariants of test/resample.jl:130-162 on a random partition
    of the particles into three unsorted index sets: per view new == old[parents] and an unchanged estimate; the other particles
    untouched; the whole filter's estimate unchanged."""
    m, ys, f = lgssm(g, o, N=120)
    parts, _ = _index_sets(120, np.random.default_rng(5))
    lml_full = f.log_ml_estimate()
    for ix in parts:
        v = f[ix]
        old, lml, rest = v.rows.copy(), v.log_ml_estimate(), np.setdiff1d(np.arange(120), ix)
        rows_rest, lw_rest = f.rows[rest].copy(), f.lw[rest].copy()
        v.resample(method, priority_alpha=alpha)
        assert np.array_equal(v.rows, old[v.parents - 1])
        assert abs(v.log_ml_estimate() - lml) < 1e-9
        assert np.array_equal(f.rows[rest], rows_rest) and np.array_equal(f.lw[rest], lw_rest)
    assert abs(f.log_ml_estimate() - lml_full) < 1e-9 and f.lml_est == 0.0


def test_oracle_index_view_keeps_the_particles_own_rng_ids(g, o):
    """a particle updated through state[idxs] draws what it would draw through the whole filter: its RNG counter is its own index"""
    m, ys, f = lgssm(g, o, N=200)
    whole = o.OracleFilter(m.model_id, m.params, 200, 3, keep_prev=True).initialize(ys[0])
    whole.update(ys[1])
    ix = np.random.default_rng(1).permutation(200)[:77]
    f[ix].update(ys[1])
    assert np.array_equal(f.rows[ix], whole.rows[ix]) and np.array_equal(f.lw[ix], whole.lw[ix])
    rest = np.setdiff1d(np.arange(200), ix)
    assert not np.array_equal(f.rows[rest], whole.rows[rest])            # the others were not updated


@pytest.mark.gpu
@pytest.mark.parametrize("method", METHODS + ["multinomial_sorted"])
@pytest.mark.parametrize("N", [97, 5000])
def test_hip_index_vector_views_bitexact(g, o, method, N):
    """every pf_* operation through state[idxs] over permuted / sorted / boolean index sets == the oracle's sub-state over the same indices"""
    rng = np.random.default_rng(N)
    model = g.models.lgssm2(); ys = g.models.simulate(model, 6)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=7, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, 7, keep_prev=True).initialize(ys[0])
    parts, half = _index_sets(N, rng)
    kw = dict(sort_particles=True) if method == "stratified" else {}
    for k, ix in enumerate(parts + [half]):
        sv, ov = st[ix], orc[ix]
        assert sv.n_particles == ix.size
        g.pf_update(sv, (2,), (None,), ys[1 + k % 3]); ov.update(ys[1 + k % 3])
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw), (k, "update")
        assert g.get_ess(sv) == ov.effective_sample_size() and g.get_lml_est(sv) == ov.log_ml_estimate()
        alpha = 0.5 if k == 1 else None
        g.pf_resample(sv, method, priority_fn=g.Tempering(alpha) if alpha else None, check=False, **kw)
        ov.resample(method, priority_alpha=alpha, check=False, **kw)
        assert np.array_equal(sv.parents, ov.parents), (k, "parents")
        g.pf_rejuvenate(sv, None, (), 1, method="move" if k % 2 else "reweight"); ov.rejuvenate("move" if k % 2 else "reweight", 1)
        assert np.array_equal(sv.traces, ov.rows), (k, "rows")
        np.testing.assert_allclose(sv.log_weights, ov.lw, rtol=1e-12, atol=1e-12)
        orc.lw[ix] = sv.log_weights                                   # (tempered weights: 1e-6 bar; keep the two in lockstep)
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.parents, orc.parents), (k, "source")
    # a boolean mask is an index vector too
    mask = rng.random(N) < 0.4
    sv, ov = st[mask], orc[np.flatnonzero(mask)]
    g.pf_resample(sv, method, check=False, **kw); ov.resample(method, check=False, **kw)
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.parents, orc.parents)
    # the filter keeps working as a whole
    g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
    g.pf_update(st, (5,), (None,), ys[5]); orc.update(ys[5])
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.parents, orc.parents)
    st.close()


@pytest.mark.gpu
def test_hip_index_view_errors(g, o):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    st = g.pf_initialize(model, (1,), ys[0], 100, seed=1)
    for bad in ([3, 5, 3], [0, 100], [-1, 2], []):                    # repeated / out of bounds / empty
        with pytest.raises(g.ErrorException):
            st[bad]
    with pytest.raises(g.ErrorException):
        st[np.zeros(7, bool)]                                         # a mask of the wrong length
    v = st[[5, 1, 99]]
    assert np.array_equal(v.traces, st.traces[[5, 1, 99]]) and np.array_equal(v.log_weights, st.log_weights[[5, 1, 99]])
    st.close()


@pytest.mark.gpu
def test_view_outliving_its_filter_fails_loudly(g):
    """gpf_destroy on a filter whose view handles are still alive (state.close() with a sub-state in hand; a host that frees the filter first): the views
    become orphans -- every call on them raises "stale view", destroying them afterwards is safe (found by the random API sequences: a segfault in the
    view's destructor after a checkpoint moved the run to a fresh handle)"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    st = g.pf_initialize(model, (1,), ys[0], 20_000, seed=4)
    v1, v2, v3 = st[100:5000], st[0:20_000:7], st[np.array([5, 17, 19_999, 3])]
    g.pf_resample(v1, "multinomial", check=False); g.pf_resample(v2, "residual", check=False); g.pf_update(v3, (2,), (None,), ys[1])
    st.close()
    for v in (v1, v2, v3):
        with pytest.raises(g.ErrorException, match="stale view"):
            g.get_ess(v)
        with pytest.raises(g.ErrorException, match="stale view"):
            g.pf_resample(v, "multinomial", check=False)
    other = g.pf_initialize(model, (1,), ys[0], 20_000, seed=4)      # (recycles the freed memory and stream)
    g.pf_update(other, (2,), (None,), ys[1])
    v1.close(); v2.close(); v3.close()
    assert np.isfinite(g.get_lml_est(other))
