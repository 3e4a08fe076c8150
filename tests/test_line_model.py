"""The reference's OWN fixture model -- line_model, /root/reference/test/runtests.jl:3-16 -- as the fifth native model, with
the closed-form assertions of the reference's tests restated with THE REFERENCE'S numbers (each test names the lines it
restates).  Oracle tests run on the CPU; every `-m gpu` twin drives the same scenario through the C ABI and compares the
device state with the oracle bit for bit.

    slope ~ uniform_discrete(-2, 2);  step t:  x = t,  outlier ~ bernoulli(0.1),  y ~ normal(x * slope, outlier ? 10. : 1.)

Differences of form, not of arithmetic: a native model takes one step per pf_update (model args (10,) = ten updates; the
log-weights add up exactly as Gen's `generate` weight does), and the per-step data vector carries the step index
(`g.models.line_obs(t, slope)` = line_choicemap of test/runtests.jl:22-23 for step t)."""
import math

import numpy as np
import pytest

N = 100                                                     # every reference test uses 100 particles


def logpdf_normal(x, mu, sigma):                            # Gen: logpdf(normal, x, mu, sigma)
    return -0.5 * ((x - mu) / sigma) ** 2 - math.log(sigma) - 0.5 * math.log(2 * math.pi)


def logpdf_bernoulli(v, p):                                 # Gen: logpdf(bernoulli, v, p)
    return math.log(p) if v else math.log(1 - p)


class OracleDriver:
    """the oracle behind the few calls the scenarios need"""

    def __init__(self, g, o, seed=11, keep_prev=False):
        self.g, self.m = g, g.models.line_model()
        self.f = o.OracleFilter(self.m.model_id, self.m.params, N, seed, keep_prev=keep_prev)

    def init(self, t, slope=0.0, proposal=False, strata=None, layout="contiguous"):
        self.f.initialize(self.g.models.line_obs(t, slope), proposal=proposal, strata=strata, layout=layout); return self

    def update(self, t, slope=0.0, proposal=False, strata=None, layout="interleaved"):
        self.f.update(self.g.models.line_obs(t, slope), proposal=proposal, strata=strata, layout=layout); return self

    def rejuvenate(self, method, q=None):
        import math
        self.f.rejuvenate(method, 1, proposal=None if q is None else (q, math.log(q), math.log1p(-q))); return self

    def view(self, sl):                                      # state[idxs] (src/view.jl:35-48): traces and weights of a sub-state
        v = self.f[sl]; return np.array(v.rows), np.array(v.lw)

    rows = property(lambda s: s.f.rows)
    lw = property(lambda s: s.f.lw)
    n_accepted = property(lambda s: s.f.n_accepted)
    mean = lambda s, c: s.f.mean(c)
    var = lambda s, c: s.f.var(c)


class DeviceDriver:
    """the same calls through the host mirror of the reference's operators -> C ABI -> HIP kernels"""

    def __init__(self, g, o, seed=11, keep_prev=False):
        self.g, self.m, self.seed, self.keep_prev, self.st = g, g.models.line_model(), seed, keep_prev, None

    def init(self, t, slope=0.0, proposal=False, strata=None, layout="contiguous"):
        g, obs = self.g, self.g.models.line_obs(t, slope)
        kw = dict(seed=self.seed, keep_prev=self.keep_prev)
        if strata is not None and proposal:
            self.st = g.pf_initialize(self.m, (t,), obs, [{"slope": v} for v in strata], g.line_fixed, ([1],), N, layout=layout, **kw)
        elif strata is not None:
            self.st = g.pf_initialize(self.m, (t,), obs, [{"slope": v} for v in strata], N, layout=layout, **kw)
        elif proposal:
            self.st = g.pf_initialize(self.m, (t,), obs, g.line_fixed, (0,), N, **kw)
        else:
            self.st = g.pf_initialize(self.m, (t,), obs, N, **kw)
        return self

    def update(self, t, slope=0.0, proposal=False, strata=None, layout="interleaved"):
        g, obs = self.g, self.g.models.line_obs(t, slope)
        if strata is not None:
            g.pf_update(self.st, (t,), (None,), obs, [{"outlier": v} for v in strata], layout=layout)
        elif proposal:
            g.pf_update(self.st, (t,), (None,), obs, g.line_fixed, (range(1, t + 1),))
        else:
            g.pf_update(self.st, (t,), (None,), obs)
        return self

    def rejuvenate(self, method, q=None):
        if q is not None and method == "move":               # mh(trace, outlier_propose, (idx,)) under pf_move_accept!
            self.g.pf_rejuvenate(self.st, self.g.mh, (self.g.outlier_propose(q), (1,)), 1, method="move", count=True)
            return self
        if q is not None:                                    # move_reweight(trace, outlier_propose, (idx,)) with outlier ~ bernoulli(q)
            self.g.pf_rejuvenate(self.st, self.g.move_reweight, (self.g.outlier_propose(q), (1,)), 1, method="reweight")
            return self
        self._acc = self.g.pf_rejuvenate(self.st, self.g.mh if method == "move" else self.g.move_reweight, (), 1, method=method, count=True)
        return self

    def view(self, sl):                                      # state[idxs] through gpf_view_create_strided
        v = self.st[sl]; return v.traces, v.log_weights

    rows = property(lambda s: s.st.traces)
    lw = property(lambda s: s.st.log_weights)
    n_accepted = property(lambda s: s.st.n_accepted)
    mean = lambda s, c: s.g.mean(s.st, c)
    var = lambda s, c: s.g.var(s.st, c)


DRIVERS = [pytest.param(OracleDriver, id="oracle"), pytest.param(DeviceDriver, id="hip", marks=pytest.mark.gpu)]


# ------------------------------------------------------------------------------------------ test/initialize.jl
@pytest.mark.parametrize("D", DRIVERS)
def test_initialize_default_proposal(g, o, D):
    """test/initialize.jl:4-10: slopes in -2..2, weights == 0 without observations; with observations the weight is the
    `generate` weight log p(y_1..n = 0 | slope, outliers)."""
    d = D(g, o).init(0)
    assert np.all((d.rows[:, 0] >= -2) & (d.rows[:, 0] <= 2)) and set(np.unique(d.rows[:, 0])) <= {-2., -1., 0., 1., 2.}   # :5
    assert np.all(d.lw == 0.0)                                                                     # :6  w ≈ 0
    d = D(g, o).init(1)                                                                             # :7  (1,), line_choicemap(1)
    exp = [logpdf_normal(0.0, 1 * s, 10.0 if out else 1.0) for s, out in d.rows[:, :2]]
    np.testing.assert_allclose(d.lw, exp, rtol=1e-12, atol=1e-12)
    d = D(g, o).init(1)                                                                             # :9  (10,), line_choicemap(10)
    total = d.lw.copy()
    for t in range(2, 11):
        d.update(t)
    assert len(set(np.unique(d.rows[:, 0])) - {-2., -1., 0., 1., 2.}) == 0
    assert np.all(np.isfinite(d.lw)) and np.all(d.lw <= total + 1e-12)       # every further observation of y = 0 costs density < 1


@pytest.mark.parametrize("D", DRIVERS)
def test_initialize_custom_proposal(g, o, D):
    """test/initialize.jl:21-31: proposal slope ~ uniform_discrete(0, 0) -> slope == 0 and w ≈ log(1/5); with the outlier
    proposal bernoulli(0.0) the outlier is false and the proposed choices' model scores enter the weight."""
    d = D(g, o).init(0, proposal=True)
    assert np.all(d.rows[:, 0] == 0.0)                                                             # :22
    np.testing.assert_allclose(d.lw, math.log(1 / 5), rtol=1e-15)                                  # :23  w ≈ log(1/5)
    d = D(g, o).init(1, proposal=True)
    assert np.all(d.rows[:, 1] == 0.0)                                                             # :26  outlier == false
    np.testing.assert_allclose(d.lw, math.log(1 / 5) + logpdf_bernoulli(False, 0.1) + logpdf_normal(0.0, 0.0, 1.0), rtol=1e-14)


@pytest.mark.parametrize("D", DRIVERS)
@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_initialize_with_stratification(g, o, D, layout):
    """test/initialize.jl:39-64: strata = the five slopes; contiguous blocks of 20 (:47-52) or interleaved k:5:100 (:58-63);
    weights ≈ 0 without observations (:45,:56).  With y_1 = 0 observed the weight is log p(slope) + log 5 + logpdf(y) =
    logpdf(normal, 0, slope, std) -- the stratum's log-probability and log(n_strata) cancel (initialize.jl:103-105)."""
    slopes = [-2., -1., 0., 1., 2.]
    d = D(g, o).init(0, strata=slopes, layout=layout)
    np.testing.assert_allclose(d.lw, 0.0, atol=1e-15)                                              # :45 / :56
    d = D(g, o).init(1, strata=slopes, layout=layout)
    for k, slope in enumerate(slopes):
        sel = slice(20 * k, 20 * k + 20) if layout == "contiguous" else slice(k, N, 5)            # state[(k-20+1):k] / state[k:5:100]
        vrows, vlw = d.view(sel)                                                                   # get_traces(state[...]), :48 / :59
        assert vrows.shape[0] == 20 and np.array_equal(vrows, d.rows[sel]) and np.array_equal(vlw, d.lw[sel])
        assert np.all(vrows[:, 0] == slope)                                                        # :49 / :60
        exp = [logpdf_normal(0.0, slope, 10.0 if out else 1.0) for out in vrows[:, 1]]
        np.testing.assert_allclose(vlw, exp, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("D", DRIVERS)
@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_initialize_with_stratification_and_custom_proposal(g, o, D, layout):
    """test/initialize.jl:66-90: strata = the five slopes, outlier_propose = bernoulli(0.0): per stratum slope == the stratum's,
    outlier == false, expected_w = logpdf(bernoulli, false, 0.1) + logpdf(normal, 0.0, slope, 1.0)  (:77-79 / :88-90);
    contiguous blocks state[(k-20+1):k] (:73) or interleaved state[k:5:100] (:84)"""
    slopes = [-2., -1., 0., 1., 2.]
    d = D(g, o).init(1, strata=slopes, proposal=True, layout=layout)
    for k, slope in enumerate(slopes):
        sel = slice(20 * k, 20 * k + 20) if layout == "contiguous" else slice(k, N, 5)
        vrows, vlw = d.view(sel)
        assert np.all(vrows[:, 0] == slope) and np.all(vrows[:, 1] == 0.0)                         # :74-75 / :85-86
        np.testing.assert_allclose(vlw, logpdf_bernoulli(False, 0.1) + logpdf_normal(0.0, slope, 1.0), rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------------------------------ test/update.jl
@pytest.mark.parametrize("D", DRIVERS)
def test_update_default_proposal(g, o, D):
    """test/update.jl:5-10: expected_ws = logpdf(normal, 0.0, tr[:slope], o ? 10.0 : 1.0)."""
    d = D(g, o).init(0).update(1)
    exp = [logpdf_normal(0.0, s, 10.0 if out else 1.0) for s, out in d.rows[:, :2]]               # :8-9
    np.testing.assert_allclose(d.lw, exp, rtol=1e-12, atol=1e-12)                                  # :10
    assert 0 < (d.rows[:, 1] != 0).sum() < N // 2                                                  # outlier ~ bernoulli(0.1)


@pytest.mark.parametrize("D", DRIVERS)
@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_update_with_stratification(g, o, D, layout):
    """test/update.jl:13-40: strata = outlier false / true; expected_ws = logpdf(bernoulli, val, 0.1) + log(2) +
    logpdf(normal, 0.0, tr[:slope], std)."""
    d = D(g, o).init(0).update(1, strata=[0.0, 1.0], layout=layout)
    for k, val in enumerate([False, True]):
        sel = slice(50 * k, 50 * k + 50) if layout == "contiguous" else slice(k, N, 2)            # :20 state[(50k-49):50k] / :33 state[k:2:100]
        vrows, vlw = d.view(sel)
        assert vrows.shape[0] == 50 and np.array_equal(vrows, d.rows[sel]) and np.array_equal(vlw, d.lw[sel])
        assert np.all((vrows[:, 1] != 0) == val)                                                   # :21 / :34
        std = 10.0 if val else 1.0
        exp = [logpdf_bernoulli(val, 0.1) + math.log(2) + logpdf_normal(0.0, s, std) for s in vrows[:, 0]]      # :23-24 / :36-37
        np.testing.assert_allclose(vlw, exp, rtol=1e-12, atol=1e-12)                               # :25 / :38


@pytest.mark.parametrize("D", DRIVERS)
def test_update_custom_proposal(g, o, D):
    """test/update.jl:47-55: ten steps with outlier_propose (bernoulli(0.0)): every outlier false, every weight != 0, and
    in closed form w = sum_t [log p(outlier_t = false) + logpdf(normal, 0, t slope, 1)]."""
    d = D(g, o).init(0)
    slopes = d.rows[:, 0].copy()
    for t in range(1, 11):
        d.update(t, proposal=True)
        assert np.all(d.rows[:, 1] == 0.0)                                                         # :53
    assert np.all(d.lw != 0.0)                                                                     # :54
    exp = [sum(logpdf_bernoulli(False, 0.1) + logpdf_normal(0.0, t * s, 1.0) for t in range(1, 11)) for s in slopes]
    np.testing.assert_allclose(d.lw, exp, rtol=1e-11)


# ------------------------------------------------------------------------------------------ test/rejuvenate.jl
@pytest.mark.parametrize("D", DRIVERS)
def test_move_reweight_kernel(g, o, D):
    """test/rejuvenate.jl:3-17, selection variant on :line => 1 => :outlier: the relative weight is
    logpdf(normal, 0, slope, out_new ? 10. : 1.) - logpdf(normal, 0, slope, out_old ? 10. : 1.)  (:10-13,:15)."""
    d = D(g, o, keep_prev=True).init(0).update(1)
    slope, out_old, lw_old = d.rows[:, 0].copy(), d.rows[:, 1] != 0, d.lw.copy()
    d.rejuvenate("reweight")
    out_new = d.rows[:, 1] != 0
    exp = [logpdf_normal(0, s, 10. if n else 1.) - logpdf_normal(0, s, 10. if ol else 1.) for s, ol, n in zip(slope, out_old, out_new)]
    np.testing.assert_allclose(d.lw - lw_old, exp, rtol=1e-9, atol=1e-12)
    assert np.array_equal(d.rows[:, 0], slope) and (out_old != out_new).any()


@pytest.mark.parametrize("D", DRIVERS)
def test_move_reweight_kernel_proposal_variant(g, o, D):
    """test/rejuvenate.jl:19-27, proposal variant with outlier_propose = {:line => idx => :outlier} ~ bernoulli(0.9):
        expected_w = logpdf(bernoulli, out_new, 0.1) - logpdf(bernoulli, out_old, 0.1)
                   + logpdf(normal, 0, slope, out_new ? 10. : 1.) - logpdf(normal, 0, slope, out_old ? 10. : 1.)
                   + (out_old == out_new ? 0.0 : logpdf(bernoulli, out_old, 0.9) - logpdf(bernoulli, out_old, 0.1))     (:20-25)"""
    d = D(g, o, keep_prev=True).init(0).update(1)
    slope, out_old, lw_old = d.rows[:, 0].copy(), d.rows[:, 1] != 0, d.lw.copy()
    d.rejuvenate("reweight", q=0.9)
    out_new = d.rows[:, 1] != 0
    exp = [logpdf_bernoulli(n, 0.1) - logpdf_bernoulli(ol, 0.1) + logpdf_normal(0, s, 10. if n else 1.) - logpdf_normal(0, s, 10. if ol else 1.)
           + (0.0 if ol == n else logpdf_bernoulli(ol, 0.9) - logpdf_bernoulli(ol, 0.1)) for s, ol, n in zip(slope, out_old, out_new)]
    np.testing.assert_allclose(d.lw - lw_old, exp, rtol=1e-9, atol=1e-12)                          # :27
    assert np.array_equal(d.rows[:, 0], slope)
    assert 0.75 * N < out_new.sum() <= N and (out_old != out_new).any()                            # proposed from bernoulli(0.9)


@pytest.mark.parametrize("D", DRIVERS)
def test_move_accept_rejuvenation(g, o, D):
    """test/rejuvenate.jl:30-50: only accepted particles change, the weights do not (rejuvenate.jl:40-53); here the kernel is
    mh on the step's outlier, accepted iff log(rand) < the likelihood ratio above."""
    d = D(g, o, keep_prev=True).init(0).update(1)
    rows_old, lw_old = d.rows.copy(), d.lw.copy()
    d.rejuvenate("move")
    changed = np.any(d.rows[:, :2] != rows_old[:, :2], axis=1)
    assert np.array_equal(d.lw, lw_old)
    assert changed.sum() <= d.n_accepted <= N and np.array_equal(d.rows[:, 0], rows_old[:, 0])
    assert np.all((rows_old[changed, 1] != 0) | (np.abs(rows_old[changed, 0]) < 3))              # (moves to an outlier are accepted only sometimes)


@pytest.mark.parametrize("D", DRIVERS)
def test_move_accept_with_a_proposal_kernel(g, o, D):
    """pf_move_accept!(state, mh, (outlier_propose, (idx,))) -- Gen.mh(trace, proposal, proposal_args) through src/rejuvenate.jl:40-53, with the
    proposal of test/rejuvenate.jl:19-27 (outlier ~ bernoulli(q)).  The weights do not change (:30-50); a particle whose proposal equals its
    current value has alpha = 0 and always "accepts" (log(rand()) < 0); a changed particle was accepted with
        alpha = [logpdf(bernoulli, new, 0.1) + logpdf(normal, 0, slope, new ? 10 : 1)] - [the same at old] - logpdf(bernoulli, new, q) + logpdf(bernoulli, old, q)
    (test/rejuvenate.jl:20-25's expected_w), so changes with alpha > 0 all happen, and the fraction of false -> true moves at slope 0 matches
    q * min(1, exp(alpha)) -- the closed form of the MH kernel."""
    import math
    q = 0.5
    d = D(g, o, keep_prev=True).init(0).update(1)
    rows_old, lw_old = d.rows.copy(), d.lw.copy()
    d.rejuvenate("move", q=q)
    assert np.array_equal(d.lw, lw_old) and np.array_equal(d.rows[:, 0], rows_old[:, 0])          # weights and the untouched choice stay
    old, new, slope = rows_old[:, 1] != 0, d.rows[:, 1] != 0, rows_old[:, 0]
    changed = old != new
    assert changed.any() and changed.sum() <= d.n_accepted <= N



def test_mh_proposal_acceptance_rates_closed_form(g, o):
    """the same kernel on 40 000 particles (oracle; the device equals it bit for bit, test_hip_line_model_bitexact[move_proposal]): per
    (slope, old outlier) class the fraction of particles that changed is P(propose the other value) * min(1, exp(alpha)) with alpha =
    test/rejuvenate.jl:20-25's expected_w -- the Metropolis-Hastings acceptance rule of Gen.mh(trace, proposal, proposal_args)"""
    q, n = 0.5, 40_000
    m = g.models.line_model()
    f = o.OracleFilter(m.model_id, m.params, n, 11, keep_prev=True)
    f.initialize(g.models.line_obs(0, 0.0)); f.update(g.models.line_obs(1, 0.0))
    rows_old, lw_old = f.rows.copy(), f.lw.copy()
    f.rejuvenate("move", 1, proposal=(q, math.log(q), math.log1p(-q)))
    assert np.array_equal(f.lw, lw_old)
    old, new, slope = rows_old[:, 1] != 0, f.rows[:, 1] != 0, rows_old[:, 0]

    def alpha(s, ol, nw):                                    # y_1 = 0 (line_obs(1, 0.0)): logpdf(normal, 0, slope * 1, sd)
        return (logpdf_bernoulli(nw, 0.1) + logpdf_normal(0, s, 10. if nw else 1.)) - (logpdf_bernoulli(ol, 0.1) + logpdf_normal(0, s, 10. if ol else 1.)) \
            - logpdf_bernoulli(nw, q) + logpdf_bernoulli(ol, q)
    checked = 0
    for sv in (-2.0, -1.0, 0.0, 1.0, 2.0):
        for ol in (False, True):
            sel = (slope == sv) & (old == ol)
            cnt = int(sel.sum())
            if cnt < 300:
                continue
            p = (q if not ol else 1 - q) * min(1.0, math.exp(alpha(sv, ol, not ol)))
            k = int((new[sel] != ol).sum())
            assert abs(k - cnt * p) <= 5 * math.sqrt(cnt * p * (1 - p)) + 2, (sv, ol, k, cnt, p)
            checked += 1
    assert checked >= 6


# ------------------------------------------------------------------------------------------ test/statistics.jl
@pytest.mark.parametrize("D", DRIVERS)
def test_statistics_on_a_degenerate_state(g, o, D):
    """test/statistics.jl:10-18 use a model whose choices are constants (x ≡ 1): mean == the constant, var ≈ 0 atol 1e-6.
    The same with line_model's slope under the fixed proposal (slope ≡ 0) and after an all-false outlier step."""
    d = D(g, o).init(0, proposal=True).update(1, proposal=True)
    assert d.mean(0) == 0.0 and abs(d.var(0)) <= 1e-6                                              # :12, :17
    assert d.mean(1) == 0.0 and abs(d.var(1)) <= 1e-6


# ------------------------------------------------------------------------------------------ device == oracle, bit for bit
@pytest.mark.gpu
@pytest.mark.parametrize("scenario", ["default", "proposal", "strata_c", "strata_i", "reweight", "move", "reweight_proposal", "move_proposal", "strata_proposal"])
def test_hip_line_model_bitexact(g, o, scenario):
    a, b = OracleDriver(g, o, seed=5, keep_prev=True), DeviceDriver(g, o, seed=5, keep_prev=True)
    for d in (a, b):
        if scenario == "proposal":
            d.init(0, proposal=True)
            for t in range(1, 4):
                d.update(t, proposal=True)
        elif scenario == "strata_proposal":
            d.init(1, strata=[-2., -1., 0., 1., 2.], proposal=True, layout="interleaved").update(2)
        elif scenario.startswith("strata"):
            lay = "contiguous" if scenario.endswith("c") else "interleaved"
            d.init(1, strata=[-2., -1., 0., 1., 2.], layout=lay).update(2, strata=[0.0, 1.0], layout=lay)
        else:
            d.init(0).update(1).update(2, slope=1.0)
            if scenario in ("reweight", "move"):
                d.rejuvenate(scenario)
            if scenario == "reweight_proposal":
                d.rejuvenate("reweight", q=0.9)
            if scenario == "move_proposal":
                d.rejuvenate("move", q=0.6)
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.lw, b.lw)
    if scenario in ("move", "move_proposal"):
        assert a.n_accepted == b.n_accepted
    st, f = b.st, a.f
    assert g.get_ess(st) == f.effective_sample_size() and g.get_lml_est(st) == f.log_ml_estimate()
    for method in ("multinomial", "residual", "stratified"):
        g.pf_resample(st, method, check=False); f.resample(method, check=False)
        assert np.array_equal(st.parents, f.parents) and np.array_equal(st.traces, f.rows)
        g.pf_update(st, (3,), (None,), g.models.line_obs(3, 0.5)); f.update(g.models.line_obs(3, 0.5))
        assert np.array_equal(st.log_weights, f.lw)
