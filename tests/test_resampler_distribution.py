"""What can be pinned about the RANDOM part of the resamplers without a reference run (the reference holds no seeds and no golden
vectors): their distribution.

  * offspring counts are unbiased, E[#children of i] = N w_i, for every resampler (multinomial resample.jl:59, residual :96-115,
    stratified :155-170, the opt-in multinomial_sorted, and pf_residual_resize! / pf_multinomial_resize! resize.jl:46-124);
  * for the two multinomial forms the counts are Multinomial(N, w): variance N w_i (1 - w_i) as well;
  * the log-ML estimate of a bootstrap filter on the linear-Gaussian model converges to the exact Kalman value with EVERY resampler.

Monte-Carlo over >= 300 seeds on the oracle (CPU); once through the HIP kernels (-m gpu)."""
import math

import numpy as np
import pytest

N, R = 48, 400
ALL = ["multinomial", "multinomial_sorted", "residual", "stratified", "stratified_sorted"]


def weights(kind: str) -> np.ndarray:
    rng = np.random.default_rng(12345)
    if kind == "spread":
        lw = rng.normal(0.0, 1.2, N)
    elif kind == "skewed":                      # one heavy particle, a light tail (residual: a deterministic head AND a random tail)
        lw = rng.normal(0.0, 0.5, N); lw[7] += 3.0; lw[30:] -= 4.0
    else:
        raise ValueError(kind)
    return lw


def norm_w(lw):
    w = np.exp(lw - lw.max())
    return w / w.sum()


def counts_of(parents, n_in):
    return np.bincount(np.asarray(parents) - 1, minlength=n_in).astype(np.float64)


def oracle_counts(g, o, lw, method, seeds, n_out=None):
    m = g.models.lgssm2()
    y = np.zeros(2)
    out = np.empty((len(seeds), lw.size))
    for r, seed in enumerate(seeds):
        f = o.OracleFilter(m.model_id, m.params, lw.size, seed).initialize(y)
        f.lw = lw.copy()
        if n_out is not None:
            f.resize(n_out, method, check=False)
        elif method == "stratified_sorted":
            f.resample("stratified", sort_particles=True, check=False)
        elif method == "stratified":
            f.resample("stratified", sort_particles=False, check=False)
        else:
            f.resample(method, check=False)
        out[r] = counts_of(f.parents, lw.size)
    return out


def check_unbiased(cnt, expect, tag):
    """|mean over the runs - N w_i| within 4.5 standard errors (sample standard deviation) for every particle"""
    Rr = cnt.shape[0]
    mean, sd = cnt.mean(axis=0), cnt.std(axis=0, ddof=1)
    # (a particle with N w_i << 1 may have no child in any run: its sample deviation is 0 although the count is Bernoulli-like; every
    # resampler here has at least the variance of rounding N w_i up or down at random, frac (1 - frac))
    fr = expect - np.floor(expect)
    sd = np.maximum(sd, np.sqrt(fr * (1.0 - fr)))
    tol = 4.5 * sd / math.sqrt(Rr) + 1e-9
    bad = np.abs(mean - expect) > tol
    assert not bad.any(), (tag, np.flatnonzero(bad), mean[bad], expect[bad], tol[bad])


def check_multinomial_variance(cnt, n_out, w, tag):
    """the counts of one particle are Binomial(n, w_i): sample variance against n p q with the standard error of a sample variance,
    sqrt((mu4 - sigma^4) / R), mu4 = n p q (1 + 3 (n - 2) p q)"""
    Rr = cnt.shape[0]
    pq = w * (1.0 - w)
    var = n_out * pq
    mu4 = n_out * pq * (1.0 + 3.0 * (n_out - 2) * pq)
    se = np.sqrt(np.maximum(mu4 - var ** 2, 0.0) / Rr)
    s2 = cnt.var(axis=0, ddof=1)
    bad = np.abs(s2 - var) > 4.5 * se + 1e-9
    assert not bad.any(), (tag, np.flatnonzero(bad), s2[bad], var[bad], se[bad])
    # and two particles' counts are negatively correlated as in a multinomial: cov = -n w_i w_j (the heaviest pair)
    i, j = np.argsort(w)[-2:]
    cov = np.cov(cnt[:, i], cnt[:, j])[0, 1]
    want = -n_out * w[i] * w[j]
    assert abs(cov - want) < 6.0 * math.sqrt(var[i] * var[j] / Rr), (tag, cov, want)


@pytest.mark.parametrize("kind", ["spread", "skewed"])
@pytest.mark.parametrize("method", ALL)
def test_offspring_counts_are_unbiased_oracle(g, o, method, kind):
    lw = weights(kind)
    cnt = oracle_counts(g, o, lw, method, range(1, R + 1))
    assert (cnt.sum(axis=1) == N).all()
    w = norm_w(lw)
    check_unbiased(cnt, N * w, (method, kind))
    if method.startswith("multinomial"):
        check_multinomial_variance(cnt, N, w, (method, kind))
    if method == "residual":                    # resample.jl:99: at least floor(N w_i) copies, always (test/resample.jl:47-52)
        assert (cnt >= np.floor(N * w - 1e-9)).all()
        assert (cnt <= np.floor(N * w + 1e-9) + N).all()
    if method.startswith("stratified"):         # a stratified draw gives floor(N w_i) or ceil(N w_i) (+-1 at the strata edges)
        assert (np.abs(cnt - N * w) < 2.0).all()


def test_sorted_and_iid_multinomial_have_the_same_law_oracle(g, o):
    """the opt-in sorted form changes the ORDER of the ancestors only: per-particle count distributions of the two forms agree
    (two-sample chi-square on the pooled count histograms of the heaviest particles)"""
    lw = weights("spread")
    a = oracle_counts(g, o, lw, "multinomial", range(1, R + 1))
    b = oracle_counts(g, o, lw, "multinomial_sorted", range(1001, 1001 + R))
    for i in np.argsort(lw)[-6:]:
        top = int(max(a[:, i].max(), b[:, i].max())) + 1
        ha = np.bincount(a[:, i].astype(int), minlength=top + 1).astype(float)
        hb = np.bincount(b[:, i].astype(int), minlength=top + 1).astype(float)
        keep = (ha + hb) >= 10
        chi2 = (((ha - hb) ** 2) / np.maximum(ha + hb, 1.0))[keep].sum()
        dof = max(int(keep.sum()) - 1, 1)
        assert chi2 < dof + 5.0 * math.sqrt(2.0 * dof) + 5.0, (i, chi2, dof)
    # the sorted form emits non-decreasing ancestors, the i.i.d. form (almost surely) does not
    m = g.models.lgssm2()
    f = o.OracleFilter(m.model_id, m.params, N, 5).initialize(np.zeros(2)); f.lw = lw.copy()
    f.resample("multinomial_sorted", check=False)
    assert (np.diff(f.parents) >= 0).all()
    f = o.OracleFilter(m.model_id, m.params, N, 5).initialize(np.zeros(2)); f.lw = lw.copy()
    f.resample("multinomial", check=False)
    assert (np.diff(f.parents) < 0).any()


@pytest.mark.parametrize("method", ["multinomial", "residual"])
@pytest.mark.parametrize("n_out", [31, 48, 80])
def test_resize_offspring_counts_are_unbiased_oracle(g, o, method, n_out):
    """pf_multinomial_resize! / pf_residual_resize! (resize.jl:46-124): E[#children of i] = n_out w_i"""
    lw = weights("skewed")
    cnt = oracle_counts(g, o, lw, method, range(1, R + 1), n_out=n_out)
    assert (cnt.sum(axis=1) == n_out).all()
    w = norm_w(lw)
    check_unbiased(cnt, n_out * w, (method, n_out))
    if method == "multinomial":
        check_multinomial_variance(cnt, n_out, w, (method, n_out))
    else:
        assert (cnt >= np.floor(n_out * w - 1e-9)).all()


@pytest.mark.parametrize("method", ["multinomial", "multinomial_sorted", "residual", "stratified"])
def test_log_ml_matches_kalman_with_every_resampler_oracle(g, o, method):
    """known answer: the exact Kalman log-likelihood of the LG-SSM; the particle estimate is within Monte-Carlo error for each
    resampler (test_oracle_reference_invariants.py has the stratified case alone)"""
    m = g.models.lgssm2()
    T = 25
    ys = g.models.simulate(m, T)
    exact = g.models.kalman_loglik(m, ys)
    est = []
    for seed in range(1, 5):
        f = o.OracleFilter(m.model_id, m.params, 20000, seed).initialize(ys[0])
        for t in range(1, T):
            kw = dict(sort_particles=False) if method == "stratified" else {}
            f.resample(method, check=False, **kw)
            f.update(ys[t])
        est.append(f.log_ml_estimate())
    assert abs(np.mean(est) - exact) < 0.08, (method, est, exact)


# ----------------------------------------------------------------------------------------------- once through the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("method", ALL)
def test_offspring_counts_are_unbiased_gpu(g, o, method):
    """the same Monte-Carlo check on the device path: ONE filter, its log-weights reset before every resample (each resample is a
    new epoch, i.e. a new random stream)"""
    lw = weights("skewed")
    m = g.models.lgssm2()
    st = g.pf_initialize(m, (1,), np.zeros(2), N, seed=77)
    Rg = 300
    cnt = np.empty((Rg, N))
    for r in range(Rg):
        st.log_weights = lw
        if method == "stratified_sorted":
            g.pf_resample(st, "stratified", sort_particles=True, check=False)
        elif method == "stratified":
            g.pf_resample(st, "stratified", sort_particles=False, check=False)
        else:
            g.pf_resample(st, method, check=False)
        cnt[r] = counts_of(st.parents, N)
    st.close()
    w = norm_w(lw)
    check_unbiased(cnt, N * w, ("gpu", method))
    if method.startswith("multinomial"):
        check_multinomial_variance(cnt, N, w, ("gpu", method))


@pytest.mark.gpu
def test_resize_offspring_counts_are_unbiased_gpu(g, o):
    lw = weights("skewed")
    m = g.models.lgssm2()
    n_out, Rg = 80, 200
    cnt = np.empty((Rg, N))
    for r in range(Rg):
        st = g.pf_initialize(m, (1,), np.zeros(2), N, seed=1000 + r)
        st.log_weights = lw
        g.pf_residual_resize(st, n_out, check=False)
        cnt[r] = counts_of(st.parents, N)
        st.close()
    check_unbiased(cnt, n_out * norm_w(lw), "gpu residual resize")
