"""The opt-in sorted form of the multinomial resampler (gpf.h GPF_RESAMPLE_MULTINOMIAL_SORTED): pf_multinomial_resample!
(src/resample.jl:48-65) with its N uniforms drawn already sorted -- uniform spacings in exact integers.

  not gpu: the integer arithmetic (reciprocal division of the kernels against Python integers; the oracle's targets against a plain-Python
           restatement), monotonicity, the reference's resample invariants (test/resample.jl:11-12,26-31);
  gpu:     HIP == oracle bit for bit (sizes around every tile boundary, skewed weights = the wide regimes of the merge, priorities, views,
           invalid weights), and the log-ML invariance at N = 10^6.
Distributional tests (offspring counts ~ Multinomial(N, w)): tests/test_resampler_distribution.py."""
import math

import numpy as np
import pytest


# ----------------------------------------------------------------------------------------------- integer arithmetic (no GPU)
def test_reciprocal_division_is_exact(g):
    """gpf_host_div128 runs the kernels' code (gpf_math.hpp div128_setup / div128): floor(P 2^64 / den) for P < den < 2^63 -- where a
    tile starts among the sorted uniforms"""
    L = g._lib.load()
    rng = np.random.default_rng(3)
    cases = [(0, 1), (1, 2), (1, 3), (2, 3), ((1 << 62) - 1, 1 << 62), (1, (1 << 62) + 12345), ((1 << 61) + 7, (1 << 62) - 1),
             (5, 7), (6, 7), (0, (1 << 63) - 1), ((1 << 63) - 2, (1 << 63) - 1)]
    for _ in range(20000):
        bits = int(rng.integers(1, 63))
        ptot = int(rng.integers(1, 1 << bits)) + 1
        p = int(rng.integers(0, ptot))
        cases.append((p, ptot))
    for p, ptot in cases:
        assert L.gpf_host_div128(p, ptot) == (p << 64) // ptot, (p, ptot)


def test_two_word_reciprocal_division_is_exact(g):
    """floor(p W / den), p < den < 2^62, W < 2^64: the slot's place inside its tile (gpf_math.hpp muldiv128)"""
    L = g._lib.load()
    rng = np.random.default_rng(4)
    cases = [(0, 5, 1), (1, 2**64 - 1, 2), (2**61, 2**64 - 1, 2**61 + 1), (12345, 0, 99999), (2**62 - 2, 2**64 - 1, 2**62 - 1)]
    for _ in range(20000):
        bits = int(rng.integers(1, 63))
        den = int(rng.integers(1, 1 << bits)) + 1
        p = int(rng.integers(0, den))
        W = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
        cases.append((p, W, den))
    for p, W, den in cases:
        assert L.gpf_host_muldiv128(p, W, den) == (p * W) // den, (p, W, den)


def _neglog_args():
    rng = np.random.default_rng(17)
    U = rng.integers(0, 2**63, 100_000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 100_000, dtype=np.uint64)
    edge = np.array([0, 1, 2**12 - 1, 2**12, 2**64 - 1, 2**64 - 2**12, 2**63, 2**63 - 1] + [(1 << b) for b in range(12, 64)]
                    + [(1 << b) - 1 for b in range(13, 64)], dtype=np.uint64)
    return np.concatenate([edge, U])


def test_spacing_logarithm_host_build_and_accuracy(g, o):
    """-log((k + 1/2) 2^-52) by table + four series terms (gpf_math.hpp neglog_u52 / gpf_oracle_math.h o_neglog_u52): the host build of
    the kernels' function == the oracle's, bit for bit; within 1e-11 of the true logarithm; never negative"""
    U = _neglog_args()
    a = U.view(np.float64)
    n = U.size
    h1, h2 = np.empty(n), np.zeros(n)
    g._lib.load().gpf_host_math(7, a.ctypes.data_as(g._lib.C.POINTER(g._lib.C.c_double)), a.ctypes.data_as(g._lib.C.POINTER(g._lib.C.c_double)), n,
                                h1.ctypes.data_as(g._lib.C.POINTER(g._lib.C.c_double)), h2.ctypes.data_as(g._lib.C.POINTER(g._lib.C.c_double)))
    o1, o2 = np.empty(n), np.zeros(n)
    o.lib().o_math_vec(7, a, a, n, o1, o2)
    assert np.array_equal(h1.view(np.uint64), o1.view(np.uint64))
    k = (U >> np.uint64(12)).astype(np.float64)
    want = -np.log((k + 0.5) * 2.0**-52)
    assert np.abs(o1 - want).max() < 1e-11 and (o1 >= 0).all()


@pytest.mark.gpu
def test_spacing_logarithm_device_bitwise(g, o):
    U = _neglog_args()
    a = U.view(np.float64).copy()
    st = g.DeviceParticleFilterState(g.models.lgssm2(), 16)
    d1, _ = st.debug_math(7, a, a)
    o1, o2 = np.empty(a.size), np.zeros(a.size)
    o.lib().o_math_vec(7, a, a, a.size, o1, o2)
    assert np.array_equal(d1.view(np.uint64), o1.view(np.uint64))


def test_tile_scale_and_gamma_variates(g, o):
    """the kernels' gamma variate (host build of gpf_math.hpp gamma_tile) == the oracle's, bit for bit; its law is Gamma(shape, 1)"""
    L = g._lib.load()
    for ntl in [1, 2, 3, 489, 512, 513, 4096, 31250, 2**20]:
        E = L.gpf_host_gamma_E(ntl)
        assert E == o.lib().o_gamma_E(ntl) == min(48, 50 - math.ceil(math.log2(ntl)) if ntl > 1 else 48)
        assert ntl * 2**12 * 2**E <= 2**62
    for shape in (1, 2, 3, 17, 2048, 2049):
        Eg = 41
        v = np.array([L.gpf_host_gamma_tile(9, gid, 3, shape, Eg) for gid in range(0, 40000, 2)], dtype=np.float64) / 2.0**Eg
        w = np.array([o.lib().o_gamma_tile_d(9, gid, 3, shape, Eg) for gid in range(0, 40000, 2)], dtype=np.float64) / 2.0**Eg
        assert np.array_equal(v, w)
        n = v.size
        assert abs(v.mean() - shape) < 4.5 * math.sqrt(shape / n)
        assert abs(v.var() - shape) < 4.5 * math.sqrt((6 * shape + 2 * shape * shape) / n)      # var of the sample variance: (mu4 - sigma^4) / n
        assert v.max() < 1.2 * shape + 60 and v.min() > 0


@pytest.mark.parametrize("n,j0", [(1, 0), (2, 0), (9, 0), (100, 0), (100, 37), (2048, 0), (2049, 5), (4100, 1)])
def test_targets_follow_the_plain_python_restatement(g, o, n, j0):
    seed, epoch, S = 11, 4, (1 << 50) + 12345
    L = o.lib()
    TILE, E = 2048, 44
    ntl = (n + TILE - 1) // TILE
    Eg = L.o_gamma_E(ntl)
    e = [int(L.o_spacing_d(seed, j0 + i, epoch)) for i in range(n + 1)]
    for i in (0, n // 2, n):                                    # the spacing itself: trunc(-log(u) 2^E) of the slot's 52-bit uniform
        u = L.o_resample_u52_d(seed, j0 + i, epoch)
        assert abs(e[i] - (-o.olog(u) * 2.0**E)) < 1e-11 * 2.0**E + 1
    G = []
    for t in range(ntl):
        cnt = min(TILE, n - t * TILE)
        G.append(int(L.o_gamma_tile_d(seed, j0 + t * TILE, epoch, cnt + (1 if t == ntl - 1 else 0), Eg)))
    gtot = sum(G) + 1
    want = []
    for t in range(ntl):
        vlo, vhi = (sum(G[:t]) << 64) // gtot, (sum(G[:t + 1]) << 64) // gtot
        tlo = (vlo * S) >> 64
        tw = ((vhi * S) >> 64) - tlo
        cnt = min(TILE, n - t * TILE)
        es = e[t * TILE:t * TILE + cnt]
        s = sum(es) + 1 + (e[n] if t == ntl - 1 else 0)
        inv_s = 1.0 / float(s)                                  # (Python floats are IEEE doubles; int -> float rounds to nearest)
        p = 0
        for k in range(cnt):
            p += es[k]
            want.append(tlo + min(tw, int((float(p) * inv_s) * float(tw))))
    T = o.targets_sorted(seed, epoch, j0, n, S)
    assert [int(t) for t in T] == want
    assert all(want[j] <= want[j + 1] for j in range(n - 1)) and want[-1] < S


def _filter(g, o, N, seed, T=3):
    m = g.models.lgssm2()
    ys = g.models.simulate(m, T)
    return m, ys, o.OracleFilter(m.model_id, m.params, N, seed).initialize(ys[0])


@pytest.mark.parametrize("N", [1, 5, 100, 3000])
def test_reference_resample_invariants_oracle(g, o, N):
    """test/resample.jl:11-12: new_traces == old_traces[parents], the log-ML estimate does not change; :26-31: all -Inf weights"""
    m, ys, f = _filter(g, o, N, 3)
    rows0, lml0 = f.rows.copy(), f.log_ml_estimate()
    f.resample("multinomial_sorted", check=False)
    assert np.array_equal(f.rows, rows0[f.parents - 1])
    assert (np.diff(f.parents) >= 0).all() and f.parents.min() >= 1 and f.parents.max() <= N
    assert abs(f.log_ml_estimate() - lml0) <= 1e-12 * max(1.0, abs(lml0))
    assert (f.lw == 0.0).all()
    f.lw[:] = -np.inf
    with pytest.raises(o.OracleError):
        f.resample("multinomial_sorted", check=True)
    f.resample("multinomial_sorted", check=False)
    assert (f.lw == 0.0).all()


def test_priorities_oracle(g, o):
    """test/resample.jl:15-23 with priority_fn = w -> w / 2"""
    m, ys, f = _filter(g, o, 2000, 8)
    lml0 = f.log_ml_estimate()
    f.resample("multinomial_sorted", priority_alpha=0.5, check=False)
    assert abs(f.log_ml_estimate() - lml0) < 1e-9 * max(1.0, abs(lml0))
    assert (np.diff(f.parents) >= 0).all()


# ----------------------------------------------------------------------------------------------- HIP == oracle
def _pair(g, o, N, seed, name="lgssm2", keep_prev=False, T=5):
    model = g.models.by_name(name)
    ys = g.models.simulate(model, T)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=keep_prev)
    orc = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=keep_prev).initialize(ys[0])
    return model, ys, st, orc


def _same(st, orc):
    return np.array_equal(st.parents, orc.parents) and np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 2, 7, 100, 2047, 2048, 2049, 4096, 4097, 50_000, 300_001, 1024 * 2048, 1024 * 2048 + 1, 2_500_000])
def test_ancestors_bitexact(g, o, N):
    """sizes around the tile boundaries (2048 slots) and around 1024 tiles (beyond: k_sorted_tiles places the tiles, below: the merge kernel)"""
    model, ys, st, orc = _pair(g, o, N, 5)
    for t in range(1, 4):
        g.pf_resample(st, "multinomial_sorted", check=False)
        orc.resample("multinomial_sorted", check=False)
        assert np.array_equal(st.parents, orc.parents), f"ancestors differ at t={t}"
        assert (np.diff(st.parents) >= 0).all()
        assert g.get_lml_est(st) == orc.log_ml_estimate()
        g.pf_update(st, (t + 1,), (None,), ys[t])
        orc.update(ys[t])
        assert _same(st, orc)
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bearings4", "sv1"])
def test_other_models_and_rejuvenation_after_it(g, o, name):
    """the deferred gather of a sorted resample feeds pf_rejuvenate! as well (k_move<GATHER>)"""
    model, ys, st, orc = _pair(g, o, 30_000, 9, name=name, keep_prev=True)
    for t in range(1, 4):
        g.pf_resample(st, "multinomial_sorted", check=False); orc.resample("multinomial_sorted", check=False)
        g.pf_rejuvenate(st, None, (), 1, method="move" if t % 2 else "reweight"); orc.rejuvenate("move" if t % 2 else "reweight", 1)
        assert _same(st, orc), (name, t)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        assert _same(st, orc), (name, t)
    st.close()


def _skewed(kind, N, rng):
    if kind == "one_heavy":
        lw = rng.normal(0, 1, N); lw[N // 3] += 40.0
    elif kind == "few_heavy":
        lw = np.full(N, -60.0); lw[rng.choice(N, 17, replace=False)] = rng.normal(0, 1, 17)
    elif kind == "front_heavy":                      # a steep CDF at the front, a flat tail: the per-16 and per-slot regimes of the merge
        lw = -np.arange(N) * (30.0 / N)
    elif kind == "back_heavy":
        lw = np.arange(N) * (30.0 / N)
    elif kind == "equal":
        lw = np.zeros(N)
    elif kind == "neg_inf_holes":
        lw = rng.normal(0, 1, N); lw[rng.random(N) < 0.7] = -np.inf
    elif kind == "last_only":
        lw = np.full(N, -np.inf); lw[-1] = 0.0
    elif kind == "first_only":
        lw = np.full(N, -np.inf); lw[0] = 0.0
    else:
        raise ValueError(kind)
    return lw


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["one_heavy", "few_heavy", "front_heavy", "back_heavy", "equal", "neg_inf_holes", "last_only", "first_only"])
@pytest.mark.parametrize("N", [5000, 200_000])
def test_skewed_weights(g, o, kind, N):
    rng = np.random.default_rng(len(kind) + N)
    model, ys, st, orc = _pair(g, o, N, 21)
    lw = _skewed(kind, N, rng)
    for rep in range(2):
        st.log_weights = lw; orc.lw = lw.copy()
        with np.errstate(invalid="ignore", divide="ignore"):
            g.pf_resample(st, "multinomial_sorted", check=False); orc.resample("multinomial_sorted", check=False)
        assert np.array_equal(st.parents, orc.parents), (kind, N, rep)
        assert _same(st, orc)
    st.close()


@pytest.mark.gpu
def test_priorities_views_and_invalid_weights(g, o):
    N = 20_000
    model, ys, st, orc = _pair(g, o, N, 13)
    g.pf_resample(st, "multinomial_sorted", priority_fn=g.Tempering(0.5), check=False)
    orc.resample("multinomial_sorted", priority_alpha=0.5, check=False)
    assert np.array_equal(st.parents, orc.parents)
    np.testing.assert_allclose(st.log_weights, orc.lw, rtol=1e-6, atol=1e-9)
    orc.lw = st.log_weights.copy()
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    # sub-state views (src/view.jl:35-48): contiguous and strided; local ancestors, slot ids start, start + 1, ...
    for sl in (slice(0, 5000), slice(5000, 20_000), slice(3, 20_000, 7)):
        v, ov = st[sl], orc[sl]
        g.pf_resample(v, "multinomial_sorted", check=False); ov.resample("multinomial_sorted", check=False)
        assert np.array_equal(v.parents, ov.parents), sl
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw), sl
    # the whole filter as a sub-state (pf_resample!(state[1:end], ...): the library's local resample, deferred gather with the kept mass)
    v, ov = st[0:N], orc[0:N]
    g.pf_resample(v, "multinomial_sorted", check=False); ov.resample("multinomial_sorted", check=False)
    assert np.array_equal(st.parents, orc.parents)
    g.pf_update(st, (3,), (None,), ys[2]); orc.update(ys[2])
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)
    # invalid weights (test/resample.jl:26-31)
    lw = np.full(N, -np.inf)
    st.log_weights = lw; orc.lw = lw.copy()
    with pytest.raises(g.ErrorException):
        g.pf_resample(st, "multinomial_sorted", check=True)
    with pytest.warns(UserWarning):
        g.pf_resample(st, "multinomial_sorted", check="warn")
    orc.resample("multinomial_sorted", check=False)
    assert np.array_equal(st.parents, orc.parents) and (st.log_weights == 0).all()
    with pytest.raises(g.ErrorException):
        g.pf_resample_blocks(st, 100, "multinomial_sorted")                  # not a block-wise method
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [3000, 2_300_000])
def test_after_an_ess_read_the_scan_is_reused(g, o, N):
    """effective_sample_size leaves the weight CDF behind; the resample that follows reuses it, so the tile totals cannot ride in a
    scan launch and take the launch of their own (k_sorted_gammas) -- the README loop's order of calls (README.md:66-72)"""
    model, ys, st, orc = _pair(g, o, N, 31)
    for t in range(1, 4):
        assert g.get_ess(st) == orc.effective_sample_size()
        g.pf_resample(st, "multinomial_sorted", check=False); orc.resample("multinomial_sorted", check=False)
        assert np.array_equal(st.parents, orc.parents), t
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        assert _same(st, orc)
    st.close()


@pytest.mark.gpu
def test_full_size_properties(g, o):
    """BASELINE configs[1] size: N = 10^6, a resample every step; size-independent properties (test/resample.jl:11-12) + the oracle"""
    N = 1_000_000
    model, ys, st, orc = _pair(g, o, N, 2, T=4)
    exact = g.models.kalman_loglik(model, ys)
    for t in range(1, 4):
        rows0 = st.traces.copy(); lml0 = g.get_lml_est(st)
        g.pf_resample(st, "multinomial_sorted", check=False)
        par = st.parents
        assert (np.diff(par) >= 0).all() and par[0] >= 1 and par[-1] <= N
        assert np.array_equal(st.traces, rows0[par - 1])
        assert abs(g.get_lml_est(st) - lml0) <= 1e-9 * abs(lml0)
        orc.resample("multinomial_sorted", check=False)
        assert np.array_equal(par, orc.parents)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        assert np.array_equal(st.log_weights, orc.lw)
    assert abs(g.get_lml_est(st) - exact) < 0.05
    st.close()
