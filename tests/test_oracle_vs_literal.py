"""The exact-integer spec (gpf_oracle.c, what the GPU matches bit-for-bit) against the literal Float64
restatement of src/resample.jl + src/utils.jl (ref_literal.c), driven by the SAME indexed uniforms."""
import numpy as np
import pytest


def uniforms(o, seed, epoch, n, j0=0):
    """U52 of the resample stream (tag 3) for slots j0..j0+n-1: top 52 bits of the 64-bit U the spec uses."""
    return np.array([o.lib().o_resample_u52_d(seed, j0 + j, epoch) for j in range(n)])


@pytest.mark.parametrize("N", [10, 100, 1000])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ancestors_match_literal(g, o, N, seed):
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, 2.0, N)
    m = g.models.lgssm2()
    L = o.lib()
    w = np.empty(N)
    assert L.lit_safe_softmax(lw, N, w) == 0
    u = uniforms(o, seed, 0, N)
    for method in ["multinomial", "stratified", "stratified_sorted", "residual"]:
        f = o.OracleFilter(m.model_id, m.params, N, seed)
        f.lw = lw.copy()
        par = np.empty(N, np.int64)
        if method == "multinomial":
            f.resample("multinomial"); L.lit_multinomial(w, N, u, par)
        elif method == "stratified":
            f.resample("stratified", sort_particles=False)
            L.lit_stratified(w, np.arange(N, dtype=np.int64), N, u, par)
        elif method == "stratified_sorted":
            f.resample("stratified", sort_particles=True)
            order = np.argsort(-lw, kind="stable").astype(np.int64)     # sortperm(lw, rev=true), stable
            L.lit_stratified(w, order, N, u, par)
        else:
            f.resample("residual")
            nres = L.lit_residual(w, N, u, par)
            assert np.array_equal(np.sort(par[:nres]), par[:nres])      # deterministic head is sorted (resample.jl:98-106)
        assert np.array_equal(f.parents - 1, par), method


def test_update_weights_with_priorities_matches_literal(g, o):
    N = 500
    rng = np.random.default_rng(5)
    lw = rng.normal(0, 1.5, N)
    m = g.models.lgssm2()
    f = o.OracleFilter(m.model_id, m.params, N, 5); f.lw = lw.copy()
    f.resample("multinomial", priority_alpha=0.5)
    want = np.empty(N)
    o.lib().lit_update_weights(lw, 0.5 * lw, f.parents - 1, N, want)
    np.testing.assert_allclose(f.lw, want, rtol=1e-9, atol=1e-9)


def test_documented_deviation_exact_floor(g, o):
    """floor(N * w_i) is evaluated on exact rationals, so uniform weights are the identity for
    EVERY N; the Float64 expression of resample.jl:99 gives floor(49 * (1/49)) = 0 and resamples randomly."""
    N = 49
    w = np.full(N, 1.0 / N)
    par = np.zeros(N, np.int64)
    nres = o.lib().lit_residual(w, N, np.full(N, 0.5), par)
    assert nres == 0                                                     # literal Float64: no deterministic copies
    m = g.models.lgssm2()
    f = o.OracleFilter(m.model_id, m.params, N, 1); f.lw[:] = 0.0
    f.resample("residual")
    assert np.array_equal(f.parents, np.arange(1, N + 1))                # spec: identity
