"""Many small filters in one state, oracle side (CPU): the loops over sub-states that gpf_resample_blocks / gpf_update_blocks /
gpf_rejuvenate_blocks batch (oracle/oracle.py resample_blocks & co.) keep the invariants of the reference's block-wise tests
(test/resample.jl:130-162, test/update.jl:179-189, test/rejuvenate.jl:73-103) for any block size."""
import numpy as np
import pytest

METHODS = ["multinomial", "residual", "stratified"]


def make(g, o, N, model_name="lgssm2", T=5):
    m = g.models.by_name(model_name); ys = np.asarray(g.models.simulate(m, T))
    return m, ys, o.OracleFilter(m.model_id, m.params, N, 9, keep_prev=True).initialize(ys[0])


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("N,nb", [(100, 50), (100, 100), (257, 64), (300, 7), (40, 1)])
def test_blockwise_resampling_invariants(g, o, method, N, nb):
    """test/resample.jl:130-162 for every block: new == old[parents] inside the block, the block's log-ML estimate unchanged, the
    filter's estimate unchanged, log_ml_est untouched; the epoch advances once per call"""
    m, ys, f = make(g, o, N)
    f.update(ys[1])
    old, lml_full, e0 = f.rows.copy(), f.log_ml_estimate(), f.epoch
    lml_b = [f[a:b].log_ml_estimate() for a, b in o.blocks_of(f, nb)]
    mask = o.resample_blocks(f, nb, method)
    assert mask.all() and f.epoch == e0 + 1 and f.lml_est == 0.0
    for k, (a, b) in enumerate(o.blocks_of(f, nb)):
        v = f[a:b]
        assert np.array_equal(v.rows, old[a:b][v.parents - 1])                     # :151
        assert abs(v.log_ml_estimate() - lml_b[k]) < 1e-9                          # :152
        assert np.all(v.lw == v.lw[0])                                             # the block's average weight everywhere (resample.jl:210)
    assert abs(f.log_ml_estimate() - lml_full) < 1e-9                              # :160


@pytest.mark.parametrize("method", METHODS)
def test_ess_gate_is_per_block(g, o, method):
    """`if effective_sample_size(state[b]) < ess_frac * n; pf_resample!(state[b]); end`: blocks above the threshold stay untouched
    (rows, weights, parents); an all -Inf block (ESS NaN) is never resampled under a gate"""
    m, ys, f = make(g, o, 400, "bearings4", T=8)
    for t in range(1, 4):
        f.update(ys[t])
    f.lw[300:400] = -np.inf
    rows0, lw0, par0 = f.rows.copy(), f.lw.copy(), f.parents.copy()
    ess = np.array([f[a:b].effective_sample_size() for a, b in o.blocks_of(f, 100)])
    thr = float(np.nanmedian(ess)) / 100 + 1e-9
    mask = o.resample_blocks(f, 100, method, ess_frac=thr)
    assert np.isnan(ess[3]) and not mask[3]
    for k, (a, b) in enumerate(o.blocks_of(f, 100)):
        assert mask[k] == (ess[k] < thr * 100)
        if not mask[k]:
            assert np.array_equal(f.rows[a:b], rows0[a:b]) and np.array_equal(f.lw[a:b], lw0[a:b], equal_nan=True) and np.array_equal(f.parents[a:b], par0[a:b])
    assert 0 < mask.sum() < 4


def test_per_block_update_and_rejuvenation(g, o):
    """test/update.jl:179-189 / test/rejuvenate.jl:73-103 with one observation vector per block: every block's weights move by ITS
    data's log-likelihood; a masked rejuvenation touches the masked blocks only; move-accept leaves weights alone"""
    m, ys, f = make(g, o, 300)
    rng = np.random.default_rng(1)
    obs = ys[1][None, :] + rng.standard_normal((3, 2))
    lw0 = f.lw.copy()
    o.update_blocks(f, 100, obs)
    sr = m.info["sr"]
    for k, (a, b) in enumerate(o.blocks_of(f, 100)):
        want = sum(-0.5 * ((obs[k][c] - f.rows[a:b, c]) / sr) ** 2 - np.log(sr) - 0.5 * np.log(2 * np.pi) for c in range(2))
        np.testing.assert_allclose(f.lw[a:b] - lw0[a:b], want, rtol=1e-10, atol=1e-10)
    rows1, lw1 = f.rows.copy(), f.lw.copy()
    acc = o.rejuvenate_blocks(f, 100, obs, "move", mask=np.array([True, False, True]))
    assert np.array_equal(f.lw, lw1) and 0 < acc <= 200
    assert np.array_equal(f.rows[100:200], rows1[100:200]) and not np.array_equal(f.rows[0:100], rows1[0:100])
    o.rejuvenate_blocks(f, 100, obs, "reweight", mask=np.array([False, True, False]))
    assert np.array_equal(f.lw[:100], lw1[:100]) and np.all(f.lw[100:200] != lw1[100:200]) and np.array_equal(f.lw[200:], lw1[200:])


def test_block_initialisation_uses_each_blocks_data(g, o):
    m, ys, _ = make(g, o, 10)
    f = o.OracleFilter(m.model_id, m.params, 300, 4)
    obs = np.array([[0.0, 0.0], [50.0, -50.0], [1.0, 2.0]])
    o.initialize_blocks(f, 100, obs)
    g1 = o.OracleFilter(m.model_id, m.params, 300, 4).initialize(obs[1])
    assert np.array_equal(f.rows[100:200], g1.rows[100:200]) and np.array_equal(f.lw[100:200], g1.lw[100:200])    # same RNG counters (global ids), block 1's data
    assert not np.array_equal(f.lw[0:100], g1.lw[0:100])
    assert f.epoch == 1 and np.array_equal(f.parents, np.arange(1, 301))


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [0.5, 2.0])
def test_blockwise_tempered_resampling_invariants(g, o, method, alpha):
    """test/resample.jl:130-162 with priority_fn = w -> alpha w on every block: new == old[parents] inside the block and the block's
    log-ML estimate is kept by the weight update log_ws + (logsumexp(block) - logsumexp(log_ws)) (resample.jl:213-216)"""
    m, ys, f = make(g, o, 300)
    f.update(ys[1])
    old, lml_full = f.rows.copy(), f.log_ml_estimate()
    lml_b = [f[a:b].log_ml_estimate() for a, b in o.blocks_of(f, 100)]
    o.resample_blocks(f, 100, method, priority_alpha=alpha)
    for k, (a, b) in enumerate(o.blocks_of(f, 100)):
        v = f[a:b]
        assert np.array_equal(v.rows, old[a:b][v.parents - 1])
        assert abs(v.log_ml_estimate() - lml_b[k]) < 1e-9
        assert len(np.unique(v.lw)) > 1                                          # weights over priorities: no longer all equal
    assert abs(f.log_ml_estimate() - lml_full) < 1e-9 and f.lml_est == 0.0
