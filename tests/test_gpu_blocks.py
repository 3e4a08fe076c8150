"""Many small filters in one state: gpf_resample_blocks / gpf_block_stats (gpf_k_block.hpp K11) -- the batched form of the
reference's loop over sub-states `for b in blocks; pf_resample!(state[b], method); end` (src/view.jl:16-48,
src/resample.jl:185-187,205-218; block-wise resampling test/resample.jl:130-162; the README loop README.md:60-79 per block).
Every block must equal the oracle's sub-state resample of that block bit for bit, all blocks under one epoch."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
METHODS = ["multinomial", "residual", "stratified"]


def make(g, o, model_name, N, seed=11, keep_prev=False, T=6):
    m = g.models.by_name(model_name); ys = g.models.simulate(m, T)
    st = g.pf_initialize(m, (1,), ys[0], N, seed=seed, keep_prev=keep_prev)
    f = o.OracleFilter(m.model_id, m.params, N, seed, keep_prev=keep_prev).initialize(ys[0])
    return m, ys, st, f


def oracle_blocks(f, nb, method, ess_frac=None, sort_particles=True, check=False, priority_alpha=None):
    """the loop over sub-states, every block under the call's one epoch (oracle/oracle.py resample_blocks); the mask of the blocks that resampled"""
    from oracle import oracle
    return oracle.resample_blocks(f, nb, method, ess_frac=ess_frac, sort_particles=sort_particles, check=check, priority_alpha=priority_alpha)


def same(st, f):
    return np.array_equal(st.traces, f.rows) and np.array_equal(st.log_weights, f.lw, equal_nan=True) and np.array_equal(st.parents, f.parents)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("sort_particles", [False, True])
@pytest.mark.parametrize("N,nb", [(100, 100), (1000, 100), (4096, 2048), (5000, 2048), (777, 64), (37, 1), (2500, 1000), (300, 7),
                                  (3000, 300), (2100, 512), (1000, 129), (640, 128), (1539, 513)])   # (<= 128: a wave, 2 per lane; <= 512: a wave, 8 per lane; else a workgroup)
def test_blocks_equal_the_loop_over_substates(g, o, method, sort_particles, N, nb):
    if sort_particles and method != "stratified":
        pytest.skip("sort_particles is a stratified option")
    m, ys, st, f = make(g, o, "lgssm2", N)
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t])
        n_res = g.pf_resample_blocks(st, nb, method, sort_particles=sort_particles, check=False)
        mask = oracle_blocks(f, nb, method, sort_particles=sort_particles)
        assert n_res == mask.sum() == (N + nb - 1) // nb
        assert same(st, f), (method, sort_particles, N, nb, t)
        ess, lml = g.block_stats(st, nb)
        for k, b0 in enumerate(range(0, N, nb)):
            v = f[b0:min(b0 + nb, N)]
            assert ess[k] == v.effective_sample_size() and lml[k] == v.log_ml_estimate()
    assert g.get_lml_est(st) == f.log_ml_estimate()                    # the whole filter's estimate: sub-states never touch log_ml_est
    st.close()


@pytest.mark.parametrize("model_name,keep_prev", [("bearings4", True), ("sv1", True), ("bearings4", False), ("object_motion", False)])
@pytest.mark.parametrize("method", METHODS)
def test_blocks_other_row_widths(g, o, model_name, keep_prev, method):
    m, ys, st, f = make(g, o, model_name, 1200, keep_prev=keep_prev)
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t])
        g.pf_resample_blocks(st, 100, method, check=False); oracle_blocks(f, 100, method)
        assert same(st, f), (model_name, method, t)
    st.close()


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("ess_frac", [0.5, 0.9, 0.05, 2.0])
def test_ess_triggered_per_block(g, o, method, ess_frac):
    """`if effective_sample_size(state[b]) < ess_frac * N; pf_resample!(state[b]); end` decided on the device, block by block"""
    N, nb = 3000, 100
    m, ys, st, f = make(g, o, "bearings4", N, keep_prev=True, T=12)
    seen = set()
    for t in range(1, 10):
        g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t])
        n_res = g.pf_resample_blocks(st, nb, method, ess_frac=ess_frac, check=False)
        mask = oracle_blocks(f, nb, method, ess_frac=ess_frac)
        assert n_res == mask.sum() and np.array_equal(g.block_resampled(st), mask)
        assert same(st, f), (method, ess_frac, t)
        seen.add(int(n_res))
    if ess_frac == 2.0:
        assert seen == {N // nb}                                         # ESS <= N: every block, every step
    if ess_frac == 0.05:
        assert len(seen) > 1                                             # some steps resample some of the blocks only
    st.close()


@pytest.mark.parametrize("method", METHODS)
def test_first_block_equals_a_device_view(g, o, method):
    """the same resample through the device's own view of block 0 (gpf_view_create + gpf_resample; its epoch is the call's epoch
    in both forms): identical particles, weights, parents"""
    N, nb = 600, 100
    m, ys, st, f = make(g, o, "lgssm2", N)
    st2 = g.pf_initialize(m, (1,), ys[0], N, seed=11)
    g.pf_update(st, (2,), (None,), ys[1]); g.pf_update(st2, (2,), (None,), ys[1])
    g.pf_resample_blocks(st, nb, method, check=False)
    g.pf_resample(st2[0:nb], method, check=False)
    assert np.array_equal(st.traces[:nb], st2.traces[:nb]) and np.array_equal(st.log_weights[:nb], st2.log_weights[:nb])
    assert np.array_equal(st.parents[:nb], st2.parents[:nb])
    st.close(); st2.close()


@pytest.mark.parametrize("method", METHODS)
def test_adversarial_block_weights(g, o, method):
    """blocks with equal weights, one dominant particle, all -Inf (uniform fallback, flagged invalid), wide ranges, -Inf-heavy"""
    N, nb = 1000, 100
    m, ys, st, f = make(g, o, "lgssm2", N)
    rng = np.random.default_rng(5)
    lw = np.concatenate([np.zeros(nb), np.where(np.arange(nb) == 17, 0.0, -800.0), np.full(nb, -np.inf), -700.0 * rng.random(nb),
                         np.where(rng.random(nb) < 0.8, -np.inf, -rng.random(nb)), -1e-9 * rng.random(nb), np.full(nb, -3.25),
                         -np.arange(nb, dtype=np.float64), np.where(np.arange(nb) % 2 == 0, -0.0, 0.0), -50.0 * rng.random(nb) ** 4])
    st.log_weights = lw; f.lw = lw.copy()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        g.pf_resample_blocks(st, nb, method, check="warn")
        assert any("Invalid weights" in str(x.message) for x in w)       # the all -Inf block
    oracle_blocks(f, nb, method, check="warn")
    assert same(st, f)
    g.pf_update(st, (2,), (None,), ys[1]); f.update(ys[1])
    assert same(st, f)
    st.close()


def test_invalid_blocks_are_reported_and_left_alone(g, o):
    N, nb = 400, 100
    m, ys, st, f = make(g, o, "lgssm2", N)
    g.pf_update(st, (2,), (None,), ys[1]); f.update(ys[1])
    lw = st.log_weights.copy(); lw[150] = np.nan; lw[300:400] = -np.inf
    st.log_weights = lw; f.lw = lw.copy()
    rows0 = st.traces.copy()
    with pytest.raises(g.ErrorException, match="Invalid weights"):
        g.pf_resample_blocks(st, nb, "multinomial", check=False)          # NaN always raises (the reference's Categorical rejects it)
    # blocks 0, 2, 3 resampled (3: uniform fallback), block 1 untouched
    e = f.epoch
    for b in (0, 2, 3):
        f.epoch = e; f[b * nb:(b + 1) * nb].resample("multinomial", check=False)
    f.epoch = e + 1
    assert np.array_equal(st.traces, f.rows) and np.array_equal(st.traces[100:200], rows0[100:200])
    assert np.array_equal(st.log_weights, f.lw, equal_nan=True)
    assert np.array_equal(g.block_resampled(st), [True, False, True, True])
    # check = true: the all -Inf block is refused as well
    lw = st.log_weights.copy(); lw[150] = -1.0; lw[0:100] = -np.inf
    st.log_weights = lw; f.lw = lw.copy()
    with pytest.raises(g.ErrorException, match="Invalid weights"):
        g.pf_resample_blocks(st, nb, "residual", check=True)
    e = f.epoch
    for b in (1, 2, 3):
        f.epoch = e; f[b * nb:(b + 1) * nb].resample("residual", check=False)
    f.epoch = e + 1
    assert np.array_equal(st.traces, f.rows) and np.array_equal(st.log_weights, f.lw, equal_nan=True)
    st.close()


def test_readme_loop_per_block(g, o):
    """README.md:60-79 on 200 filters of 100 particles at once: update, ESS-triggered residual resample per block, MH move"""
    N, nb = 20_000, 100
    m, ys, st, f = make(g, o, "object_motion", N, keep_prev=True, T=10)
    for t in range(1, 9):
        g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t])
        g.pf_resample_blocks(st, nb, "residual", ess_frac=0.5, check=False); oracle_blocks(f, nb, "residual", ess_frac=0.5)
        g.pf_rejuvenate(st, None, (), 1, method="move"); f.rejuvenate("move", 1)
        assert same(st, f), t
    ess, lml = g.block_stats(st, nb)
    assert all(lml[k] == f[k * nb:(k + 1) * nb].log_ml_estimate() for k in range(0, N // nb, 17))
    assert np.isfinite(lml).all() and lml.std() > 0                     # 200 independent estimates of the same log p(y)
    st.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_sequences_with_block_resamples(g, o, seed):
    rng = np.random.default_rng(100 + seed)
    N = int(rng.choice([500, 1024, 3001])); nb = int(rng.choice([50, 128, 2048, 999, 300, 512]))
    m, ys, st, f = make(g, o, "bearings4", N, keep_prev=True, T=40)
    t = 1
    for _ in range(25):
        op = rng.choice(["update", "blocks", "blocks_ess", "whole", "move", "stats"], p=[0.3, 0.25, 0.15, 0.1, 0.1, 0.1])
        method = str(rng.choice(METHODS)); sp = bool(rng.integers(2))
        if op == "update":
            g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t]); t += 1
        elif op == "blocks":
            g.pf_resample_blocks(st, nb, method, sort_particles=sp, check=False); oracle_blocks(f, nb, method, sort_particles=sp)
        elif op == "blocks_ess":
            fr = float(rng.choice([0.3, 0.6, 0.95]))
            assert g.pf_resample_blocks(st, nb, method, ess_frac=fr, sort_particles=sp, check=False) == oracle_blocks(f, nb, method, ess_frac=fr, sort_particles=sp).sum()
        elif op == "whole":
            kw = {"sort_particles": sp} if method == "stratified" else {}
            g.pf_resample(st, method, check=False, **kw); f.resample(method, check=False, **kw)
        elif op == "move":
            g.pf_rejuvenate(st, None, (), 1, method="move"); f.rejuvenate("move", 1)
        else:
            ess, lml = g.block_stats(st, nb)
            k = int(rng.integers(len(ess))); v = f[k * nb:min((k + 1) * nb, N)]
            assert (ess[k] == v.effective_sample_size() or (np.isnan(ess[k]) and np.isnan(v.effective_sample_size()))) and lml[k] == v.log_ml_estimate()
        assert same(st, f), (seed, op, method)
    assert g.get_lml_est(st) == f.log_ml_estimate()
    st.close()


def test_argument_errors(g, o):
    m, ys, st, f = make(g, o, "lgssm2", 300)
    with pytest.raises(g.ErrorException):
        g.pf_resample_blocks(st, 0, "multinomial")
    with pytest.raises(g.ErrorException, match="not recognized"):
        g.pf_resample_blocks(st, 100, "systematic")
    with pytest.raises(g.ErrorException):
        g.pf_resample_blocks(st[0:100], 50, "multinomial")
    with pytest.raises(g.ErrorException):
        g.block_resampled(g.pf_initialize(m, (1,), ys[0], 10, seed=1))
    st.close()


# ----------------------------------------------------------------------------- per-block observations: many DATASETS in one state
def oracle_init_blocks(o, f, nb, obs_rows):
    return o.initialize_blocks(f, nb, obs_rows)


def oracle_update_blocks(f, nb, obs_rows):
    from oracle import oracle
    oracle.update_blocks(f, nb, obs_rows)


def oracle_rejuvenate_blocks(f, nb, obs_rows, method, mask=None, n_iters=1):
    from oracle import oracle
    return oracle.rejuvenate_blocks(f, nb, obs_rows, method, mask=mask, n_iters=n_iters)


@pytest.mark.parametrize("model_name", ["lgssm2", "bearings4", "object_motion", "sv1"])
@pytest.mark.parametrize("N,nb", [(1000, 100), (1030, 100), (4096, 2048), (600, 7)])
def test_per_block_observations(g, o, model_name, N, nb):
    """every block is a filter on ITS OWN data: initialise / update / rejuvenate block by block in one launch each ==
    the loop over sub-states with per-view observations (test/update.jl:179-189, test/rejuvenate.jl:73-103)"""
    m = g.models.by_name(model_name)
    B, T = (N + nb - 1) // nb, 6
    rng = np.random.default_rng(7)
    base = np.asarray(g.models.simulate(m, T))
    ys = base[None, :, :] + 0.3 * rng.standard_normal((B,) + base.shape)           # ys[b][t]: a perturbed copy of one sequence per block
    st = g.pf_initialize_blocks(m, (1,), ys[:, 0], N, nb, seed=13, keep_prev=True)
    f = o.OracleFilter(m.model_id, m.params, N, 13, keep_prev=True)
    oracle_init_blocks(o, f, nb, ys[:, 0])
    assert same(st, f)
    for t in range(1, T):
        g.pf_update_blocks(st, (t + 1,), (None,), ys[:, t], nb); oracle_update_blocks(f, nb, ys[:, t])
        assert same(st, f), (model_name, "update", t)
        n_res = g.pf_resample_blocks(st, nb, "residual", ess_frac=0.5, check=False)
        mask = oracle_blocks(f, nb, "residual", ess_frac=0.5)
        assert n_res == mask.sum() and same(st, f), (model_name, "resample", t)
        if t % 2:
            acc = g.pf_rejuvenate_blocks(st, None, (), 1, method="move", only_resampled=True, count=True)
            assert acc == oracle_rejuvenate_blocks(f, nb, ys[:, t], "move", mask)
        else:
            acc = g.pf_rejuvenate_blocks(st, None, (), 1, method="reweight", only_resampled=bool(t % 4), count=True)
            assert acc == oracle_rejuvenate_blocks(f, nb, ys[:, t], "reweight", mask if t % 4 else None)
        assert same(st, f), (model_name, "rejuvenate", t)
    ess, lml = g.block_stats(st, nb)
    for k in range(0, B, max(1, B // 7)):
        v = f[k * nb:min((k + 1) * nb, N)]
        assert lml[k] == v.log_ml_estimate()
    # the whole-filter pf_rejuvenate after a block-wise update uses the per-block observations; pf_update goes back to one observation
    g.pf_rejuvenate(st, None, (), 1, method="move"); oracle_rejuvenate_blocks(f, nb, ys[:, T - 1], "move")
    assert same(st, f)
    g.pf_update(st, (T + 1,), (None,), base[0]); f.update(base[0])
    g.pf_rejuvenate(st, None, (), 1, method="move"); f.rejuvenate("move", 1)
    assert same(st, f)
    st.close()


def test_block_step_argument_errors(g, o):
    m = g.models.lgssm2(); ys = np.asarray(g.models.simulate(m, 3))
    st = g.pf_initialize(m, (1,), ys[0], 300, seed=1, keep_prev=True)
    with pytest.raises(g.ErrorException):
        g.pf_update_blocks(st, (2,), (None,), np.zeros((2, 2)), 100)             # 3 blocks, 2 rows
    with pytest.raises(g.ErrorException):
        g.pf_update_blocks(st, (2,), (None,), np.zeros((3, 5)), 100)             # wrong observation width
    with pytest.raises(g.ErrorException):
        g.pf_rejuvenate_blocks(st, None, (), 1)                                  # no per-block observations yet
    g.pf_update_blocks(st, (2,), (None,), np.tile(ys[1], (3, 1)), 100)
    with pytest.raises(g.ErrorException):
        g.pf_rejuvenate_blocks(st, None, (), 1, only_resampled=True)             # no block resample yet
    with pytest.raises(g.ErrorException):
        g.pf_rejuvenate(st[0:100], None, (), 1)                                  # a view after a block-wise update
    st.close()


@pytest.mark.parametrize("nb", [100, 512, 2048])
def test_many_blocks_at_once(g, o, nb):
    """a grid of thousands of teams (more than one workgroup round per CU): 200 000 particles in blocks of 100 / 512 / 2048, per-block data,
    ESS-gated residual resample, masked MH move -- against the loop over sub-states"""
    N = 200_000
    m = g.models.bearings4(); base = np.asarray(g.models.simulate(m, 4))
    B = (N + nb - 1) // nb
    rng = np.random.default_rng(nb)
    ys = base[None, :, :] + 0.3 * rng.standard_normal((B,) + base.shape)
    st = g.pf_initialize_blocks(m, (1,), ys[:, 0], N, nb, seed=21, keep_prev=True)
    f = oracle_init_blocks(o, o.OracleFilter(m.model_id, m.params, N, 21, keep_prev=True), nb, ys[:, 0])
    for t in range(1, 3):
        g.pf_update_blocks(st, (t + 1,), (None,), ys[:, t], nb); oracle_update_blocks(f, nb, ys[:, t])
        n_res = g.pf_resample_blocks(st, nb, "residual", ess_frac=0.6, check=False)
        mask = oracle_blocks(f, nb, "residual", ess_frac=0.6)
        assert n_res == mask.sum() and np.array_equal(g.block_resampled(st), mask)
        g.pf_rejuvenate_blocks(st, None, (), 1, method="move", only_resampled=True); oracle_rejuvenate_blocks(f, nb, ys[:, t], "move", mask)
        assert same(st, f), (nb, t)
    ess, lml = g.block_stats(st, nb)
    ks = rng.choice(B, size=min(B, 25), replace=False)
    assert all(lml[k] == f[int(k) * nb:min((int(k) + 1) * nb, N)].log_ml_estimate() for k in ks)
    st.close()


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("alpha", [0.5, 2.0])
@pytest.mark.parametrize("N,nb", [(1000, 100), (4100, 2048), (900, 300), (260, 64)])
def test_blocks_with_tempering(g, o, method, alpha, N, nb):
    """priority_fn = w -> alpha w per block (src/resample.jl:51-52, the tempering family of test/resample.jl:15 on sub-states,
    test/resample.jl:130-162): ancestors from the priorities, weights log_ws + (logsumexp(block) - logsumexp(log_ws)) -- with and
    without sort_particles and the ESS gate (which tests the RAW weights), against the loop over sub-states"""
    m, ys, st, f = make(g, o, "bearings4", N, keep_prev=True, T=8)
    for t in range(1, 6):
        g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t])
        fr = None if t % 2 else 0.7
        sp = bool(t % 3 == 0)
        n_res = g.pf_resample_blocks(st, nb, method, priority_fn=g.Tempering(alpha), ess_frac=fr, sort_particles=sp, check=False)
        mask = oracle_blocks(f, nb, method, ess_frac=fr, sort_particles=sp, priority_alpha=alpha)
        assert n_res == mask.sum() and same(st, f), (method, alpha, N, nb, t)
        ess, lml = g.block_stats(st, nb)
        k = t % len(ess); v = f[k * nb:min((k + 1) * nb, N)]
        assert lml[k] == v.log_ml_estimate()                            # the block's estimate is kept by the weight update (:215-216)
    st.close()


# ----------------------------------------------------------------------------------------------- blocks of more than 2048 particles
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("N,nb", [(10_000, 4096), (8192, 4096), (120_000, 50_000), (5000, 2049), (300, 4096)])
def test_big_blocks_equal_the_loop_over_substates(g, o, method, N, nb):
    """test/resample.jl:130-162 puts no limit on the size of a sub-state: blocks beyond the one-workgroup kernels' 2048 particles go through
    the full-size kernels block by block -- same results as the loop over views, one epoch for all blocks, ESS gate, priorities, statistics"""
    m, ys, st, f = make(g, o, "lgssm2", N, keep_prev=True)
    B = (N + nb - 1) // nb
    for t in range(1, 4):
        g.pf_update(st, (t + 1,), (None,), ys[t]); f.update(ys[t])
        if t == 1:
            n_res = g.pf_resample_blocks(st, nb, method, sort_particles=(method == "stratified"), check=False)
            mask = oracle_blocks(f, nb, method, sort_particles=(method == "stratified"))
        elif t == 2:
            n_res = g.pf_resample_blocks(st, nb, method, ess_frac=0.7, sort_particles=False, check=False)
            mask = oracle_blocks(f, nb, method, ess_frac=0.7, sort_particles=False)
        else:
            n_res = g.pf_resample_blocks(st, nb, method, priority_fn=g.Tempering(0.5), sort_particles=False, check=False)
            mask = oracle_blocks(f, nb, method, priority_alpha=0.5, sort_particles=False)
        assert n_res == mask.sum() and np.array_equal(g.block_resampled(st), mask), (method, N, nb, t)
        assert np.array_equal(st.traces, f.rows) and np.array_equal(st.parents, f.parents), (method, N, nb, t)
        np.testing.assert_allclose(st.log_weights, f.lw, rtol=1e-12, atol=1e-12)
        f.lw = st.log_weights.copy()                                   # (tempered weights: keep the two in lockstep)
        ess, lml = g.block_stats(st, nb)
        for k in range(B):
            v = f[k * nb:min((k + 1) * nb, N)]
            assert ess[k] == v.effective_sample_size() and lml[k] == v.log_ml_estimate()
    assert g.get_lml_est(st) == f.log_ml_estimate()
    st.close()


def test_big_blocks_steps_and_invalid_blocks(g, o):
    """per-block observations and masked rejuvenation at block sizes beyond 2048; a NaN block is left as it stands and reported"""
    N, nb = 9000, 4000
    m = g.models.bearings4(); base = np.asarray(g.models.simulate(m, 4))
    B = (N + nb - 1) // nb
    rng = np.random.default_rng(3)
    ys = base[None, :, :] + 0.3 * rng.standard_normal((B,) + base.shape)
    st = g.pf_initialize_blocks(m, (1,), ys[:, 0], N, nb, seed=21, keep_prev=True)
    f = oracle_init_blocks(o, o.OracleFilter(m.model_id, m.params, N, 21, keep_prev=True), nb, ys[:, 0])
    for t in range(1, 3):
        g.pf_update_blocks(st, (t + 1,), (None,), ys[:, t], nb); oracle_update_blocks(f, nb, ys[:, t])
        n_res = g.pf_resample_blocks(st, nb, "residual", ess_frac=0.6, check=False)
        mask = oracle_blocks(f, nb, "residual", ess_frac=0.6)
        assert n_res == mask.sum() and np.array_equal(g.block_resampled(st), mask)
        g.pf_rejuvenate_blocks(st, None, (), 1, method="move", only_resampled=True); oracle_rejuvenate_blocks(f, nb, ys[:, t], "move", mask)
        assert same(st, f), t
    lw = st.log_weights.copy(); lw[nb + 5] = np.nan                    # block 1 is invalid
    st.log_weights = lw; f.lw = lw.copy()
    rows0 = st.traces.copy()
    with pytest.raises(g.ErrorException, match="NaN"):
        g.pf_resample_blocks(st, nb, "multinomial", check="warn")
    assert np.array_equal(st.traces[nb:2 * nb], rows0[nb:2 * nb])       # left as it stands
    assert not np.array_equal(st.traces[:nb], rows0[:nb])               # the others resampled
    st.close()


@pytest.mark.parametrize("N,nb", [(1000, 100), (1030, 100), (6000, 2500), (600, 7)])
def test_per_block_proposals(g, o, N, nb):
    """"Update with different proposals per view" (test/update.jl:179-189) in ONE launch: every block with its own observation AND its own
    choice of proposal -- the model's native one (update.jl:79-96) or the default (update.jl:12-25) -- equals the loop over sub-states bit for bit"""
    m = g.models.lgssm2()
    B, T = (N + nb - 1) // nb, 5
    rng = np.random.default_rng(11)
    base = np.asarray(g.models.simulate(m, T))
    ys = base[None, :, :] + 0.3 * rng.standard_normal((B,) + base.shape)
    st = g.pf_initialize_blocks(m, (1,), ys[:, 0], N, nb, seed=13, keep_prev=True)
    f = o.OracleFilter(m.model_id, m.params, N, 13, keep_prev=True)
    oracle_init_blocks(o, f, nb, ys[:, 0])
    for t in range(1, T):
        flags = rng.random(B) < 0.5 if t < T - 1 else np.ones(B, bool)
        g.pf_update_blocks(st, (t + 1,), (None,), ys[:, t], nb, proposals=[g.locally_optimal if fl else None for fl in flags])
        from oracle import oracle
        oracle.update_blocks(f, nb, ys[:, t], proposals=flags)
        assert same(st, f), ("update with per-block proposals", t)
        n_res = g.pf_resample_blocks(st, nb, "stratified", ess_frac=0.7, check=False)
        assert n_res == oracle_blocks(f, nb, "stratified", ess_frac=0.7).sum() and same(st, f)
    with pytest.raises(g.ErrorException):                                     # one entry per block
        g.pf_update_blocks(st, (T + 1,), (None,), ys[:, 0], nb, proposals=[None])
    sv = g.pf_initialize_blocks(g.models.sv1(), (1,), np.zeros((2, 1)), 100, 50)
    with pytest.raises(g.ErrorException):                                     # a model without a native proposal
        g.pf_update_blocks(sv, (2,), (None,), np.zeros((2, 1)), 50, proposals=[g.locally_optimal, None])
    st.close(); sv.close()


def test_reference_update_with_different_proposals_per_view_in_one_launch(g, o):
    """test/update.jl:179-189 restated on the batched call: line_model, 100 particles, state[1:50] extended to step 10 with the default proposal,
    state[51:end] with outlier_propose (outlier ~ bernoulli(0.0)): y_10 == 0 everywhere (the observation), outlier == false in the second half,
    no weight is 0"""
    m = g.models.line_model()
    st = g.pf_initialize(m, (0,), g.models.line_obs(0), 100, seed=5)
    obs = np.stack([g.models.line_obs(10), g.models.line_obs(10)])
    g.pf_update_blocks(st, (10,), (None,), obs, 50, proposals=[None, g.line_fixed])
    rows, lw = st.traces, st.log_weights
    assert np.all(rows[50:, 1] == 0.0)                                       # :line => 10 => :outlier == false under outlier_propose
    assert np.all(lw != 0.0) and np.all(np.isfinite(lw))
    f = o.OracleFilter(m.model_id, m.params, 100, 5).initialize(g.models.line_obs(0))
    from oracle import oracle
    oracle.update_blocks(f, 50, obs, proposals=[False, True])
    assert np.array_equal(rows, f.rows) and np.array_equal(lw, f.lw)
    st.close()


@pytest.mark.parametrize("model_name,init_strata,upd_strata", [("object_motion", [0.0, 1.0], [0.0, 1.0]), ("line_model", [-2.0, -1.0, 0.0, 1.0, 2.0], [0.0, 1.0])])
@pytest.mark.parametrize("N,nb", [(1000, 100), (1030, 100), (5000, 2300), (600, 7)])
@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_per_block_strata(g, o, model_name, init_strata, upd_strata, N, nb, layout):
    """stratified initialisation / update of every block by itself (src/initialize.jl:92-109, src/update.jl:193-210 on each sub-state, with
    stratified_map! of src/utils.jl:29-55 over the block's own particles -- ragged last block, more strata than particles, both layouts) in one launch
    each == the loop over sub-states, bit for bit"""
    m = g.models.by_name(model_name)
    B, T = (N + nb - 1) // nb, 4
    rng = np.random.default_rng(3)
    if model_name == "line_model":
        ys = np.stack([[g.models.line_obs(t + 1, float(rng.integers(-2, 3))) for t in range(T)] for _ in range(B)])
    else:
        base = np.asarray(g.models.simulate(m, T))
        ys = base[None, :, :] + 0.2 * rng.standard_normal((B,) + base.shape)
    st = g.pf_initialize_blocks(m, (1,), ys[:, 0], N, nb, seed=13, keep_prev=True, strata=init_strata, layout=layout)
    f = o.OracleFilter(m.model_id, m.params, N, 13, keep_prev=True)
    o.initialize_blocks(f, nb, ys[:, 0], strata=init_strata, layout=layout)
    assert same(st, f), "stratified initialisation per block"
    for t in range(1, T):
        g.pf_update_blocks(st, (t + 1,), (None,), ys[:, t], nb, strata=upd_strata, layout=layout)
        o.update_blocks(f, nb, ys[:, t], strata=upd_strata, layout=layout)
        assert same(st, f), ("stratified update per block", t)
        n_res = g.pf_resample_blocks(st, nb, "residual", ess_frac=0.9, check=False)
        assert n_res == oracle_blocks(f, nb, "residual", ess_frac=0.9).sum() and same(st, f)
    with pytest.raises(g.ErrorException):
        g.pf_update_blocks(st, (T + 1,), (None,), ys[:, 0], nb, strata=upd_strata, proposals=[None] * B)
    st.close()
    sv = g.pf_initialize_blocks(g.models.sv1(), (1,), np.zeros((2, 1)), 100, 50)
    with pytest.raises(g.ErrorException):                                     # a model without a discrete latent
        sv._check(sv._L.gpf_update_blocks_strata(sv._h, np.zeros(2).ctypes.data_as(g._lib.C.POINTER(g._lib.C.c_double)), 1, 50,
                                                 np.zeros(2).ctypes.data_as(g._lib.C.POINTER(g._lib.C.c_double)), 2, 1))
    sv.close()
