"""The "lazy search" of the multinomial resampler (gpf_k_fused.hpp; an option, off by default: gpf_set_lazy_search):
pf_resample!(state, :multinomial) enqueues the weight
scan only; the plain pf_update! that follows draws the targets, finds the ancestors, gathers, propagates and writes state.parents in ONE
kernel (k_step_search); every other consumer of the resampled population runs the stand-alone search first.  Whatever the path, the
result is the oracle's, bit for bit (src/resample.jl:48-65 + src/update.jl:12-25)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(g, o, N, seed, name="lgssm2", keep_prev=False, T=8):
    model = g.models.by_name(name)
    ys = g.models.simulate(model, T)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=keep_prev).set_lazy_search(True)
    orc = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=keep_prev).initialize(ys[0])
    return model, ys, st, orc


def _same(st, orc):
    return np.array_equal(st.parents, orc.parents) and np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)


@pytest.mark.parametrize("name", ["lgssm2", "bearings4", "sv1", "object_motion", "line_model"])
@pytest.mark.parametrize("keep_prev", [False, True])
def test_fused_search_and_propagate_every_model(g, o, name, keep_prev):
    """no getter between pf_resample! and pf_update!: the fused kernel runs; state.parents is read AFTER the update"""
    model, ys, st, orc = _pair(g, o, 30_000, 4, name=name, keep_prev=keep_prev)
    for t in range(1, 5):
        g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        assert _same(st, orc), (name, keep_prev, t)
        assert g.get_lml_est(st) == orc.log_ml_estimate()
    st.close()


@pytest.mark.parametrize("N", [1, 2, 3, 100, 2047, 2048, 2049, 65_537, 1_000_000, 1_300_000, 2_400_000])
def test_fused_sizes(g, o, N):
    """odd sizes (ragged last slot pair), both key-group widths (32 cells up to 1.25 M particles, 64 up to 2.5 M)"""
    model, ys, st, orc = _pair(g, o, N, 9, T=4)
    for t in range(1, 3):
        g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
    assert _same(st, orc)
    assert g.get_lml_est(st) == orc.log_ml_estimate()
    st.close()


def test_every_other_consumer_runs_the_search_first(g, o):
    N = 50_000
    model, ys, st, orc = _pair(g, o, N, 12, keep_prev=True, T=20)
    t = 1

    def both(fg, fo):
        fg(); fo()

    def upd():
        nonlocal t
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t]); t += 1
    res = lambda: both(lambda: g.pf_resample(st, "multinomial", check=False), lambda: orc.resample("multinomial", check=False))
    # parents read between resample and update
    res(); assert np.array_equal(st.parents, orc.parents); upd(); assert _same(st, orc)
    # log-weights / rows read between
    res(); assert (st.log_weights == 0).all() and np.array_equal(st.traces, orc.rows); upd(); assert _same(st, orc)
    # ESS / log-ML between (they need the gathered population's weights: all 0)
    res(); assert g.get_ess(st) == orc.effective_sample_size() and g.get_lml_est(st) == orc.log_ml_estimate(); upd(); assert _same(st, orc)
    # rejuvenation right behind the resample (README.md:69-71)
    res(); g.pf_rejuvenate(st, None, (), 1, method="move"); orc.rejuvenate("move", 1); assert _same(st, orc); upd(); assert _same(st, orc)
    res(); g.pf_rejuvenate(st, None, (), 1, method="reweight"); orc.rejuvenate("reweight", 1); upd(); assert _same(st, orc)
    # two resamples in a row, of every kind
    for second in ("multinomial", "residual", "stratified", "multinomial_sorted"):
        res(); g.pf_resample(st, second, check=False); orc.resample(second, check=False); upd(); assert _same(st, orc), second
    # a view operation between (the view materialises its parent)
    res()
    v, ov = st[1000:3000], orc[1000:3000]
    g.pf_resample(v, "residual", check=False); ov.resample("residual", check=False)
    assert np.array_equal(st.traces, orc.rows)
    upd(); assert _same(st, orc)
    # a resize between
    res(); g.pf_multinomial_resize(st, 40_000, check=False); orc.resize(40_000, "multinomial", check=False); upd(); assert _same(st, orc)
    # a custom-proposal update is not the plain propagate
    res(); g.pf_update(st, (t + 1,), (None,), ys[t], g.locally_optimal, ()); orc.update(ys[t], proposal=True); t += 1
    assert _same(st, orc)
    # a checked resample (check = true / :warn read the flags while the scan runs) stays lazy
    g.pf_resample(st, "multinomial", check=True); orc.resample("multinomial", check=True); upd(); assert _same(st, orc)
    # priorities are not lazy (their update_weights! needs the ancestors at once)
    g.pf_resample(st, "multinomial", priority_fn=g.Tempering(0.5), check=False); orc.resample("multinomial", priority_alpha=0.5, check=False)
    upd(); assert np.array_equal(st.parents, orc.parents) and np.array_equal(st.traces, orc.rows)
    st.close()


def test_invalid_weights_through_the_lazy_path(g, o):
    """test/resample.jl:26-31: all -Inf weights, check = false: uniform fallback, log-weights 0 afterwards; the estimate becomes -Inf"""
    N = 10_000
    model, ys, st, orc = _pair(g, o, N, 3)
    lw = np.full(N, -np.inf)
    st.log_weights = lw; orc.lw = lw.copy()
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    assert _same(st, orc)
    assert g.get_lml_est(st) == orc.log_ml_estimate() == -np.inf
    st.close()


def test_trajectory_store_and_views_stay_eager(g, o):
    model = g.models.object_motion(); ys = g.models.simulate(model, 6)
    st = g.pf_initialize(model, (1,), ys[0], 5000, seed=6, keep_prev=True, history=16).set_lazy_search(True)
    orc = o.OracleFilter(model.model_id, model.params, 5000, 6, keep_prev=True, history=True).initialize(ys[0])
    for t in range(1, 5):
        g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
    assert _same(st, orc)
    assert g.mean(st, (2, 0)) == orc.history_mean(2, 0)
    st.close()


def test_eager_and_lazy_paths_agree_in_a_fresh_process(g, o):
    """GPF_LAZY_SEARCH=1 in the environment turns the lazy search on for every new handle; unset / 0 keeps the chain k_search_multi +
    k_step<GATHER>: same bits"""
    code = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import gpf_amd as g
m = g.models.lgssm2(); ys = g.models.simulate(m, 6)
st = g.pf_initialize(m, (1,), ys[0], 200_000, seed=5)
for t in range(1, 5):
    g.pf_resample(st, "multinomial", check=False)
    g.pf_update(st, (t + 1,), (None,), ys[t])
h = hashlib.sha256(); h.update(st.parents.tobytes()); h.update(st.traces.tobytes()); h.update(st.log_weights.tobytes())
print(h.hexdigest(), repr(g.get_lml_est(st)))
""" % ROOT
    outs = []
    for v in ("1", "0"):
        env = dict(os.environ, GPF_LAZY_SEARCH=v)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) > 64
