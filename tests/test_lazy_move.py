"""The "lazy move" (gpf_k_step.hpp k_move_step): pf_rejuvenate!(state, kern, args, n_iters; method) enqueues nothing when its
acceptance count is not asked for; the plain pf_update! that follows runs gather (if a resample is pending) -> move -> propagate in ONE
kernel, every other consumer of the state runs the stand-alone move first.  Whatever the path, the state is the oracle's, bit for bit
(src/rejuvenate.jl:40-90,125-132 + src/update.jl:12-25; the README loop's order of calls, README.md:66-76)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = ["multinomial", "residual", "stratified", "multinomial_sorted"]


def _pair(g, o, N, seed, name, T=10):
    model = g.models.by_name(name)
    ys = g.models.simulate(model, T)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=True).initialize(ys[0])
    return model, ys, st, orc


def _same(st, orc):
    return np.array_equal(st.parents, orc.parents) and np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw)


@pytest.mark.parametrize("name", ["lgssm2", "bearings4", "sv1", "object_motion", "line_model"])
@pytest.mark.parametrize("method", ["move", "reweight"])
@pytest.mark.parametrize("N", [1, 63, 64, 65, 5000, 70_001])
def test_move_then_update_every_model(g, o, name, method, N):
    """resample -> rejuvenate -> update with no getter in between: the fused kernel with the pending gather; then rejuvenate -> update
    without a resample (no gather); the state is read only after the update"""
    model, ys, st, orc = _pair(g, o, N, 6, name)
    t = 1
    for rep, res in enumerate(RES):
        kw = dict(sort_particles=False) if res == "stratified" else {}
        g.pf_resample(st, res, check=False, **kw); orc.resample(res, check=False, **kw)
        it = 1 + rep % 3
        g.pf_rejuvenate(st, None, (), it, method=method); orc.rejuvenate(method, it)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t]); t += 1
        assert _same(st, orc), (name, method, N, res)
        g.pf_rejuvenate(st, None, (), 1, method=method); orc.rejuvenate(method, 1)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t]); t += 1
        assert _same(st, orc), (name, method, N, res, "no gather")
    assert g.get_lml_est(st) == orc.log_ml_estimate() and g.get_ess(st) == orc.effective_sample_size()
    st.close()


def test_every_other_consumer_runs_the_move_first(g, o):
    N = 20_000
    model, ys, st, orc = _pair(g, o, N, 9, "bearings4", T=24)
    t = 1

    def upd():
        nonlocal t
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t]); t += 1

    def mv(method="move", it=1):
        g.pf_rejuvenate(st, None, (), it, method=method); orc.rejuvenate(method, it)
    res = lambda m="residual": (g.pf_resample(st, m, check=False), orc.resample(m, check=False))
    upd()
    # a getter between the move and the update
    res(); mv(); assert _same(st, orc); upd(); assert _same(st, orc)
    res(); mv("reweight"); assert g.get_ess(st) == orc.effective_sample_size(); upd(); assert _same(st, orc)
    # two moves in a row, a move of zero iterations, a move whose count is asked for (eager)
    mv("reweight", 2); mv("move", 1); upd(); assert _same(st, orc)
    mv("move", 0); upd(); assert _same(st, orc)
    g.pf_move_accept(st, g.mh, (), 2, count=True); orc.rejuvenate("move", 2); assert st.n_accepted == orc.n_accepted
    upd(); assert _same(st, orc)
    # a resample, a resize, a view operation, a block-wise step, a custom-proposal update behind a pending move
    mv(); res("multinomial"); upd(); assert _same(st, orc)
    mv("reweight"); g.pf_multinomial_resize(st, 15_000, check=False); orc.resize(15_000, "multinomial", check=False); upd(); assert _same(st, orc)
    mv(); v, ov = st[100:9000], orc[100:9000]; g.pf_resample(v, "stratified", check=False); ov.resample("stratified", check=False)
    upd(); assert _same(st, orc)
    mv("reweight"); g.pf_resample_blocks(st, 500, "residual", check=False)
    from test_gpu_blocks import oracle_blocks
    oracle_blocks(orc, 500, "residual"); upd(); assert _same(st, orc)
    # a move, then synchronize (enqueues it), then the state
    mv(); st.synchronize(); assert _same(st, orc)
    # set_log_weights / set rows behind a pending move
    lw = np.linspace(-3.0, 0.0, st.n_particles)
    mv("reweight"); st.log_weights = lw; orc.lw = lw.copy(); upd(); assert _same(st, orc)              # (no getter in between: the setter runs the move)
    rows = orc.rows.copy(); rows[:, 0] += 0.25
    mv("move")
    # set rows behind a pending move: the move runs first, then the rows are replaced
    st.traces = rows; orc.rows = rows.copy(); upd(); assert _same(st, orc)
    st.close()


def test_lgssm_proposal_paths_stay_eager(g, o):
    model, ys, st, orc = _pair(g, o, 8000, 2, "lgssm2")
    g.pf_rejuvenate(st, g.move_reweight, (g.locally_optimal_move, ()), 1, method="reweight"); orc.rejuvenate("reweight", 1, proposal=())
    g.pf_update(st, (2,), (None,), ys[1]); orc.update(ys[1])
    assert _same(st, orc)
    g.pf_rejuvenate(st, None, (), 1, method="move"); orc.rejuvenate("move", 1)
    g.pf_update(st, (3,), (None,), ys[2], g.locally_optimal, ()); orc.update(ys[2], proposal=True)      # not the plain propagate
    assert _same(st, orc)
    st.close()


def test_eager_and_lazy_moves_agree_in_a_fresh_process(g, o):
    """GPF_LAZY_MOVE=0 keeps k_move + k_step: the same bits as the fused kernel (BASELINE config 4's loop: ESS-triggered residual + MH)"""
    code = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import gpf_amd as g
m = g.models.bearings4(); ys = g.models.simulate(m, 12)
N = 100_000
st = g.pf_initialize(m, (1,), ys[0], N, seed=5, keep_prev=True)
for t in range(1, 10):
    if g.get_ess(st) < 0.5 * N:
        g.pf_resample(st, "residual", check=False)
        g.pf_rejuvenate(st, None, (), 1, method="move")
    g.pf_update(st, (t + 1,), (None,), ys[t])
h = hashlib.sha256(); h.update(st.parents.tobytes()); h.update(st.traces.tobytes()); h.update(st.log_weights.tobytes())
print(h.hexdigest(), repr(g.get_lml_est(st)))
""" % ROOT
    outs = []
    for v in ("1", "0"):
        env = dict(os.environ, GPF_LAZY_MOVE=v)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) > 64
