"""Checkpoint / resume (include/gpf.h gpf_checkpoint_*): a filter saved in the middle of a run -- with a resample, a lazy move or a lazy search
still deferred -- and loaded into a fresh handle continues bit for bit, against the uninterrupted run and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [pytest.param("lgssm2", "multinomial", None, None, False, marks=pytest.mark.gpu_soak),
         ("lgssm2", "multinomial", None, None, True),          # (True: lazy search pending at the save)
         ("bearings4", "residual", 0.5, "move", False), ("sv1", "stratified", None, "reweight", False),
         pytest.param("object_motion", "residual", 0.9, "move", False, marks=pytest.mark.gpu_soak)]


def _loop(g, st, ys, t0, t1, method, ess, rejuv, save_at=None):
    blob = None
    for t in range(t0, t1):
        if ess is None or g.get_ess(st) < ess * st.n_particles:
            g.pf_resample(st, method, check=False)
            if rejuv:
                g.pf_rejuvenate(st, None, (), 1, method=rejuv)
        if t == save_at:
            blob = st.checkpoint()                               # the resample's gather / the move / the search are still deferred here
        g.pf_update(st, (t + 1,), (None,), ys[t])
    return blob


@pytest.mark.parametrize("name,method,ess,rejuv,lazy", CASES)
def test_checkpoint_resumes_bit_for_bit(g, o, name, method, ess, rejuv, lazy):
    model = g.models.by_name(name); T, N, k = 14, 30_000, 7
    ys = g.models.simulate(model, T)
    a = g.pf_initialize(model, (1,), ys[0], N, seed=11, keep_prev=rejuv is not None)
    if lazy:
        a.set_lazy_search(True)
    blob = _loop(g, a, ys, 1, T, method, ess, rejuv, save_at=k)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=11, keep_prev=rejuv is not None)
    if lazy:
        b.set_lazy_search(True)
    g.pf_update(b, (2,), (None,), ys[1])                          # (a state that has moved on: everything of it is replaced)
    b.restore(blob)
    g.pf_update(b, (k + 1,), (None,), ys[k])
    _loop(g, b, ys, k + 1, T, method, ess, rejuv)
    assert np.array_equal(a.traces, b.traces) and np.array_equal(a.log_weights, b.log_weights) and np.array_equal(a.parents, b.parents)
    assert g.get_lml_est(a) == g.get_lml_est(b) and g.get_ess(a) == g.get_ess(b)
    f = o.OracleFilter(model.model_id, model.params, N, 11, keep_prev=rejuv is not None).initialize(ys[0])
    for t in range(1, T):
        if ess is None or f.effective_sample_size() < ess * N:
            f.resample(method, sort_particles=True, check=False)
            if rejuv:
                f.rejuvenate(rejuv, 1)
        f.update(ys[t])
    assert np.array_equal(b.traces, f.rows) and np.array_equal(b.parents, f.parents) and g.get_lml_est(b) == f.log_ml_estimate()


def test_checkpoint_refuses_another_configuration(g):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    a = g.pf_initialize(model, (1,), ys[0], 5000, seed=3)
    blob = a.checkpoint()
    for kw in (dict(n=5001, seed=3), dict(n=5000, seed=4)):
        b = g.pf_initialize(model, (1,), ys[0], kw["n"], seed=kw["seed"])
        with pytest.raises(g.ErrorException, match="another gpf_config|truncated"):
            b.restore(blob)
    other = g.pf_initialize(g.models.bearings4(), (1,), g.models.simulate(g.models.bearings4(), 2)[0], 5000, seed=3)
    with pytest.raises(g.ErrorException, match="another gpf_config"):
        other.restore(blob)
    with pytest.raises(g.ErrorException, match="not a checkpoint"):
        a.restore(np.zeros(64, np.uint8))
    with pytest.raises(g.ErrorException, match="not a checkpoint"):
        a.restore(np.zeros(blob.size, np.uint8))
    a.restore(bytes(blob))                                        # (bytes work too)
    with pytest.raises(g.ErrorException, match="sub-state"):
        a[0:100].checkpoint()


def test_checkpoint_of_a_shard(g, o):
    """one shard through sharded.py (library engine, a deferred sharded commit pending at the save): the blob resumes on a fresh sharded state"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 10); N = 40_000
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    blob = None
    for t in range(1, 9):
        sharded.pf_resample(a, ("stratified", "multinomial", "residual")[t % 3], check=False)
        if t == 4:
            blob = a.local.checkpoint()
        sharded.pf_update(a, (t + 1,), (None,), ys[t])
    b = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    b.local.restore(blob)
    sharded.pf_update(b, (5,), (None,), ys[4])
    for t in range(5, 9):
        sharded.pf_resample(b, ("stratified", "multinomial", "residual")[t % 3], check=False)
        sharded.pf_update(b, (t + 1,), (None,), ys[t])
    assert np.array_equal(a.local.traces, b.local.traces) and np.array_equal(a.local.log_weights, b.local.log_weights)
    assert sharded.get_lml_est(a) == sharded.get_lml_est(b)
