"""gpf_step_ess / pf_step_ess: one iteration of the reference's README loop (README.md:66-77) -- `if effective_sample_size(state) <
threshold; pf_resample!; pf_rejuvenate!; end; pf_update!` -- in ONE call, with the ESS verdict also formed on the device and the propagate
enqueued speculatively behind it.  The contract: bit-identical to the four separate calls (the host-decided loop) and to the
oracle, whatever path the call takes inside (speculative, or the plain sequence for sub-states / pending work / cached summaries)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # model, N, method, rejuvenate, threshold, T
    ("bearings4", 50_000, "residual", "move", 0.5, 30),          # BASELINE configs[3]'s loop
    ("bearings4", 4097, "stratified", "reweight", 0.7, 25),
    ("object_motion", 100, "residual", "move", 0.5, 12),         # configs[0]: the README's own model and loop
    ("lgssm2", 30_000, "multinomial", None, 0.9, 20),            # no rejuvenation (keep_prev = False)
    ("sv1", 20_001, "multinomial_sorted", "reweight", 0.95, 20),
    ("lgssm2", 5000, "multinomial", None, 0.0, 6),               # never resamples
    ("lgssm2", 5000, "residual", None, 1.1, 6),                  # always resamples (ESS <= N < 1.1 N)
]


def separate_calls(g, model, ys, N, method, rejuv, thr, T, seed=11):
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=rejuv is not None)
    res = []
    for t in range(1, T):
        go = g.get_ess(st) < thr * N
        if go:
            g.pf_resample(st, method, check=False)
            if rejuv:
                g.pf_rejuvenate(st, None, (), 1, method=rejuv)
        g.pf_update(st, (t + 1,), (None,), ys[t])
        res.append(go)
    return st, res


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-{c[2]}-{c[3]}-{c[4]}" for c in CASES])
def test_step_ess_equals_the_separate_calls_and_the_oracle(g, o, case):
    name, N, method, rejuv, thr, T = case
    model = g.models.by_name(name); ys = g.models.simulate(model, T)
    a, res_a = separate_calls(g, model, ys, N, method, rejuv, thr, T)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=11, keep_prev=rejuv is not None)
    orc = o.OracleFilter(model.model_id, model.params, N, 11, keep_prev=rejuv is not None).initialize(ys[0])
    res_b = []
    for t in range(1, T):
        res_b.append(g.pf_step_ess(b, (t + 1,), (None,), ys[t], ess_threshold=thr, method=method, rejuvenate=rejuv, check=False))
        if orc.effective_sample_size() < thr * N:
            orc.resample(method, check=False)
            if rejuv:
                orc.rejuvenate(rejuv, 1)
        orc.update(ys[t])
        if t % 7 == 0:                                  # getters in between (they leave summaries behind: the next call takes the plain sequence)
            assert g.get_ess(b) == orc.effective_sample_size() and g.get_lml_est(b) == orc.log_ml_estimate()
    assert res_a == res_b
    assert any(res_b) or thr == 0.0
    for st in (a, b):
        assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw) and np.array_equal(st.parents, orc.parents)
        assert g.get_lml_est(st) == orc.log_ml_estimate()


def test_step_ess_mixed_with_the_other_entry_points(g, o):
    """a pending resample / a pending lazy move / a view / invalid weights in front of the call: every one of them takes the plain sequence or
    finishes the pending work first; results stay the oracle's"""
    import warnings
    model = g.models.bearings4(); N = 20_000; ys = g.models.simulate(model, 12)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=3, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, 3, keep_prev=True).initialize(ys[0])

    def both(t, thr=0.5):
        r = g.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=thr, method="residual", rejuvenate="move", check=False)
        go = orc.effective_sample_size() < thr * N
        if go:
            orc.resample("residual", check=False); orc.rejuvenate("move", 1)
        orc.update(ys[t])
        assert r == go
    both(1); both(2)
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)          # a pending gather in front of the call
    both(3, 1.1)
    g.pf_rejuvenate(st, None, (), 1, method="reweight"); orc.rejuvenate("reweight", 1)               # a pending lazy move
    both(4)
    v = st[100:9000]; ov = orc[100:9000]                                                           # a sub-state: always the plain sequence
    r = g.pf_step_ess(v, (6,), (None,), ys[5], ess_threshold=1.1, method="stratified", check=False)
    ov.resample("stratified", check=False); ov.update(ys[5])
    assert r is True
    both(6); both(7, 1.1)
    lw = st.log_weights; lw[:] = -np.inf; st.log_weights = lw; orc.lw[:] = -np.inf                 # invalid weights: ESS is NaN, NaN < thr is false
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        both(8)
    assert np.array_equal(st.traces, orc.rows) and np.array_equal(st.log_weights, orc.lw) and np.array_equal(st.parents, orc.parents)
    with pytest.raises(g.ErrorException):
        g.pf_step_ess(st, (9,), (None,), ys[8], method="nonsense")
    with pytest.raises(g.ErrorException):
        g.pf_step_ess(st, (9,), (None,), ys[8], rejuvenate="nonsense")


def test_step_ess_speculation_can_be_switched_off(g, o, monkeypatch):
    """GPF_STEP_SPECULATE=0: the plain sequence inside the call (A/B measurements); same results -- hashed in fresh processes"""
    import subprocess, sys, os, hashlib
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import gpf_amd as g
model = g.models.bearings4(); ys = g.models.simulate(model, 40); N = 30_000
st = g.pf_initialize(model, (1,), ys[0], N, seed=5, keep_prev=True)
res = [g.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=0.5, method="residual", rejuvenate="move", check=False) for t in range(1, 40)]
h = hashlib.sha256(); h.update(st.traces.tobytes()); h.update(st.log_weights.tobytes()); h.update(st.parents.tobytes()); h.update(repr(res).encode())
print(h.hexdigest(), sum(res), g.get_lml_est(st))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for spec in ("1", "0"):
        env = dict(os.environ, GPF_STEP_SPECULATE=spec)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and int(outs[0].split()[1]) > 0, outs
