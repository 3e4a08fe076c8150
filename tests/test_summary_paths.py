"""effective_sample_size(state) / log_ml_estimate(state) (src/utils.jl:163-178) when no scan of the current weights is at hand: the
summary {maximum, flags, S = sum q, sum q^2} comes from k_sum_host (every workgroup's partial sums in one line of pinned memory, the host
adds them up), from k_sum_reduce (GPF_SUM_REDUCE=device: workgroup 0 adds them up) or from the weight scan (GPF_SUM_REDUCE=0).  Exact
integers: the three agree bit for bit, and with the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("N", [1, 2, 63, 8191, 8192, 8193, 70_001, 1_000_000, 2_200_007])
def test_ess_and_log_ml_getters_against_the_oracle(g, o, N):
    """sizes around the 8192 weights a workgroup of k_sum_host takes per trip, one and several trips per workgroup (> 256 x 8192)"""
    model = g.models.lgssm2(); ys = g.models.simulate(model, 5)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=2)
    orc = o.OracleFilter(model.model_id, model.params, N, 2).initialize(ys[0])
    for t in range(1, 4):
        assert g.get_ess(st) == orc.effective_sample_size(), (N, t)
        assert g.get_lml_est(st) == orc.log_ml_estimate(), (N, t)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        assert g.get_lml_est(st) == orc.log_ml_estimate(), (N, t)          # (log-ML first: the host copy must carry the maximum too)
        assert g.get_ess(st) == orc.effective_sample_size(), (N, t)
    rng = np.random.default_rng(N)
    for lw in (np.full(N, -np.inf), np.where(rng.random(N) < 0.7, -np.inf, -3.0 * rng.random(N)), -700.0 * rng.random(N), np.zeros(N)):
        st.log_weights = lw; orc.lw = lw.copy()
        e, eo = g.get_ess(st), orc.effective_sample_size()
        assert e == eo or (np.isnan(e) and np.isnan(eo)), N
        l, lo = g.get_lml_est(st), orc.log_ml_estimate()
        assert l == lo or (np.isnan(l) and np.isnan(lo)), N
    sub, osub = st[0:max(1, N // 3)], o.OracleSubState(orc, 0, max(1, N // 3))
    assert g.get_ess(sub) == osub.effective_sample_size()
    st.close()


def test_the_three_summary_paths_agree():
    code = ("import sys, json, numpy as np; sys.path.insert(0, ROOT); import gpf_amd as g\n"
            "m = g.models.bearings4(); ys = g.models.simulate(m, 9); N = 300_007\n"
            "st = g.pf_initialize(m, (1,), ys[0], N, seed=4, keep_prev=True); out = []\n"
            "for t in range(1, 8):\n"
            "    e = g.get_ess(st); out.append(e); out.append(g.get_lml_est(st))\n"
            "    if e < 0.5 * N:\n"
            "        g.pf_resample(st, 'residual', check=False); g.pf_rejuvenate(st, None, (), 1, method='move')\n"
            "    g.pf_update(st, (t + 1,), (None,), ys[t])\n"
            "print(json.dumps(out + [g.get_lml_est(st), g.get_ess(st)]))\n").replace("ROOT", repr(ROOT))
    outs = []
    for mode in ("", "device", "0"):
        env = dict(os.environ); env.pop("GPF_SUM_REDUCE", None)
        if mode:
            env["GPF_SUM_REDUCE"] = mode
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.parametrize("name", ["lgssm2", "bearings4"])
@pytest.mark.parametrize("N", [1, 7, 2048, 2049, 70_001, 1_000_000])
def test_residual_resample_right_after_an_ess_read(g, o, name, N):
    """the README loop's `if effective_sample_size(state) < thresh; pf_resample!(state, :residual)`: the getter left {maximum, flags, S} with
    the host, so the resample runs no weight scan -- k_scan_residual2<DIRECT> converts the weights itself (src/resample.jl:96-115, the
    same fixed-point weights).  Ancestors, log-ML and the state after the update that follows are the oracle's; then the same with the
    log-ML getter in front, with a sub-state, and with extreme weights"""
    model = g.models.by_name(name); ys = g.models.simulate(model, 9)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=12, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, 12, keep_prev=True).initialize(ys[0])
    for t in range(1, 6):
        if t % 2:
            assert g.get_ess(st) == orc.effective_sample_size(), (N, t)
        else:
            assert g.get_lml_est(st) == orc.log_ml_estimate(), (N, t)
        g.pf_resample(st, "residual", check=False if t < 4 else "warn"); orc.resample("residual", check=False)
        assert np.array_equal(st.parents, orc.parents), (N, t)
        assert g.get_lml_est(st) == orc.log_ml_estimate(), (N, t)
        if t == 3:
            g.pf_rejuvenate(st, None, (), 1, method="move"); orc.rejuvenate("move", 1)
        g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
        assert np.array_equal(st.log_weights, orc.lw), (N, t)
    rng = np.random.default_rng(N + 1)
    for lw in (np.where(rng.random(N) < 0.7, -np.inf, -3.0 * rng.random(N)), -300.0 * rng.random(N), np.zeros(N)):
        lw[rng.integers(N)] = 0.5
        st.log_weights = lw; orc.lw = lw.copy()
        assert g.get_ess(st) == orc.effective_sample_size()
        g.pf_resample(st, "residual", check=False); orc.resample("residual", check=False)
        assert np.array_equal(st.parents, orc.parents), N
        g.pf_update(st, (7,), (None,), ys[6]); orc.update(ys[6])
    if N >= 7:
        sub, osub = st[1:N - 2], o.OracleSubState(orc, 1, N - 3)
        assert g.get_ess(sub) == osub.effective_sample_size()
        g.pf_resample(sub, "residual", check=False); osub.resample("residual", check=False)
        assert np.array_equal(st.log_weights, orc.lw) and np.array_equal(st.traces, orc.rows)
    st.close()


def test_direct_and_scanned_residual_agree():
    code = ("import sys, json, numpy as np; sys.path.insert(0, ROOT); import gpf_amd as g\n"
            "m = g.models.bearings4(); ys = g.models.simulate(m, 9); N = 300_007\n"
            "st = g.pf_initialize(m, (1,), ys[0], N, seed=4, keep_prev=True); out = []\n"
            "for t in range(1, 8):\n"
            "    e = g.get_ess(st); out.append(e)\n"
            "    g.pf_resample(st, 'residual', check=False); p = st.parents; out.append(int((p * np.arange(1, N + 1) % 1000003).sum()))\n"
            "    g.pf_update(st, (t + 1,), (None,), ys[t])\n"
            "print(json.dumps(out + [g.get_lml_est(st)]))\n").replace("ROOT", repr(ROOT))
    outs = []
    for mode in ("", "0"):
        env = dict(os.environ); env.pop("GPF_RESIDUAL_DIRECT", None)
        if mode:
            env["GPF_RESIDUAL_DIRECT"] = mode
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1], outs
