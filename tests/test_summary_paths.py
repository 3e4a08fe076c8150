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
