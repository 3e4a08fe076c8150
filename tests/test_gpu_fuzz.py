"""Random API sequences on the device filter against the CPU oracle in lockstep: whatever order a host calls the operations in --
resamples of every kind, rejuvenation, getters that force the deferred gather, sub-state views, resizes -- rows, log-weights and
parents stay bit-identical and the scalar getters equal.  (The deferred gather / deferred constant / cached summaries / live views
are a state machine; this walks it at random.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("GPF_FUZZ_SEEDS", "12"))           # GPF_FUZZ_SEEDS=300 for a longer hunt
METHODS = ["multinomial", "residual", "stratified"]


def check(g, st, orc, tag):
    assert np.array_equal(st.traces, orc.rows), tag
    assert np.array_equal(st.log_weights, orc.lw), tag
    assert np.array_equal(st.parents, orc.parents), tag
    assert g.get_ess(st) == orc.effective_sample_size() or (np.isnan(g.get_ess(st)) and np.isnan(orc.effective_sample_size())), tag
    assert g.get_lml_est(st) == orc.log_ml_estimate(), tag


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_api_sequences(g, o, seed):
    rng = np.random.default_rng(1000 + seed)
    name = ["lgssm2", "bearings4", "sv1"][seed % 3]
    model = g.models.by_name(name)
    N = int(rng.choice([37, 1000, 4099, 70_001]))
    T = 40
    ys = g.models.simulate(model, T + 2)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed + 5, keep_prev=True)
    orc = o.OracleFilter(model.model_id, model.params, N, seed + 5, keep_prev=True).initialize(ys[0])
    t = 1
    log = []
    for step in range(T):
        op = rng.choice(["update", "resample", "rejuvenate", "getters", "view", "whole_view", "resize", "nothing"],
                        p=[0.25, 0.22, 0.12, 0.1, 0.12, 0.07, 0.07, 0.05])
        n = st.n_particles
        if op == "update":
            g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t]); t += 1
        elif op == "resample":
            m = str(rng.choice(METHODS)); alpha = None if rng.random() < 0.7 else 0.5
            kw = {"sort_particles": bool(rng.random() < 0.5)} if m == "stratified" else {}
            g.pf_resample(st, m, priority_fn=None if alpha is None else g.Tempering(alpha), check=False, **kw)
            orc.resample(m, priority_alpha=alpha, check=False, **kw)
            op = f"resample {m} {alpha} {kw}"
        elif op == "rejuvenate":
            meth = str(rng.choice(["move", "reweight"])); it = int(rng.integers(1, 3))
            g.pf_rejuvenate(st, g.mh if meth == "move" else g.move_reweight, (), it, method=meth); orc.rejuvenate(meth, it)
            op = f"rejuvenate {meth} {it}"
        elif op == "getters":
            np.testing.assert_allclose(g.mean(st, 0), orc.mean(0), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(g.var(st, 0), orc.var(0), rtol=1e-9, atol=1e-12)
        elif op in ("view", "whole_view"):
            a, b = (0, n) if op == "whole_view" else sorted(int(x) for x in rng.choice(n + 1, 2, replace=False))
            if b - a < 2:
                continue
            sv, ov = st[a:b], orc[a:b]
            sub = rng.choice(["update", "resample", "rejuvenate"])
            if sub == "update":
                g.pf_update(sv, (t + 1,), (None,), ys[t]); ov.update(ys[t])          # (only the view's particles advance)
            elif sub == "resample":
                m = str(rng.choice(METHODS)); kw = {"sort_particles": bool(rng.random() < 0.5)} if m == "stratified" else {}
                g.pf_resample(sv, m, check=False, **kw); ov.resample(m, check=False, **kw)
                assert np.array_equal(sv.parents, ov.parents)
            else:
                g.pf_rejuvenate(sv, g.mh, (), 1); ov.rejuvenate("move", 1)
            assert g.get_ess(sv) == ov.effective_sample_size() and g.get_lml_est(sv) == ov.log_ml_estimate()
            op = f"{op}[{a}:{b}] {sub}"
        elif op == "resize":
            kind = rng.choice(["multinomial", "residual", "optimal", "replicate"])
            if kind == "replicate":
                if n > 40_000:
                    continue
                g.pf_replicate(st, 2); orc.replicate(2)
                g.pf_dereplicate(st, 2); orc.dereplicate(2)
            else:
                n_new = max(8, int(n * rng.choice([0.5, 1.0, 1.5])))
                if kind == "optimal":
                    n_new = min(n_new, n)
                    g.pf_resize(st, n_new, "optimal", check=False); orc.optimal_resize(n_new, check=False)
                else:
                    g.pf_resize(st, n_new, str(kind), check=False); orc.resize(n_new, str(kind), check=False)
            op = f"resize {kind}"
        log.append(op)
        check(g, st, orc, f"seed {seed} {name} N={N} after {log[-6:]}")
    st.close()
