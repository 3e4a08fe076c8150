"""Random API sequences on the device filter against the CPU oracle in lockstep: whatever order a host calls the operations in --
resamples of every kind, rejuvenation, getters that force the deferred gather, sub-state views, resizes -- rows, log-weights and
parents stay bit-identical and the scalar getters equal.  (The deferred gather / deferred constant / cached summaries / live views
are a state machine; this walks it at random.)"""
import os
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("GPF_FUZZ_SEEDS", "12"))           # GPF_FUZZ_SEEDS=300 for a longer hunt
OFFSET = int(os.environ.get("GPF_FUZZ_OFFSET", "0"))             # ... GPF_FUZZ_OFFSET=100000 for other sequences
METHODS = ["multinomial", "residual", "stratified"]
RES_METHODS = METHODS + ["multinomial_sorted"]              # whole-filter and view resamples also draw the opt-in sorted multinomial
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_blocks import oracle_blocks, oracle_rejuvenate_blocks, oracle_update_blocks  # noqa: E402


def same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def both(dev, orc, tag):
    """run the device call and the oracle call; both raise "Invalid weights" or neither does.  True: both raised"""
    err = []
    for f in (dev, orc):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                f()
            err.append(None)
        except Exception as e:                                          # noqa: BLE001
            err.append(str(e))
    assert (err[0] is None) == (err[1] is None), (err, tag)
    if err[0] is not None:
        assert "Invalid weights" in err[0] and "Invalid weights" in err[1], (err, tag)
    return err[0] is not None


def check(g, st, orc, tag):
    assert np.array_equal(st.traces, orc.rows, equal_nan=True), tag
    assert np.array_equal(st.log_weights, orc.lw, equal_nan=True), tag
    assert np.array_equal(st.parents, orc.parents), tag
    assert same(g.get_ess(st), orc.effective_sample_size()), tag
    assert same(g.get_lml_est(st), orc.log_ml_estimate()), tag


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_api_sequences(g, o, seed):
    rng = np.random.default_rng(1000 + OFFSET + seed)
    name = ["lgssm2", "bearings4", "sv1"][seed % 3]
    model = g.models.by_name(name)
    N = int(rng.choice([1, 2, 5, 37, 1000, 4099, 70_001, 300_000], p=[0.05, 0.05, 0.08, 0.2, 0.2, 0.2, 0.17, 0.05]))
    T = 40
    ys = g.models.simulate(model, T + 2)
    hist = seed % 4 == 3                                              # every fourth run keeps the trajectory store
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed + 5, keep_prev=True, history=T + 2 if hist else 0)
    orc = o.OracleFilter(model.model_id, model.params, N, seed + 5, keep_prev=True, history=hist).initialize(ys[0])
    st.set_lazy_search(bool(seed % 2))                                # every other run leaves the multinomial search to the update that follows
    t = 1
    log = []
    blk = None                                                        # (block size, observation rows) while the latest observations are per block
    for step in range(T):
        op = rng.choice(["update", "resample", "rejuvenate", "getters", "view", "whole_view", "resize", "set_weights", "blocks", "step_ess", "checkpoint"],
                        p=[0.15, 0.15, 0.10, 0.08, 0.11, 0.07, 0.07, 0.07, 0.07, 0.08, 0.05])
        n = st.n_particles
        if op == "checkpoint" and (hist or blk is not None):
            op = "getters"                                              # (the trajectory store and per-block observations are not part of a blob)
        if op == "checkpoint":
            # gpf_checkpoint_save with whatever is deferred at this point, then either a FRESH handle takes the blob and the old one goes away, or this
            # one moves on (an update the oracle never sees) and is set back; the oracle is not told: the sequence must go on bit for bit
            blob = st.checkpoint()
            if rng.random() < 0.5:
                st2 = g.pf_initialize(model, (1,), ys[0], n, seed=seed + 5, keep_prev=True)
                st2.set_lazy_search(bool(seed % 2)); st2.restore(blob)
                st.close(); st = st2
                log.append("checkpoint -> fresh handle")
            else:
                g.pf_update(st, (t + 1,), (None,), ys[t])
                if rng.random() < 0.5:
                    g.pf_resample(st, "multinomial", check=False)
                st.restore(blob)
                log.append("checkpoint -> moved on -> restored")
            check(g, st, orc, f"seed {seed} {name} N={N} after {log[-6:]}")
            continue
        if op == "blocks" and hist:
            with pytest.raises(g.ErrorException):                       # (no block-wise steps on a filter with a trajectory store)
                g.pf_resample_blocks(st, 64, "multinomial", check=False)
            op = "getters"
        if op == "blocks":
            # many small filters in one state (gpf_k_block.hpp): block-wise resample / update with per-block data / rejuvenate
            nb = int(rng.choice([1, 7, 64, 100, 128, 300, 512, 999, 2048]))
            nblk = (n + nb - 1) // nb
            sub = str(rng.choice(["resample", "resample_ess", "update", "rejuvenate"]))
            m = str(rng.choice(METHODS)); sp = bool(rng.integers(2))
            if sub in ("resample", "resample_ess"):
                fr = None if sub == "resample" else float(rng.choice([0.3, 0.7, 1.5]))
                al = None if rng.random() < 0.6 else float(rng.choice([0.5, 2.0]))
                out = {}
                if both(lambda: out.update(d=g.pf_resample_blocks(st, nb, m, priority_fn=None if al is None else g.Tempering(al), ess_frac=fr, sort_particles=sp, check=False)),
                        lambda: out.update(o=oracle_blocks(orc, nb, m, ess_frac=fr, sort_particles=sp, priority_alpha=al)), log[-4:]):
                    st.close()
                    return                                              # NaN weights in some block: the run ends
                assert out["d"] == out["o"].sum()
            elif sub == "update":
                obs = np.asarray(ys[t], np.float64)[None, :] + 0.2 * rng.standard_normal((nblk, len(ys[t])))
                g.pf_update_blocks(st, (t + 1,), (None,), obs, nb); oracle_update_blocks(orc, nb, obs)
                blk = (nb, obs); t += 1
            else:
                if blk is None or blk == "lost":
                    with pytest.raises(g.ErrorException):
                        g.pf_rejuvenate_blocks(st, None, (), 1)
                else:
                    meth = str(rng.choice(["move", "reweight"]))
                    g.pf_rejuvenate_blocks(st, None, (), 1, method=meth); oracle_rejuvenate_blocks(orc, blk[0], blk[1], meth)
            log.append(f"blocks {sub} nb={nb} {m} sort={sp}")
            check(g, st, orc, f"seed {seed} {name} N={N} after {log[-6:]}")
            continue
        if hist and op in ("resize", "view", "whole_view"):
            if op != "resize":
                with pytest.raises(g.ErrorException):                   # one ancestor map per step for the whole filter: no sub-states
                    st[0:max(1, n // 2)]
            op = "getters"                                              # (nor can such a filter be resized)
        if op == "step_ess" and blk is not None:
            op = "update"                                               # (per-block observations: the one-call step takes one observation for all)
        if op == "step_ess":
            # one iteration of the README loop in ONE call (gpf_step_ess: the verdict on the device, the propagate speculatively behind it --
            # or the plain sequence when a resample / move is pending, a summary is cached, the filter keeps a trajectory store)
            thr = float(rng.choice([0.0, 0.3, 0.7, 1.1])); m = str(rng.choice(RES_METHODS)); rj = [None, "move", "reweight"][int(rng.integers(3))]
            kw = {"sort_particles": bool(rng.random() < 0.5)} if m == "stratified" else {}
            out = {}

            def oracle_step():
                go = orc.effective_sample_size() < thr * n
                if go:
                    orc.resample(m, check="warn", **kw)
                    if rj:
                        orc.rejuvenate(rj, 1)
                orc.update(ys[t])
                out["o"] = bool(go)
            if both(lambda: out.update(d=g.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=thr, method=m, rejuvenate=rj, check="warn", **kw)),
                    oracle_step, log[-4:]):
                st.close()
                return
            assert out["d"] == out["o"], (thr, m, rj, log[-4:])
            t += 1
            op = f"step_ess {thr} {m} {rj} -> {out['d']}"
        elif op == "update":
            if name == "lgssm2" and rng.random() < 0.3:               # the native locally optimal proposal (update.jl:79-96)
                g.pf_update(st, (t + 1,), (None,), ys[t], g.locally_optimal, ()); orc.update(ys[t], proposal=True)
                op = "update (proposal)"
            elif rng.random() < 0.1:                                    # an observation far outside anything the model expects
                with np.errstate(over="ignore"):
                    y = np.asarray(ys[t], np.float64) * float(rng.choice([1e3, 1e150, 1e308, -1e308]))      # (may overflow to inf: wanted)
                g.pf_update(st, (t + 1,), (None,), y); orc.update(y)
                op = "update (extreme observation)"
            else:
                g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
            t += 1
            blk = None                                                  # one observation for all particles again
        elif op == "resample":
            m = str(rng.choice(RES_METHODS)); alpha = None if rng.random() < 0.6 else (0.5 if rng.random() < 0.6 else "closure")
            kw = {"sort_particles": bool(rng.random() < 0.5)} if m == "stratified" else {}
            closure = lambda w: 0.25 * w - 0.125 * np.abs(w) ** 0.5      # an arbitrary priority_fn: evaluated by the host (resample.jl:51-52)
            if alpha == "closure":
                dev = lambda: g.pf_resample(st, m, priority_fn=closure, check="warn", **kw)
                lp = closure(orc.lw)
                orf = lambda: orc.resample(m, log_priorities=lp, check="warn", **kw)
            else:
                dev = lambda: g.pf_resample(st, m, priority_fn=None if alpha is None else g.Tempering(alpha), check="warn", **kw)
                orf = lambda: orc.resample(m, priority_alpha=alpha, check="warn", **kw)
            # check = :warn, so that NaN weights (-inf weights under a priority: lw - lp = NaN) raise on both sides
            if both(dev, orf, log[-4:]):
                st.close()
                return                                                  # error("Invalid weights."): the run ends, as a host's would
            op = f"resample {m} {alpha} {kw}"
        elif op == "rejuvenate":
            meth = str(rng.choice(["move", "reweight"])); it = int(rng.integers(1, 3))
            if blk == "lost":                                           # resized after a block-wise update: no current observation until the next update
                with pytest.raises(g.ErrorException):
                    g.pf_rejuvenate(st, g.mh, (), 1)
                log.append("rejuvenate refused"); continue
            if blk is not None:                                         # after a block-wise update the move uses every block's own observation
                g.pf_rejuvenate(st, g.mh if meth == "move" else g.move_reweight, (), it, method=meth)
                oracle_rejuvenate_blocks(orc, blk[0], blk[1], meth, n_iters=it)
                meth = f"{meth} (per-block observations)"
            elif name == "lgssm2" and meth == "reweight" and rng.random() < 0.5:       # move_reweight(trace, proposal, args), rejuvenate.jl:134-148
                g.pf_rejuvenate(st, g.move_reweight, (g.locally_optimal_move, ()), it, method="reweight"); orc.rejuvenate("reweight", it, proposal=())
                meth = "reweight (locally optimal proposal)"
            elif name == "lgssm2" and meth == "move" and rng.random() < 0.5:           # mh(trace, proposal, args) under pf_move_accept!, rejuvenate.jl:40-53
                g.pf_rejuvenate(st, g.mh, (g.locally_optimal_move, ()), it, method="move", count=True); orc.rejuvenate("move", it, proposal=())
                assert st.n_accepted == orc.n_accepted
                meth = "move (locally optimal proposal)"
            else:
                g.pf_rejuvenate(st, g.mh if meth == "move" else g.move_reweight, (), it, method=meth); orc.rejuvenate(meth, it)
            op = f"rejuvenate {meth} {it}"
        elif op == "set_weights":
            # ParticleFilterState(trs, ws): weights no filter step would produce -- equal, collapsed, with zeros, far apart
            kind = str(rng.choice(["equal", "one heavy", "some -inf", "wide", "two values", "all -inf"]))
            i = np.arange(n, dtype=np.float64)
            lw = {"equal": np.full(n, -3.25), "one heavy": np.where(i == int(rng.integers(n)), 0.0, -745.0),
                  "some -inf": np.where(rng.random(n) < 0.7, -np.inf, -rng.random(n)), "wide": -700.0 * rng.random(n),
                  "two values": np.where(i % 3 == 0, -1.0, -1.0 - 2.0 ** -40), "all -inf": np.full(n, -np.inf)}[kind]
            st.log_weights = lw; orc.lw = lw.copy()
            op = f"set_weights {kind}"
        elif op == "getters":
            if hist:                                                    # a past choice along the surviving ancestry (README.md:97-104)
                step_q = int(rng.integers(1, t + 1)); c = int(rng.integers(model.dim))
                assert np.array_equal(st.history_column(step_q, c), orc.history_column(step_q, c)), (step_q, c, log[-6:])
            assert same(g.mean(st, 0), orc.mean(0)) and same(g.var(st, 0), orc.var(0)), log[-6:]     # (the summation order is part of the spec)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                np.testing.assert_allclose(g.get_norm_weights(st), orc.norm_weights(), rtol=1e-12, atol=0, equal_nan=True)
                np.testing.assert_allclose(g.get_log_norm_weights(st), orc.log_norm_weights(), rtol=1e-12, atol=1e-12, equal_nan=True)
            out = {}                                                    # Gen.sample_unweighted_traces (utils.jl:189-194)
            if not both(lambda: out.update(d=g.sample_unweighted_traces(st, 64, return_indices=True)),
                        lambda: out.update(o=orc.sample_unweighted(64)), log[-4:]):
                assert np.array_equal(out["d"][1], out["o"][1]) and np.array_equal(out["d"][0], out["o"][0], equal_nan=True)
        elif op in ("view", "whole_view"):
            a, b = (0, n) if op == "whole_view" else sorted(int(x) for x in rng.choice(n + 1, 2, replace=False))
            if b - a < 2:
                continue
            stride = 1 if (op == "whole_view" or rng.random() < 0.5) else int(rng.integers(2, 8))      # state[a:stride:b] (view.jl:35-48)
            if len(range(a, b, stride)) < 2:
                continue
            if op == "view" and rng.random() < 0.3:                    # state[idxs] over an arbitrary vector of distinct indices (view.jl:35-48)
                ix = rng.permutation(n)[:max(2, int(rng.integers(2, min(n, 5000) + 1)))]
                sv, ov = st[ix], orc[ix]
                stride = f"index vector of {ix.size}"
            else:
                sv, ov = st[a:b:stride], orc[a:b:stride]
            sub = rng.choice(["update", "resample", "rejuvenate"])
            if sub == "rejuvenate" and blk == "lost":
                sub = "resample"
            if sub == "rejuvenate" and blk is not None:
                with pytest.raises(g.ErrorException):                   # the filter's latest observations are per block: the view cannot know which
                    g.pf_rejuvenate(sv, g.mh, (), 1)
                sub = "resample"
            if sub == "update":
                g.pf_update(sv, (t + 1,), (None,), ys[t]); ov.update(ys[t])          # (only the view's particles advance)
            elif sub == "resample":
                m = str(rng.choice(RES_METHODS)); kw = {"sort_particles": bool(rng.random() < 0.5)} if m == "stratified" else {}
                if both(lambda: g.pf_resample(sv, m, check="warn", **kw), lambda: ov.resample(m, check="warn", **kw), log[-4:]):
                    st.close()
                    return
                assert np.array_equal(sv.parents, ov.parents)
            else:
                g.pf_rejuvenate(sv, g.mh, (), 1); ov.rejuvenate("move", 1)
            assert same(g.get_ess(sv), ov.effective_sample_size()) and same(g.get_lml_est(sv), ov.log_ml_estimate())
            op = f"{op}[{a}:{b}:{stride}] {sub}"
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
                    bad = both(lambda: g.pf_resize(st, n_new, "optimal", check="warn"), lambda: orc.optimal_resize(n_new, check="warn"), log[-4:])
                else:
                    bad = both(lambda: g.pf_resize(st, n_new, str(kind), check="warn"), lambda: orc.resize(n_new, str(kind), check="warn"), log[-4:])
                if bad:
                    st.close()
                    return
            if blk is not None:
                blk = "lost"                                            # the per-block observations do not survive a change of the particle count
            op = f"resize {kind}"
        log.append(op)
        check(g, st, orc, f"seed {seed} {name} N={N} after {log[-6:]}")
    st.close()


@pytest.mark.parametrize("seed", range(max(4, N_SEEDS // 3)))
def test_random_api_sequences_discrete_latent_models(g, o, seed):
    """the same walk on the models with a discrete latent (README's object_motion, the reference tests' line_model): plain and
    STRATIFIED initialisation / updates (src/initialize.jl:92-109, src/update.jl:193-210; both layouts), every resampler, MH,
    views"""
    rng = np.random.default_rng(5000 + OFFSET + seed)
    name = ["object_motion", "line_model"][seed % 2]
    model = g.models.by_name(name)
    N = int(rng.choice([10, 100, 1000, 5001, 40_000]))
    T = 25
    if name == "line_model":
        ys = [g.models.line_obs(0)] + [g.models.line_obs(t, float(rng.integers(-2, 3))) + np.array([rng.normal(), 0.0]) for t in range(1, T + 2)]
    else:
        ys = g.models.simulate(model, T + 2)
    strata0 = [0.0, 1.0] if name == "object_motion" else [-2.0, -1.0, 0.0, 1.0, 2.0]      # moving / slope
    strata_t = [0.0, 1.0]                                                                 # moving / outlier
    lay0 = str(rng.choice(["contiguous", "interleaved"]))
    if rng.random() < 0.5:
        st = g.pf_initialize(model, (0,), ys[0], strata0, N, seed=seed + 3, keep_prev=True, layout=lay0)
        orc = o.OracleFilter(model.model_id, model.params, N, seed + 3, keep_prev=True).initialize(ys[0], strata=strata0, layout=lay0)
    else:
        st = g.pf_initialize(model, (0,), ys[0], N, seed=seed + 3, keep_prev=True)
        orc = o.OracleFilter(model.model_id, model.params, N, seed + 3, keep_prev=True).initialize(ys[0])
    check(g, st, orc, "init")
    t, log = 1, []
    for step in range(T):
        op = str(rng.choice(["update", "update_strata", "resample", "rejuvenate", "view_strata", "getters"], p=[0.25, 0.2, 0.25, 0.12, 0.08, 0.1]))
        n = st.n_particles
        if op == "update":
            g.pf_update(st, (t,), (None,), ys[t]); orc.update(ys[t]); t += 1
        elif op == "update_strata":
            lay = str(rng.choice(["contiguous", "interleaved"]))
            k = [strata_t, [1.0, 0.0, 1.0]][int(rng.integers(2))]
            g.pf_update(st, (t,), (None,), ys[t], k, layout=lay); orc.update(ys[t], strata=k, layout=lay); t += 1
            op = f"update_strata {k} {lay}"
        elif op == "resample":
            m = str(rng.choice(METHODS)); kw = {"sort_particles": bool(rng.random() < 0.5)} if m == "stratified" else {}
            if both(lambda: g.pf_resample(st, m, check="warn", **kw), lambda: orc.resample(m, check="warn", **kw), log[-4:]):
                st.close()
                return
            op = f"resample {m} {kw}"
        elif op == "rejuvenate":
            if name == "line_model" and t > 1 and rng.random() < 0.5:   # outlier_propose = bernoulli(q) on the current step (test/rejuvenate.jl:19-27)
                import math
                q = float(rng.choice([0.9, 0.5, 0.1]))
                g.pf_rejuvenate(st, g.move_reweight, (g.outlier_propose(q), (t - 1,)), 1, method="reweight")
                orc.rejuvenate("reweight", 1, proposal=(q, math.log(q), math.log1p(-q)))
                op = f"rejuvenate reweight outlier_propose({q})"
            else:
                g.pf_rejuvenate(st, g.mh, (), 1); orc.rejuvenate("move", 1)
        elif op == "view_strata":
            a, b = sorted(int(x) for x in rng.choice(n + 1, 2, replace=False))
            if b - a < 4:
                continue
            stride = int(rng.choice([1, 1, 2, 5]))
            if len(range(a, b, stride)) < 4:
                continue
            g.pf_update(st[a:b:stride], (t,), (None,), ys[t], strata_t, layout="contiguous"); orc[a:b:stride].update(ys[t], strata=strata_t, layout="contiguous")
            op = f"view_strata[{a}:{b}:{stride}]"
        else:
            assert same(g.mean(st, 0), orc.mean(0)), log[-6:]
        log.append(op)
        check(g, st, orc, f"seed {seed} {name} N={N} after {log[-6:]}")
    st.close()
