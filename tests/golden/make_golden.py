"""Generates the golden vectors in this directory from the CPU oracle (NOT from the reference: the
reference is Julia and cannot run in the build container -- SURVEY.md §8c -- so these freeze the
build's own numerical spec; random-stream parity with the reference stays unpinned).

    python tests/golden/make_golden.py

Each fixture is a small full-run trace (N=64, T=6): per-step data, and after every operation the
particle rows, log-weights, 1-based parents, ESS and log-ML estimate, for one model x resampling method
(+ priorities, + rejuvenation).  < 100 KB each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpf_amd as g                      # noqa: E402  (model descriptors only; no GPU needed)
from oracle import oracle as o           # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
N, T, SEED = 64, 6, 20240001

CASES = [
    # name, model, method, sort, alpha, rejuvenate (method, iters) or None, ess_triggered
    ("lgssm2_multinomial", "lgssm2", "multinomial", True, None, None, False),
    ("lgssm2_stratified_sorted", "lgssm2", "stratified", True, None, None, False),
    ("lgssm2_stratified_unsorted_prio", "lgssm2", "stratified", False, 0.5, None, False),
    ("lgssm2_residual", "lgssm2", "residual", True, None, None, False),
    ("bearings4_residual_mh_ess", "bearings4", "residual", True, None, ("move", 1), True),
    ("sv1_multinomial_reweight", "sv1", "multinomial", True, None, ("reweight", 1), False),
    ("object_motion_residual_mh_ess", "object_motion", "residual", True, None, ("move", 1), True),
    # round 5: the sorted-uniforms multinomial (opt-in), mh with a native proposal (Gen.mh(trace, proposal, args): the LG-SSM's locally optimal
    # proposal, 2 sweeps), and the README loop's iteration as ONE call (pf_step_ess on the device against the oracle's separate calls)
    ("lgssm2_multinomial_sorted", "lgssm2", "multinomial_sorted", True, None, None, False),
    ("lgssm2_stratified_mh_proposal", "lgssm2", "stratified", False, None, ("move_proposal", 2), False),
    ("bearings4_residual_mh_step_ess", "bearings4", "residual", False, None, ("move", 1), "one_call"),
]


def run_case(model_name, method, sort, alpha, rejuv, ess_trig, backend):
    """backend: 'oracle' or 'hip'; returns dict of arrays (same code path drives both for the tests)."""
    model = g.models.by_name(model_name)
    ys = g.models.simulate(model, T)
    keep = rejuv is not None
    rec = {"ys": ys}
    if backend == "oracle":
        f = o.OracleFilter(model.model_id, model.params, N, SEED, keep_prev=keep).initialize(ys[0])
        snap = lambda: (f.rows.copy(), f.lw.copy(), f.parents.copy(), f.effective_sample_size(), f.log_ml_estimate())
        ess = f.effective_sample_size
        res = lambda: f.resample(method, priority_alpha=alpha, sort_particles=sort, check=False)
        rej = (lambda: f.rejuvenate("move", rejuv[1], proposal=())) if rejuv and rejuv[0] == "move_proposal" else (lambda: f.rejuvenate(*rejuv))
        upd = lambda y: f.update(y)
    else:
        st = g.pf_initialize(model, (1,), ys[0], N, seed=SEED, keep_prev=keep)
        snap = lambda: (st.traces, st.log_weights, st.parents, g.get_ess(st), g.get_lml_est(st))
        ess = lambda: g.get_ess(st)
        pf = None if alpha is None else g.Tempering(alpha)
        kw = dict(sort_particles=sort) if method == "stratified" else {}
        res = lambda: g.pf_resample(st, method, priority_fn=pf, check=False, **kw)
        if rejuv and rejuv[0] == "move_proposal":
            rej = lambda: g.pf_move_accept(st, g.mh, (g.locally_optimal_move,), rejuv[1])
        else:
            rej = lambda: g.pf_rejuvenate(st, None, (), rejuv[1], method=rejuv[0])
        upd = lambda y: g.pf_update(st, (0,), (None,), y)
    steps = []
    steps.append(snap())
    for t in range(1, T):
        if ess_trig == "one_call":                   # one snapshot per iteration: the device runs the whole iteration inside one call
            if backend == "hip":
                g.pf_step_ess(st, (0,), (None,), ys[t], ess_threshold=0.5, method=method, rejuvenate=rejuv[0], n_iters=rejuv[1],
                              check=False, sort_particles=sort)
            else:
                if ess() < 0.5 * N:
                    res(); rej()
                upd(ys[t])
            steps.append(snap())
            continue
        if (not ess_trig) or ess() < 0.5 * N:
            res()
            if rejuv:
                rej()
        steps.append(snap())
        upd(ys[t])
        steps.append(snap())
    rec["rows"] = np.stack([s[0] for s in steps])
    rec["lw"] = np.stack([s[1] for s in steps])
    rec["parents"] = np.stack([s[2] for s in steps])
    rec["ess"] = np.array([s[3] for s in steps])
    rec["lml"] = np.array([s[4] for s in steps])
    return rec


if __name__ == "__main__":
    for name, *cfg in CASES:
        rec = run_case(*cfg, backend="oracle")
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **rec)
        print(name, os.path.getsize(path), "bytes")
