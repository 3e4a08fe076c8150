"""Stand-alone resample gather (k_gather) timing: i.i.d. multinomial vs monotone (stratified / residual) ancestors."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g

for name, keep, N in [("lgssm2", False, 1_000_000), ("lgssm2", True, 1_000_000), ("bearings4", True, 1_000_000), ("lgssm2", False, 8_000_000)]:
    model = g.models.by_name(name); ys = g.models.simulate(model, 40)
    for method, kw in [("multinomial", {}), ("stratified", {"sort_particles": False}), ("residual", {})]:
        st = g.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=keep)
        for t in range(1, 6):
            g.pf_resample(st, method, check=False, **kw); g.get_ess(st); g.pf_update(st, (t,), (None,), ys[t])
        st.kernel_timing(g._lib.K_GATHER, True)
        for t in range(6, 36):
            g.pf_resample(st, method, check=False, **kw); g.get_ess(st)      # ESS forces the stand-alone gather
            g.pf_update(st, (t,), (None,), ys[t])
        ms, cnt = st.kernel_time(g._lib.K_GATHER)
        W = st.row_width
        us = ms / cnt * 1e3
        by = (4 + 8 * W + 8 * W + 8) * N
        print(json.dumps(dict(model=name, W=W, N=N, method=method, gather_us=round(us, 2), alg_MB=by / 1e6,
                              TBps=round(by / us / 1e6, 3), frac_of_8TBps=round(by / us / 1e6 / 8, 3))), flush=True)
        st.close()
