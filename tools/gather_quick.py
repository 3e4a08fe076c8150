import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
N = 1_000_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 80)
for method, kw in [("multinomial", {}), ("stratified", {"sort_particles": False})]:
    st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
    for t in range(1, 6):
        g.pf_resample(st, method, check=False, **kw); g.get_ess(st); g.pf_update(st, (t,), (None,), ys[t])
    for k in (g._lib.K_GATHER, g._lib.K_STEP): st.kernel_timing(k, True)
    for t in range(6, 36):
        g.pf_resample(st, method, check=False, **kw); g.get_ess(st); g.pf_update(st, (t,), (None,), ys[t])
    ga = st.kernel_time(g._lib.K_GATHER); sp = st.kernel_time(g._lib.K_STEP)
    for k in (g._lib.K_GATHER, g._lib.K_STEP): st.kernel_timing(k, False)
    st.kernel_timing(g._lib.K_STEP, True)
    for t in range(36, 66):
        g.pf_resample(st, method, check=False, **kw); g.pf_update(st, (t,), (None,), ys[t])
    fs = st.kernel_time(g._lib.K_STEP)
    print(json.dumps(dict(mtype=os.environ.get("GPF_ROWS_MTYPE", "default"), method=method, gather_us=round(ga[0] / ga[1] * 1e3, 2),
                          step_plain_us=round(sp[0] / sp[1] * 1e3, 2), step_fused_gather_us=round(fs[0] / fs[1] * 1e3, 2))))
    st.close()
