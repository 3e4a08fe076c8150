"""k_search times on skewed weights: bearings-only (BASELINE config 4's model), N = 1e6, resampling only when ESS < frac * N,
for each resampler -- the ancestor searches must not depend on the weights being well spread.
usage: python tools/skew_search.py [frac ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
fracs = [float(x) for x in sys.argv[1:]] or [0.5, 0.05, 0.001]
model = g.models.bearings4(); ys = g.models.simulate(model, 260); N = 1_000_000
for method in ("multinomial", "stratified", "residual"):
    for frac in fracs:
        st = g.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=True)
        st.kernel_timing(g._lib.K_SEARCH, True)
        n_res, ess_min = 0, N
        kw = {"sort_particles": False} if method == "stratified" else {}
        for t in range(1, 250):
            ess = g.get_ess(st); ess_min = min(ess_min, ess)
            if ess < frac * N:
                g.pf_resample(st, method, check=False, **kw); n_res += 1
            g.pf_update(st, (t + 1,), (None,), ys[t])
        st.synchronize()
        ms, cnt = st.kernel_time(g._lib.K_SEARCH)
        print(f"{method:12s} resample when ESS < {frac:g} N: {n_res:3d} resamples, min ESS {ess_min:10.1f}, k_search {ms / max(cnt, 1) * 1e3:8.2f} us avg over {cnt} launches")
        st.close()

# synthetic extremes on the LG-SSM rows: all mass on one particle / on 1 % of the particles / two-level weights
import numpy as np
model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
cases = {"one particle": lambda i: np.where(i == 777_777, 0.0, -800.0),
         "1 % of the particles": lambda i: np.where(i % 100 == 0, 0.0, -60.0),
         "half heavy, half 1e-9": lambda i: np.where(i % 2 == 0, 0.0, -20.7)}
for name, f in cases.items():
    for method in ("multinomial", "stratified", "residual"):
        st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
        st.kernel_timing(g._lib.K_SEARCH, True)
        kw = {"sort_particles": False} if method == "stratified" else {}
        for _ in range(5):
            st.log_weights = f(np.arange(N, dtype=np.float64))
            g.pf_resample(st, method, check=False, **kw)
        st.synchronize()
        ms, cnt = st.kernel_time(g._lib.K_SEARCH)
        print(f"{name:24s} {method:12s} k_search {ms / max(cnt, 1) * 1e3:8.2f} us")
        st.close()
