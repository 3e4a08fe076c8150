"""Residual resample on weight vectors no filter step would produce, N = 1e6: time per resample (HIP events: scan bucket = k_scan + k_scan_residual2,
search bucket = k_search<1>), head written by the scan (default) or looked up by the search (GPF_RESIDUAL_HEAD=search)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
N = 1_000_000
i = np.arange(N, dtype=np.float64); rng = np.random.default_rng(1)
cases = {"filter step": None, "all equal": np.zeros(N), "one particle": np.where(i == 777_777, 0.0, -800.0), "1 % heavy": np.where(i % 100 == 0, 0.0, -60.0),
         "ten heavy": np.where(i % 100_000 == 0, 0.0, -60.0), "descending ramp": -i * 1e-5, "first half -inf": np.where(i < N / 2, -np.inf, -rng.random(N))}
m = g.models.lgssm2(); ys = g.models.simulate(m, 3)
for name, lw in cases.items():
    st = g.pf_initialize(m, (1,), ys[0], N, seed=3)
    g.pf_update(st, (2,), (None,), ys[1])
    for rep in range(12):
        if rep == 2:
            st.kernel_timing(g._lib.K_SCAN, True); st.kernel_timing(g._lib.K_SEARCH, True)
        if lw is not None:
            st.log_weights = lw
        else:
            g.pf_update(st, (rep + 3,), (None,), ys[2])
        g.pf_resample(st, "residual", check=False)
        st.synchronize()
    a, ac = st.kernel_time(g._lib.K_SCAN); b, bc = st.kernel_time(g._lib.K_SEARCH)
    print(json.dumps({"case": name, "head": os.environ.get("GPF_RESIDUAL_HEAD", "scan"), "scans_us_per_resample": round(a / 10 * 1e3, 1), "search_us": round(b / max(bc, 1) * 1e3, 1)}))
    st.close()
