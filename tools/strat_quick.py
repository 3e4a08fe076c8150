"""Per-kernel times of a resample + update loop (HIP events on the dispatches): python tools/strat_quick.py METHOD [N]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g
method = sys.argv[1] if len(sys.argv) > 1 else "stratified"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
kw = {"sort_particles": False} if method == "stratified" else {}
model = g.models.lgssm2()
ys = g.models.simulate(model, 120)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
def step(t):
    g.pf_resample(st, method, check=False, **kw); g.pf_update(st, (t + 1,), (None,), ys[t])
for t in range(1, 20): step(t)
kids = list(g._lib.KERNEL_NAMES)
for k in kids: st.kernel_timing(k, True)
for t in range(20, 100): step(t)
per = {}
for k in kids:
    ms, cnt = st.kernel_time(k); st.kernel_timing(k, False)
    if cnt: per[g._lib.KERNEL_NAMES[k]] = round(ms / cnt * 1e3, 2)
print(json.dumps({"lib": os.path.basename(g._lib.LIB_PATH), "method": method, "N": N, "kernels_us": per, "lml": g.get_lml_est(st)}))
