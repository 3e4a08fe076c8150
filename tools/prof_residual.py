import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
method = os.environ.get("METHOD", "residual")
model = g.models.lgssm2(); ys = g.models.simulate(model, 130)
st = g.pf_initialize(model, (1,), ys[0], 1_000_000, seed=1)
kw = {"sort_particles": os.environ.get("SORT") == "1"} if method == "stratified" else {}
for t in range(1, 121):
    g.pf_resample(st, method, check=False, **kw); g.pf_update(st, (t + 1,), (None,), ys[t])
st.synchronize()
