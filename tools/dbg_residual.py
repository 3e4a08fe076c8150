import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
from oracle import oracle as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 594 * 2048 + 3
model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
st = g.pf_initialize(model, (1,), ys[0], N, seed=9)
orc = o.OracleFilter(model.model_id, model.params, N, 9).initialize(ys[0])
for t, (method, kw) in enumerate((("multinomial", {}), ("stratified", {"sort_particles": False}), ("residual", {}))):
    g.pf_resample(st, method, check=False, **kw); orc.resample(method, check=False, **kw)
    a, b = st.parents, orc.parents
    bad = np.nonzero(a != b)[0]
    print(method, "mismatches", bad.size, bad[:10], a[bad[:10]], b[bad[:10]])
    if t < 2:
        g.pf_update(st, (t + 2,), (None,), ys[t + 1]); orc.update(ys[t + 1])
