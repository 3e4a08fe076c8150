import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g
from oracle import oracle as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 4)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
orc = o.OracleFilter(model.model_id, model.params, N, 1).initialize(ys[0])
for t in range(1, 3):
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
    a, b = st.parents, orc.parents
    bad = np.nonzero(a != b)[0]
    print("step", t, "mismatches", bad.size, "of", N)
    if bad.size:
        d = (a[bad] - b[bad])
        print(" first slots", bad[:10], "gpu", a[bad[:10]], "orc", b[bad[:10]], "diff hist", np.unique(d, return_counts=True))
        print(" orc anc mod 32 of bad:", np.unique((b[bad]-1) % 32, return_counts=True))
    g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
