"""resample + stand-alone gather + update loop for one method (profiling target for tools/gpu_pmc_gather.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
method = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
kw = {"sort_particles": False} if method == "stratified" else {}
model = g.models.lgssm2(); ys = g.models.simulate(model, 40)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
for t in range(1, 36):
    g.pf_resample(st, method, check=False, **kw); g.get_ess(st)      # ESS forces the stand-alone gather (k_gather<2>)
    g.pf_update(st, (t,), (None,), ys[t])
st.synchronize()
