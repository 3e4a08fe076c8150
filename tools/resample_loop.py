"""Plain resample + update loop for one method (profiling target: rocprofv3 ... -- python3 tools/resample_loop.py stratified)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "multinomial"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
kw = {"sort_particles": False} if method == "stratified" else {}
if method == "stratified_sorted":
    method, kw = "stratified", {"sort_particles": True}
model = g.models.lgssm2()
ys = g.models.simulate(model, steps + 1)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
import time  # noqa: E402
st.synchronize(); t0 = time.perf_counter()
for t in range(1, steps + 1):
    g.pf_resample(st, method, check=False, **kw)
    g.pf_update(st, (t + 1,), (None,), ys[t])
st.synchronize()
print("us/step", round((time.perf_counter() - t0) / steps * 1e6, 2), "(first steps included)", "log-ML", g.get_lml_est(st))
