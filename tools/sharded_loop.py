"""One-rank sharded resample + update loop through the library engine (profiling target: rocprofv3 ... -- python3 tools/sharded_loop.py multinomial).
No process group: gpf_comm_create with world 1 (the kernels of the sharded path without RCCL traffic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402
from gpf_amd import sharded  # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "multinomial"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
model = g.models.lgssm2()
ys = g.models.simulate(model, steps + 1)
st = sharded.pf_initialize(model, (1,), ys[0], N, seed=1)
import time
for rep in range(2):
    st.backend.synchronize(); t0 = time.perf_counter()
    for t in range(1, steps + 1):
        if method == "stratified_sorted":                   # the reference's default sort_particles = true: gpf_shard_resample_sorted (every rank sorts all n_global weights)
            sharded.pf_resample(st, "stratified", sort_particles=True, check=False)
        else:
            sharded.pf_resample(st, method, check=False)
        sharded.pf_update(st, (t + 1,), (None,), ys[t])
    st.backend.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e6
print("us/step", round(dt, 2), "log-ML", sharded.get_lml_est(st))
