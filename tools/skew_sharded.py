"""one-rank sharded resample (library engine) on extreme weight vectors: wall time per resample + update, N = 1e6 -- a probe for
data-dependent slow paths in the push kernels"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import gpf_amd as g
from gpf_amd import sharded
N = 1_000_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
i = np.arange(N, dtype=np.float64); rng = np.random.default_rng(1)
cases = {"spread": -0.5 * rng.standard_normal(N) ** 2, "all equal": np.zeros(N), "one particle": np.where(i == 777_777, 0.0, -800.0),
         "1 % heavy": np.where(i % 100 == 0, 0.0, -60.0), "first half -inf": np.where(i < N / 2, -np.inf, -0.5 * rng.standard_normal(N) ** 2),
         "ascending ramp": i * 1e-5}
st = sharded.pf_initialize(model, (1,), ys[0], N, seed=1)
print(f"{'weights':18s}" + "".join(f"{m:>14s}" for m in ("multinomial", "stratified", "residual")))
for name, lw in cases.items():
    row = []
    for method in ("multinomial", "stratified", "residual"):
        best = 1e9
        for _ in range(3):
            st.local.log_weights = lw
            st.synchronize(); t0 = time.perf_counter()
            sharded.pf_resample(st, method, check=False); sharded.pf_update(st, (2,), (None,), ys[1])
            st.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e6)
        row.append(best)
    print(f"{name:18s}" + "".join(f"{x:14.1f}" for x in row))
