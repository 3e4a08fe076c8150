"""wall time per resample + update (synchronised) on extreme weight vectors, every resampler incl. sort_particles=true: a probe for
data-dependent slow paths anywhere in the step.  N = 1e6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import gpf_amd as g
N = 1_000_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
i = np.arange(N, dtype=np.float64)
rng = np.random.default_rng(1)
cases = {
    "spread (gaussian)": -0.5 * rng.standard_normal(N) ** 2,
    "all equal": np.zeros(N),
    "one particle": np.where(i == 777_777, 0.0, -800.0),
    "one particle, rest -inf": np.where(i == 5, 0.0, -np.inf),
    "1 % heavy": np.where(i % 100 == 0, 0.0, -60.0),
    "ascending ramp": i * 1e-5,
    "descending ramp": -i * 1e-5,
    "two values alternating": np.where(i % 2 == 0, 0.0, -1e-9),
    "first half -inf": np.where(i < N / 2, -np.inf, -0.5 * rng.standard_normal(N) ** 2),
    "denormal-scale spread": -700.0 - 40.0 * rng.random(N),
}
variants = [("multinomial", {}), ("residual", {}), ("stratified", {"sort_particles": False}), ("stratified", {"sort_particles": True}),
            ("multinomial", {"priority_fn": g.Tempering(0.5)})]
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
print(f"{'weights':28s}" + "".join(f"{(m + ('+sort' if kw.get('sort_particles') else '') + ('+prio' if 'priority_fn' in kw else ''))[:16]:>18s}" for m, kw in variants))
for name, lw in cases.items():
    row = []
    for method, kw in variants:
        best = 1e9
        for rep in range(3):
            st.log_weights = lw
            st.synchronize(); t0 = time.perf_counter()
            g.pf_resample(st, method, check=False, **kw)
            g.pf_update(st, (2,), (None,), ys[1])
            st.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e6)
        row.append(best)
    print(f"{name:28s}" + "".join(f"{x:18.1f}" for x in row))
