"""ESS-triggered loop (BASELINE config 4's control flow on the LG-SSM): get_ess every step, resample when ESS < N/2"""
import sys, time, gc, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
model = g.models.lgssm2(); ys = g.models.simulate(model, 700); N = 1_000_000
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
def step(t):
    if g.get_ess(st) < 0.5 * N:
        g.pf_resample(st, "multinomial")
    g.pf_update(st, (t,), (None,), ys[t])
for t in range(1, 50): step(t)
st.synchronize(); gc.collect(); gc.disable(); t0 = time.perf_counter()
for t in range(50, 650): step(t)
st.synchronize(); el = time.perf_counter() - t0
print("ESS-triggered loop", round(el / 600 * 1e6, 1), "us/step")
t0 = time.perf_counter()
for _ in range(300): g.get_ess(st)
print("get_ess alone (cached summary)", round((time.perf_counter() - t0) / 300 * 1e6, 1), "us")
