"""the headline loop (LG-SSM, N = 1e6, multinomial every step) timed in windows of 100 steps: is the step time flat over a long run?
   python3 tools/headline_drift.py [check: warn|false] [windows]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402

check = {"warn": "warn", "false": False}[sys.argv[1] if len(sys.argv) > 1 else "warn"]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 12
model = g.models.lgssm2(); ys = g.models.simulate(model, 100 * W + 30)
st = g.pf_initialize(model, (1,), ys[0], 1_000_000, seed=1)
t = 1
for _ in range(20):
    g.pf_resample(st, "multinomial", check=check); g.pf_update(st, (t + 1,), (None,), ys[t]); t += 1
st.synchronize()
gc.collect(); gc.freeze(); gc.disable()
out = []
for w in range(W):
    t0 = time.perf_counter()
    for _ in range(100):
        g.pf_resample(st, "multinomial", check=check); g.pf_update(st, (t + 1,), (None,), ys[t]); t += 1
    st.synchronize()
    out.append(round((time.perf_counter() - t0) / 100 * 1e6, 2))
print("check =", check, "us/step per window of 100:", out)
