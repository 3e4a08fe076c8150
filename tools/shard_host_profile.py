"""Host-side cost of one sharded step (world = 1, no wire): wall time of every Python-level phase, GPU work asynchronous."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import gpf_amd as g
from gpf_amd import sharded

model = g.models.lgssm2(); ys = g.models.simulate(model, 400)
st = sharded.pf_initialize(model, (1,), ys[0], 1_000_000, seed=1)
b = st.backend
acc = collections.defaultdict(float)
def timed(name, f, *a):
    t0 = time.perf_counter(); r = f(*a); acc[name] += time.perf_counter() - t0; return r
for name in ("weight_max", "weight_scan", "push_count", "push", "commit", "update"):
    orig = getattr(b, name)
    setattr(b, name, (lambda n, o: (lambda *a: timed(n, o, *a)))(name, orig))
T = 300
for t in range(1, 20):
    sharded.pf_resample(st, "multinomial", check=False); sharded.pf_update(st, (t,), (None,), ys[t])
st.synchronize(); acc.clear()
t0 = time.perf_counter()
for t in range(20, 20 + T):
    t1 = time.perf_counter(); sharded.pf_resample(st, "multinomial", check=False); acc["pf_resample total"] += time.perf_counter() - t1
    t1 = time.perf_counter(); sharded.pf_update(st, (t,), (None,), ys[t]); acc["pf_update total"] += time.perf_counter() - t1
st.synchronize()
wall = time.perf_counter() - t0
print(f"wall per step {wall / T * 1e6:.1f} us")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {v / T * 1e6:8.1f} us/step")
