"""which code path does not give its device memory back (development aid)"""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import gpf_amd as g
from gpf_amd import sharded
model = g.models.bearings4(); ys = g.models.simulate(model, 4)

def a():
    st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3, keep_prev=True, history=8)
    g.pf_update(st, (2,), (None,), ys[1]); g.mean(st, (1, 0)); st.close()
def b():
    st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3, keep_prev=True)
    g.pf_resample(st, "stratified", check=False, sort_particles=True); g.pf_rejuvenate(st, g.mh, (), 1); st.close()
def c():
    st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3)
    g.pf_resample(st, "residual", check=False); g.pf_update(st, (2,), (None,), ys[1]); st.close()
def d():
    st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3)
    g.pf_update(st[1000:5000], (3,), (None,), ys[2]); st.close()
def e():
    st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3)
    g.pf_resize(st, 50_000, "optimal", check=False); g.pf_replicate(st, 3); g.sample_unweighted_traces(st, 1000); st.close()
def f():
    sh = sharded.pf_initialize(model, (1,), ys[0], 200_000, seed=3)
    sharded.pf_resample(sh, "multinomial", check=False); sharded.pf_update(sh, (2,), (None,), ys[1]); sharded.get_lml_est(sh); sh.local.close()
def z():
    st = g.pf_initialize(model, (1,), ys[0], 200_000, seed=3); st.close()

def f0():
    sh = sharded.pf_initialize(model, (1,), ys[0], 200_000, seed=3); sh.local.close()
def f1():
    s_ = torch.cuda.Stream(torch.device("cuda", 0)); del s_
def f2():
    sh = sharded.pf_initialize(model, (1,), ys[0], 200_000, seed=3)
    b = sh.backend; mf = b.weight_max(); b.weight_scan(mf.unsqueeze(0).contiguous(), False); sh.local.close()
def f3():
    x = torch.empty((465_536, 5), dtype=torch.float64, device="cuda"); del x
for name, fn in (("sh-init", f0), ("stream", f1), ("sh-scan", f2), ("tensor", f3), ("plain", z), ("history", a), ("sort+mh", b), ("residual", c), ("view", d), ("resize", e), ("sharded", f)):
    for _ in range(3): fn()
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache(); f0, _ = torch.cuda.mem_get_info()
    for _ in range(20): fn()
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache(); f1, _ = torch.cuda.mem_get_info()
    print(f"{name:10s} {(f0 - f1) / 20 / 2**20:8.2f} MiB per lifetime")
