import sys, time, gc
sys.path.insert(0, "/root/repo")
import gpf_amd as g
model = g.models.lgssm2(); ys = g.models.simulate(model, 700)
for chk in (False, "warn", True):
    st = g.pf_initialize(model, (1,), ys[0], 1_000_000, seed=1)
    for t in range(1, 50):
        g.pf_resample(st, "multinomial", check=chk); g.pf_update(st, (t,), (None,), ys[t])
    st.synchronize(); gc.collect(); gc.disable(); t0 = time.perf_counter()
    for t in range(50, 650):
        g.pf_resample(st, "multinomial", check=chk); g.pf_update(st, (t,), (None,), ys[t])
    st.synchronize(); el = time.perf_counter() - t0; gc.enable()
    print(chk, round(el / 600 * 1e6, 1), "us/step")
    st.close()
