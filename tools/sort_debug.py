"""Per-workgroup phase times of k_sort_pass (second pass) from a -DGPF_DBG_SORT build: wall_clock64 (100 MHz) stamps of thread 0."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
N = 1_000_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 12)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
for t in range(1, 10):
    g.pf_resample(st, "stratified", check=False, sort_particles=True)
    g.pf_update(st, (t + 1,), (None,), ys[t])
g.pf_resample(st, "stratified", check=False, sort_particles=True)
st.synchronize()
lib = C.CDLL(os.environ["GPF_LIB_OVERRIDE"])
nb = (N + 4095) // 4096
buf = (C.c_ulonglong * (8 * 4096))()
assert lib.gpf_debug_sort(buf, 8 * 4096) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[:nb, :7].astype(np.int64)
t0 = a[:, 0].min()
names = ["start", "ticket + histogram scan", "tile loads issued", "rank (ballots, LDS counters)", "publish + reorder in LDS", "look-back", "scatter"]
print("k_sort_pass, second pass, %d workgroups; kernel span %.2f us" % (nb, (a[:, 6].max() - t0) / 100.0))
print("  %-30s mean %6.2f max %6.2f us after kernel start" % (names[0], ((a[:, 0] - t0) / 100.0).mean(), ((a[:, 0] - t0) / 100.0).max()))
for k in range(1, 7):
    d = (a[:, k] - a[:, k - 1]) / 100.0
    print("  %-30s mean %6.2f max %6.2f us" % (names[k], d.mean(), d.max()))
