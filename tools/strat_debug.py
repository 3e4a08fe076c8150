"""Per-workgroup phase times of k_search_strat (a -DGPF_DBG_STRAT build: tools/build_variant.sh dbg -DGPF_DBG_STRAT; run with
GPF_LIB_OVERRIDE=.../libgpf_dbg.so): wall_clock64 (100 MHz) at entry / after the prologue / after the search / at the end."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
mode = sys.argv[1] if len(sys.argv) > 1 else "sorted"          # sorted | unsorted (stratified) | multinomial_sorted
sorted_ = mode == "sorted"
N = 1_000_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 12)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
def resample():
    if mode == "multinomial_sorted":
        g.pf_resample(st, "multinomial_sorted", check=False)
    else:
        g.pf_resample(st, "stratified", check=False, sort_particles=sorted_)
for t in range(1, 10):
    resample()
    g.pf_update(st, (t + 1,), (None,), ys[t])
resample()
st.synchronize()
lib = C.CDLL(os.environ["GPF_LIB_OVERRIDE"])
spb = 2048 if mode == "multinomial_sorted" else 1024            # slots per workgroup (gpf_k_search.hpp: SP_TILE / MJB_STRAT)
nb = (N + spb - 1) // spb
buf = (C.c_ulonglong * (8 * 4096))()
assert lib.gpf_debug_strat(buf, 8 * 4096) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[:nb].astype(np.int64)
t0 = a[:, 0].min()
start, pro, sea, end, ncell = (a[:, 0] - t0) / 100.0, (a[:, 1] - a[:, 0]) / 100.0, (a[:, 2] - a[:, 1]) / 100.0, (a[:, 3] - a[:, 2]) / 100.0, a[:, 4]
print(mode, "blocks", nb, "kernel span us", ((a[:, 3] - t0) / 100.0).max())
print("start us: max %.2f" % start.max(), " prologue: mean %.2f max %.2f" % (pro.mean(), pro.max()), " search: mean %.2f max %.2f" % (sea.mean(), sea.max()),
      " epilogue: mean %.2f max %.2f" % (end.mean(), end.max()))
wide = ncell > 8 * spb
print("wide blocks", int(wide.sum()), "search us wide mean %.2f max %.2f" % (sea[wide].mean() if wide.any() else 0, sea[wide].max() if wide.any() else 0),
      " streamed mean %.2f max %.2f" % (sea[~wide].mean(), sea[~wide].max()))
for lo, hi in ((0, 2048), (2048, 4096), (4096, 8192), (8192, 16385), (16385, 1 << 40)):
    m = (ncell >= lo) & (ncell < hi)
    if m.any():
        print("  cells in [%d, %d): %d blocks, search mean %.2f max %.2f, prologue mean %.2f, end-of-block (us after kernel start) mean %.2f max %.2f"
              % (lo, hi, m.sum(), sea[m].mean(), sea[m].max(), pro[m].mean(), ((a[:, 3] - t0) / 100.0)[m].mean(), ((a[:, 3] - t0) / 100.0)[m].max()))
