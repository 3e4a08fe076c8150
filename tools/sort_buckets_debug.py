"""Per-workgroup phase times of k_sort_buckets from a -DGPF_DBG_SORT build (tools/build_variant.sh dbg -DGPF_DBG_SORT; run with
GPF_LIB_OVERRIDE=.../libgpf_dbg.so): wall_clock64 (100 MHz) stamps of thread 0, the last sorted resample of a 10-step filter."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
model = g.models.lgssm2(); ys = g.models.simulate(model, steps + 3)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
lib = C.CDLL(os.environ["GPF_LIB_OVERRIDE"])
names = ["start", "loads landed, coarse keys", "min / max barrier", "count, scan, ord", "rank inside the bins", "barrier", "LDS scatter + stores"]
for t in range(1, steps + 2):
    g.pf_resample(st, "stratified", check=False, sort_particles=True)
    st.synchronize()
    buf = (C.c_ulonglong * (8 * 256))()
    assert lib.gpf_debug_sort_buckets(buf, 8 * 256) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
    L = a[:, 7]; a = a[:, :7]; t0 = a[:, 0].min()
    print("step %d: span %.2f us; bucket sizes mean %.0f max %d; " % (t, (a[:, 6].max() - t0) / 100.0, L.mean(), L.max()), end="")
    print("start max %.2f | " % ((a[:, 0] - t0) / 100.0).max(), end="")
    print(" | ".join("%s %.2f/%.2f" % (names[k].split(",")[0][:12], ((a[:, k] - a[:, k - 1]) / 100.0).mean(), ((a[:, k] - a[:, k - 1]) / 100.0).max()) for k in range(1, 7)))
    g.pf_update(st, (t + 1,), (None,), ys[t])
