import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
model = g.models.lgssm2(); ys = g.models.simulate(model, 4)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
L = g._lib.load()
def level(which, dtype, cap):
    buf = np.zeros(cap, dtype); nb = C.c_int64(buf.nbytes)
    assert L.gpf_debug_levels(st._h, which, buf.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0, L.gpf_last_error(st._h)
    return buf[: nb.value // buf.itemsize]
ntiles = (N + 2047) // 2048
for t in range(1, 3):
    g.pf_resample(st, "multinomial", check=False)
    cdf = level(0, np.uint64, ntiles * 2048); k32 = level(3, np.uint32, ntiles * 64)
    off = level(4, np.uint16, ntiles * 2048); coarse = level(5, np.uint16, ntiles * 512)
    G = 32
    ends = cdf[G - 1::G]
    print("step", t, "keys ok:", np.array_equal(k32, (ends >> np.uint64(30)).astype(np.uint32)))
    klo = np.concatenate([[0], k32[:-1]]).astype(np.uint64); khi = k32.astype(np.uint64)
    w = khi - klo + 1
    cl = np.array([0 if x <= 1 else int(x - 1).bit_length() for x in w])
    sh = (14 + cl).astype(np.uint64)
    exp_off = ((cdf.reshape(-1, G) - (klo << np.uint64(30))[:, None]) >> sh[:, None]).reshape(-1)
    bad = np.nonzero(exp_off != off)[0]
    print(" off16 mismatches", bad.size, "max expected", exp_off.max(), "first bad", bad[:8], exp_off[bad[:8]], off[bad[:8]])
    print(" coarse ok:", np.array_equal(coarse, off[3::4]))
    g.pf_update(st, (t + 1,), (None,), ys[t])

# ---- emulate the narrow levels of k_search_multi in numpy on the oracle's targets, compare with the GPU's parents
from oracle import oracle as o
st = g.pf_initialize(model, (1,), ys[0], N, seed=1)
orc = o.OracleFilter(model.model_id, model.params, N, 1).initialize(ys[0])
for t in range(1, 3):
    sp = o.WeightSummary(orc.lw, N)
    T = o.targets_multinomial(orc.seed, orc.epoch, 0, N, sp.S)
    g.pf_resample(st, "multinomial", check=False); orc.resample("multinomial", check=False)
    cdf = level(0, np.uint64, ntiles * 2048); k32 = level(3, np.uint32, ntiles * 64)
    off = level(4, np.uint16, ntiles * 2048).astype(np.int64); coarse = level(5, np.uint16, ntiles * 512).astype(np.int64)
    print("step", t, "cdf == oracle cdf:", np.array_equal(cdf[:N], sp.cdf))
    tk = (T >> np.uint64(30)).astype(np.uint32)
    gi = np.searchsorted(k32, tk, side="left")            # number of keys < t
    amb = (gi < k32.size) & (k32[np.minimum(gi, k32.size - 1)] == tk)
    print(" key-level ambiguous slots:", int(amb.sum()))
    gexact = np.searchsorted(cdf[31::32], T, side="right")
    gi = gexact
    klo = np.where(gi > 0, k32[np.maximum(gi - 1, 0)], 0).astype(np.uint64); khi = k32[gi].astype(np.uint64)
    w = khi - klo + 1
    cl = np.array([0 if x <= 1 else int(x - 1).bit_length() for x in w])
    sh = (14 + cl).astype(np.uint64)
    q = ((T - (klo << np.uint64(30))) >> sh).astype(np.int64)
    rows = coarse.reshape(-1, 8)[gi]
    run = np.minimum((rows < q[:, None]).sum(1), 7)
    fine = off.reshape(-1, 32)[gi].reshape(N, 8, 4)[np.arange(N), run]
    lt = (fine < q[:, None]).sum(1); tie = (fine == q[:, None]).any(1)
    idx = gi * 32 + run * 4 + lt
    truth = orc.parents - 1
    print(" emulated == oracle where no tie:", np.array_equal(idx[~tie], truth[~tie]), " ties:", int(tie.sum()), " q max", q.max())
    gp = st.parents - 1
    bad = np.nonzero(gp != truth)[0]
    print(" gpu mismatches", bad.size, " of which tie slots:", int(tie[bad].sum()), " emulated==gpu on bad:", int((idx[bad] == gp[bad]).sum()))
    if bad.size:
        b = bad[:6]
        print("  slot", b, "T>>30", tk[b], "g", gi[b], "q", q[b], "run", run[b], "lt", lt[b], "gpu", gp[b], "truth", truth[b])
    g.pf_update(st, (t + 1,), (None,), ys[t]); orc.update(ys[t])
