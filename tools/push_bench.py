"""Kernel timing of the sharded push exchange at a SIMULATED shard count on ONE GPU: this process plays shard `me` of G,
the other shards' summaries are copies of its own (balanced weights).  No collective runs; what is measured is the
per-shard kernel work of one resample (count pass, bases, push) as a function of G.  Run under rocprofv3 for per-kernel times."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import gpf_amd as g
from gpf_amd import sharded

n = int(os.environ.get("PUSH_N", 1_000_000))
iters = int(os.environ.get("PUSH_ITERS", 30))
model = g.models.lgssm2(); ys = g.models.simulate(model, 4)
for method in os.environ.get("PUSH_METHODS", "multinomial,stratified,residual").split(","):
    mid = sharded.RESAMPLE_METHODS[method]
    for G in [int(x) for x in os.environ.get("PUSH_G", "1,2,4,8").split(",")]:
        me = G // 2
        bounds = [r * n for r in range(G + 1)]
        b = sharded.HipShardBackend(model, G * n, me * n, n, 1, False, 0)
        b.initialize(ys[0]); b.update(ys[1])
        mf = b.weight_max(); mf_all = mf.unsqueeze(0).repeat(G, 1).contiguous()
        tot = b.weight_scan(mf_all, False); tot_all = tot.unsqueeze(0).repeat(G, 1).contiguous()
        cr_all = None
        if mid == 1:
            cr = b.residual_scan(tot_all); cr_all = cr.unsqueeze(0).repeat(G, 1).contiguous()
        st = b.state
        for phase in ("warm", "timed"):
            st.kernel_timing(g._lib.K_SEARCH, True); st.kernel_timing(g._lib.K_GATHER, True)
            b.synchronize(); t0 = time.perf_counter()
            for _ in range(iters if phase == "timed" else 3):
                tot = b.weight_scan(mf_all, False)                     # clears the exchange counters
                b.push_count(mid, tot_all, cr_all, me, bounds)
                packed = b.push(mid, tot_all, cr_all, me, bounds, 2 * n + 65536)  # noqa
                c = b.counts(G)
            b.synchronize(); wall = (time.perf_counter() - t0) / iters * 1e6
        cms, cc = st.kernel_time(g._lib.K_SEARCH); pms, pc = st.kernel_time(g._lib.K_GATHER)
        print(json.dumps(dict(method=method, G=G, n_local=n, sent=c[:G], recv=c[G:], count_us=round(cms / cc * 1e3, 2),
                              push_us=round(pms / pc * 1e3, 2), wall_us_per_resample=round(wall, 1))), flush=True)
        st.close()
