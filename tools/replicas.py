"""Concurrent independent filters: what the idle GPU is worth ("R = 32 independent seeds"; the block-wise
pattern of the reference's test/resample.jl:130-162 at full size).  Every kernel of a step at N <= 2e6 is a latency chain (one tile per workgroup,
0.1 - 0.26 of the HBM roofline), so R filters on R streams can overlap.  R in {1, 2, 4, 8} filters of BASELINE config 2 (LG-SSM, N = 1e6,
multinomial every step) and config 5 (SV, N = 2e6, multinomial + move-reweight), seeds 1..R, one handle each on its own library-owned stream,
stepped round-robin from ONE host thread; prints aggregate particle-steps/s, the speed-up over R = 1, and the per-filter log-ML spread (config 5's
R-seed estimator variance).   python3 tools/replicas.py [config2|config5|config2s|config3] [--steps K] [--replicas R]"""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import gpf_amd as g  # noqa: E402

CONFIGS = {"config2": ("lgssm2", 1_000_000, "multinomial", None), "config5": ("sv1", 2_000_000, "multinomial", "reweight"),
           "config2s": ("lgssm2", 1_000_000, "multinomial_sorted", None), "config3": ("lgssm2", 1_000_000, "stratified", None)}


def run(name, R, steps, warm=10):
    model_name, N, method, rejuv = CONFIGS[name]
    model = g.models.by_name(model_name)
    ys = g.models.simulate(model, steps + warm + 2)
    sts = [g.pf_initialize(model, (1,), ys[0], N, seed=1 + r, keep_prev=rejuv is not None) for r in range(R)]
    kw = {"sort_particles": False} if method == "stratified" else {}

    def round_(t):
        for st in sts:
            g.pf_resample(st, method, check=False, **kw)
            if rejuv:
                g.pf_rejuvenate(st, None, (), 1, method=rejuv)
            g.pf_update(st, (t + 1,), (None,), ys[t])
    t = 1
    for _ in range(warm):
        round_(t); t += 1
    for st in sts:
        st.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(steps):
        round_(t); t += 1
    for st in sts:
        st.synchronize()
    el = time.perf_counter() - t0
    gc.enable()
    lml = np.array([g.get_lml_est(st) for st in sts])
    out = dict(config=name, replicas=R, N=N, steps=steps, us_per_round=round(el / steps * 1e6, 2), us_per_filter_step=round(el / steps / R * 1e6, 2),
               particle_steps_per_s=round(R * N * steps / el, 1), log_ml_mean=float(lml.mean()), log_ml_std_over_seeds=float(lml.std(ddof=1)) if R > 1 else None)
    for st in sts:
        st.close()
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    steps = 200
    if "--steps" in argv:
        i = argv.index("--steps"); steps = int(argv[i + 1]); del argv[i:i + 2]
    counts = (1, 2, 4, 8)
    if "--replicas" in argv:                   # one count only (under rocprofv3: the kernels' durations with R filters in flight)
        i = argv.index("--replicas"); counts = (int(argv[i + 1]),); del argv[i:i + 2]
    for name in (argv or ["config2", "config5"]):
        base = None
        for R in counts:
            o = run(name, R, steps)
            base = base or o["particle_steps_per_s"]
            o["speedup_over_one_filter"] = round(o["particle_steps_per_s"] / base, 3)
            print(json.dumps(o), flush=True)
