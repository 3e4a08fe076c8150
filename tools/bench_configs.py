"""Single-GPU timing of the BASELINE.json configs other than the headline one (per-GPU sizes), with per-kernel
HIP-event times.  Not the driver's bench; numbers go to DESIGN.md / BASELINE table."""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402

CONFIGS = [
    ("config2 lgssm2 multinomial", "lgssm2", 1_000_000, "multinomial", {}, None, None),
    ("config2s lgssm2 multinomial_sorted (opt-in: sorted uniforms)", "lgssm2", 1_000_000, "multinomial_sorted", {}, None, None),
    ("config2l lgssm2 multinomial, lazy search (k_step_search)", "lgssm2", 1_000_000, "multinomial", {"_lazy": True}, None, None),
    ("config3 lgssm2 stratified(unsorted) [1 of 8 shards' worth]", "lgssm2", 1_000_000, "stratified", {"sort_particles": False}, None, None),
    ("lgssm2 stratified(sorted)", "lgssm2", 1_000_000, "stratified", {"sort_particles": True}, None, None),
    ("lgssm2 residual", "lgssm2", 1_000_000, "residual", {}, None, None),
    ("config4 bearings4 ESS<N/2 residual + MH [1 of 4 shards' worth]", "bearings4", 1_000_000, "residual", {}, "move", 0.5),
    ("config5 sv1 multinomial + move-reweight", "sv1", 2_000_000, "multinomial", {}, "reweight", None),
    ("config5s sv1 multinomial_sorted + move-reweight", "sv1", 2_000_000, "multinomial_sorted", {}, "reweight", None),
]


def run(name, model_name, N, method, kw, rejuv, ess_frac, steps=200, warm=10):
    model = g.models.by_name(model_name)
    ys = g.models.simulate(model, steps + warm + 1)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=rejuv is not None)
    kw = dict(kw)
    if kw.pop("_lazy", False):
        st.set_lazy_search(True)
    n_res = 0

    def step(t):
        nonlocal n_res
        if ess_frac is None or g.get_ess(st) < ess_frac * N:
            n_res += 1
            g.pf_resample(st, method, check=False, **kw)
            if rejuv:
                g.pf_rejuvenate(st, None, (), 1, method=rejuv)
        g.pf_update(st, (t + 1,), (None,), ys[t])

    t = 1
    for _ in range(warm):
        step(t); t += 1
    st.synchronize(); n_res = 0
    gc.collect(); gc.disable()          # as bench.py: keep generation-2 collections (tens of ms) out of the timed loop
    t0 = time.perf_counter()
    for _ in range(steps):
        step(t); t += 1
    st.synchronize()
    el = time.perf_counter() - t0
    gc.enable()
    kids = list(g._lib.KERNEL_NAMES)
    for k in kids:
        st.kernel_timing(k, True)
    for i in range(50):
        step(1 + i)
    per = {}
    for k in kids:
        ms, cnt = st.kernel_time(k); st.kernel_timing(k, False)
        if cnt:
            per[g._lib.KERNEL_NAMES[k]] = round(ms / cnt * 1e3, 2)
    out = dict(config=name, N=N, steps=steps, us_per_step=round(el / steps * 1e6, 2), particle_steps_per_s=round(N * steps / el, 1),
               resampled_steps=n_res, kernels_us=per, log_ml=g.get_lml_est(st))
    print(json.dumps(out), flush=True)
    st.close()


if __name__ == "__main__":
    want = sys.argv[1:]                    # e.g. `config4 config5`: only the configs whose name starts with one of these
    for c in CONFIGS:
        if not want or any(c[0].startswith(w) for w in want):
            run(*c)
