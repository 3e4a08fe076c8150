"""Single-GPU timing of the BASELINE.json configs (per-GPU sizes) with per-kernel HIP-event times, and -- SURVEY 8(d) / BASELINE.md 3 --
the CPU path timed beside every config: the C oracle (a port of the reference algorithm; the Julia reference cannot run here) on a bounded
sample of the same loop, single-threaded (the reference is) and on up to 16 OpenMP threads.  Not the driver's bench; numbers go to
DESIGN.md 7.  `--no-cpu` skips the CPU legs; `--cpu-seconds S` bounds each single-thread leg (default 6 s)."""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402

CPU_SECONDS = 6.0          # bound of every single-thread CPU leg (the sample = as many steps of the same loop as fit, at least 3)

CONFIGS = [
    ("config1 object_motion N=100, ESS<N/2 residual + MH (the README loop, the reference's own CPU-sized case; timed over 200 steps, not T=10)", "object_motion", 100, "residual", {}, "move", 0.5),
    ("config2 lgssm2 multinomial", "lgssm2", 1_000_000, "multinomial", {}, None, None),
    ("config2s lgssm2 multinomial_sorted (opt-in: sorted uniforms)", "lgssm2", 1_000_000, "multinomial_sorted", {}, None, None),
    ("config2l lgssm2 multinomial, lazy search (k_step_search)", "lgssm2", 1_000_000, "multinomial", {"_lazy": True}, None, None),
    ("config3 lgssm2 stratified(unsorted) [1 of 8 shards' worth]", "lgssm2", 1_000_000, "stratified", {"sort_particles": False}, None, None),
    ("lgssm2 stratified(sorted)", "lgssm2", 1_000_000, "stratified", {"sort_particles": True}, None, None),
    ("bearings4 stratified(sorted) (weights beyond the coarse key's range every step)", "bearings4", 1_000_000, "stratified", {"sort_particles": True}, None, None),
    ("big lgssm2 stratified(sorted), N = 2 x 10^6 (the bucket sort's wide form)", "lgssm2", 2_000_000, "stratified", {"sort_particles": True}, None, None),
    ("big bearings4 stratified(sorted), N = 2 x 10^6", "bearings4", 2_000_000, "stratified", {"sort_particles": True}, None, None),
    ("big lgssm2 stratified(sorted), N = 2.4 x 10^6 (beyond the bucket sort: three coarse passes + finish)", "lgssm2", 2_400_000, "stratified", {"sort_particles": True}, None, None),
    ("lgssm2 residual", "lgssm2", 1_000_000, "residual", {}, None, None),
    ("config4 bearings4 ESS<N/2 residual + MH [1 of 4 shards' worth]", "bearings4", 1_000_000, "residual", {}, "move", 0.5),
    ("config4g the same loop, one pf_step_ess call per step (gpf_step_ess: verdict on the device, speculative propagate)", "bearings4", 1_000_000, "residual", {"_step_ess": True}, "move", 0.5),
    ("config5 sv1 multinomial + move-reweight", "sv1", 2_000_000, "multinomial", {}, "reweight", None),
    ("config5s sv1 multinomial_sorted + move-reweight", "sv1", 2_000_000, "multinomial_sorted", {}, "reweight", None),
]


def cpu_baseline(model, ys, N, method, kw, rejuv, ess_frac, seconds):
    """the oracle on the same loop: single thread, then OpenMP; returns the two cpu_baseline objects (bench.py's format)"""
    from oracle import oracle as o          # the checker as the timed CPU comparator: tools/ only, never the product
    o.lib()
    okw = {k: v for k, v in kw.items() if not k.startswith("_")}
    om = {"multinomial_sorted": "multinomial_sorted"}.get(method, method)

    def loop(threads, budget, max_steps):
        used = o.set_threads(threads)
        orc = o.OracleFilter(model.model_id, model.params, N, 1, keep_prev=rejuv is not None).initialize(ys[0])
        k, c0 = 0, time.perf_counter()
        while k < max_steps and (k < 3 or time.perf_counter() - c0 < budget):
            if ess_frac is None or orc.effective_sample_size() < ess_frac * N:
                orc.resample(om, check=False, **okw)
                if rejuv:
                    orc.rejuvenate(rejuv, 1)
            orc.update(ys[1 + k % (len(ys) - 1)])
            k += 1
        ce = time.perf_counter() - c0
        o.set_threads(1)
        return used, k, ce
    u1, k1, c1 = loop(1, seconds, 10_000)
    if N < 100_000:                      # (a fork-join per primitive at N = 100 measures OpenMP, not the filter)
        un, kn, cn = u1, k1, c1
    else:
        un, kn, cn = loop(min(os.cpu_count() or 1, 16), seconds / 2, k1)
    one = {"value": round(N * k1 / c1, 1), "unit": "particle-steps/sec", "cores": 1, "kind": "port",
           "sample": f"same loop, N={N}, first {k1} steps, single-thread C oracle ({c1:.1f} s); reference (Julia) not runnable on this box"}
    many = {"value": round(N * kn / cn, 1), "unit": "particle-steps/sec", "cores": un, "kind": "port",
            "sample": f"same loop, first {kn} steps, OpenMP over particles on {un} threads ({cn:.1f} s)"}
    return one, many


NO_CPU = False


def run(name, model_name, N, method, kw, rejuv, ess_frac, steps=200, warm=10):
    model = g.models.by_name(model_name)
    ys = g.models.simulate(model, steps + warm + 1)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=rejuv is not None)
    kw = dict(kw)
    lazy = kw.pop("_lazy", False)
    if lazy:
        st.set_lazy_search(True)
    n_res = 0
    gated = kw.pop("_step_ess", False)

    def step(t):
        nonlocal n_res
        if gated:
            n_res += g.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=ess_frac, method=method, rejuvenate=rejuv, check=False, **kw)
            return
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
    n_res_timed = n_res                    # (the per-kernel timing steps below replay early observations and resample almost every time)
    # the estimate of the TIMED run (steps 1 .. warm + steps, what the compiled host's line reports too) -- read before the replay below moves it
    log_ml = g.get_lml_est(st)
    kids = list(g._lib.KERNEL_NAMES)
    for k in kids:
        st.kernel_timing(k, True)
    for i in range(min(50, steps)):
        step(1 + i)
    per = {}
    for k in kids:
        ms, cnt = st.kernel_time(k); st.kernel_timing(k, False)
        if cnt:
            per[g._lib.KERNEL_NAMES[k]] = round(ms / cnt * 1e3, 2)
    out = dict(config=name, N=N, steps=steps, us_per_step=round(el / steps * 1e6, 2), particle_steps_per_s=round(N * steps / el, 1),
               resampled_steps=n_res_timed, kernels_us=per, log_ml=log_ml)
    st.close()
    if not NO_CPU and not lazy and not gated:
        out["cpu_baseline"], out["cpu_baseline_multithread"] = cpu_baseline(model, ys, N, method, kw, rejuv, ess_frac, CPU_SECONDS)
        out["gpu_over_cpu_1core"] = round(out["particle_steps_per_s"] / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out), flush=True)
    return out


def run_c_host(name, model_name, N, method_id, rejuvenate, ess_frac, steps=200, warm=10, one_call=0):
    """the same ESS-triggered loop from the compiled host (examples/lgssm_filter.c over the C ABI): what the loop costs without the
    Python wrappers on the critical path between the ESS read and the next launch -- the drop-in's host is Julia (ccall), not Python"""
    import subprocess
    import tempfile
    model = g.models.by_name(model_name)
    ys = g.models.simulate(model, steps + warm + 1)
    tmp = tempfile.mkdtemp()
    exe, inp = os.path.join(tmp, "host"), os.path.join(tmp, "input.txt")
    libdir = os.path.join(ROOT, "genparticlefilters.jl_amd")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "lgssm_filter.c"), "-o", exe,
                           os.path.join(libdir, "libgpf_hip.so"), "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])
    with open(inp, "w") as f:
        f.write(f"{model.model_id} {model.params.size}\n" + " ".join(repr(float(v)) for v in model.params) + "\n")
        f.write(f"{ys.shape[1]} {ys.shape[0]}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in ys) + "\n")
    out = subprocess.check_output([exe, inp, str(N), "1", str(method_id), str(ess_frac), str(rejuvenate), str(warm + 1), str(one_call)], text=True).splitlines()
    first = out[0].split(); us = float(out[1].split()[1])
    res = dict(config=name, N=N, steps=steps, us_per_step=round(us, 2), particle_steps_per_s=round(N / us * 1e6, 1),
               resampled_steps_incl_warmup=int(first[4]), warmup=warm, log_ml=float(first[0]))
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--no-cpu" in argv:
        NO_CPU = True; argv.remove("--no-cpu")
    if "--cpu-seconds" in argv:
        i = argv.index("--cpu-seconds"); CPU_SECONDS = float(argv[i + 1]); del argv[i:i + 2]
    want = argv                            # e.g. `config4 config5`: only the configs whose name starts with one of these
    for c in CONFIGS:
        if not want or any(c[0].startswith(w) for w in want):
            run(*c)
            if c[0].startswith("config4 "):
                run_c_host("config4, the same loop from a compiled host (examples/lgssm_filter.c)", "bearings4", c[2], 1, 1, 0.5)
            if c[0].startswith("config4g"):
                run_c_host("config4g from the compiled host: one gpf_step_ess call per step", "bearings4", c[2], 1, 1, 0.5, one_call=1)
