"""The C ABI from a compiled host (examples/lgssm_filter.c): it builds against include/gpf.h with plain gcc and links
libgpf_hip.so (CPU box: build + link only); on the GPU its output equals the Python host's bit for bit."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "lgssm_filter.c")
LIBDIR = os.path.join(ROOT, "genparticlefilters.jl_amd")


def build(tmp_path):
    exe = os.path.join(tmp_path, "lgssm_filter")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-std=c11", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                           os.path.join(LIBDIR, "libgpf_hip.so"), "-Wl,-rpath," + LIBDIR, "-Wl,--allow-shlib-undefined"])
    return exe


def test_c_example_builds_and_links(g, tmp_path):
    """plain C11 against the header, no C++/HIP/torch types needed (no GPU: nothing is run)"""
    exe = build(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
@pytest.mark.parametrize("name,method,ess_fraction,rejuv", [("lgssm2", 0, 2.0, 0), ("sv1", 2, 0.5, 0), ("object_motion", 1, 0.5, 0),
                                                            ("bearings4", 1, 0.5, 1), ("sv1", 0, 2.0, 2)])
def test_c_example_equals_python_host(g, tmp_path, name, method, ess_fraction, rejuv):
    """rejuv 1 / 2: one MH / move-reweight sweep after every resample (the README loop's `if` body); with a timing request the host
    prints a second line, the first stays the same"""
    exe = build(str(tmp_path))
    model = g.models.by_name(name); T, N, seed = 12, 20_000, 31
    ys = g.models.simulate(model, T)
    inp = os.path.join(tmp_path, "input.txt")
    with open(inp, "w") as f:
        f.write(f"{model.model_id} {model.params.size}\n" + " ".join(repr(float(v)) for v in model.params) + "\n")
        f.write(f"{ys.shape[1]} {T}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in ys) + "\n")
    lines = subprocess.check_output([exe, inp, str(N), str(seed), str(method), str(ess_fraction), str(rejuv), "3"], text=True).splitlines()
    # the loop body as ONE gpf_step_ess call per step (README.md:66-77): the same numbers
    one = subprocess.check_output([exe, inp, str(N), str(seed), str(method), str(ess_fraction), str(rejuv), "3", "1"], text=True).splitlines()
    assert one[0] == lines[0]
    out = lines[0].split()
    assert lines[1].split()[0] == "us_per_step" and float(lines[1].split()[1]) > 0.0
    lml, ess, mean0, var0, n_res = float(out[0]), float(out[1]), float(out[2]), float(out[3]), int(out[4])
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=rejuv != 0)
    mname = ["multinomial", "residual", "stratified"][method]
    k = 0
    for t in range(1, T):
        if g.get_ess(st) < ess_fraction * N:
            g.pf_resample(st, mname, check=False, **({"sort_particles": False} if method == 2 else {})); k += 1
            if rejuv:
                g.pf_rejuvenate(st, None, (), 1, method="move" if rejuv == 1 else "reweight")
        g.pf_update(st, (t + 1,), (None,), ys[t])
    assert (lml, ess, mean0, var0, n_res) == (g.get_lml_est(st), g.get_ess(st), g.mean(st, 0), g.var(st, 0), k)


@pytest.mark.gpu
@pytest.mark.parametrize("one_call", [0, 1])
def test_bench_configs_python_and_compiled_host_lines_agree(g, one_call):
    """tools/bench_configs.py prints config 4 twice -- the loop driven from Python and from the compiled host (examples/lgssm_filter.c) --: the two
    lines describe the SAME filter run (same seed, same observations, warm-up + timed steps), so their log_ml must agree to the last digit (round 5:
    the Python line read its estimate after the kernel-timing replay: -88 582 against 773.43)"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    bc = importlib.util.module_from_spec(spec); sys.modules["bench_configs"] = bc; spec.loader.exec_module(bc)
    bc.NO_CPU = True
    N, steps, warm = 30_000, 40, 5
    py = bc.run("config4 at test size", "bearings4", N, "residual", {"_step_ess": True} if one_call else {}, "move", 0.5, steps=steps, warm=warm)
    c = bc.run_c_host("config4 at test size, compiled host", "bearings4", N, 1, 1, 0.5, steps=steps, warm=warm, one_call=one_call)
    assert py["log_ml"] == c["log_ml"], (py["log_ml"], c["log_ml"])
    assert py["resampled_steps"] <= c["resampled_steps_incl_warmup"] <= py["resampled_steps"] + warm
