"""bench.py's command line, on the CPU: main() runs against a stand-in for `gpf_amd` (same host interface, the
oracle underneath, fake kernel timers) with the driver's own flags and with degenerate ones, and must print exactly
ONE JSON line carrying `roofline` and `cpu_baseline`.  Round 1's bench indexed observations past the end whenever
--steps + --warmup < 30 (the gather and cpu_baseline legs run a fixed number of steps); this keeps that from coming back.
The real library path of the same command is covered by tests/test_gpu_bench_cli.py on the GPU box."""
import importlib
import io
import json
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import pytest


class _FakeState:
    """What bench.py touches on a DeviceParticleFilterState."""

    def __init__(self, o, model, obs, n, seed):
        self.orc = o.OracleFilter(model.model_id, model.params, n, seed).initialize(obs)
        self.row_width = model.row_width(False)
        self._timing = {}
        self._count = {}

    def synchronize(self):
        pass

    def kernel_timing(self, kid, on):
        self._timing[kid] = bool(on)
        if on:
            self._count[kid] = 0

    def kernel_time(self, kid):
        c = self._count.get(kid, 0)
        return 0.02 * c, c                         # 20 us per launch

    def _launched(self, *kids):
        for k in kids:
            if self._timing.get(k):
                self._count[k] = self._count.get(k, 0) + 1


def _install_stub(monkeypatch, o):
    import gpf_amd as real
    stub = types.ModuleType("gpf_amd")
    stub.models, stub._lib = real.models, real._lib
    L = real._lib

    def pf_initialize(model, args, obs, n, seed=1, device=0):
        return _FakeState(o, model, obs, n, seed)

    def pf_resample(state, method="multinomial", check="warn", **kw):
        state.orc.resample(method, check=False, **kw)
        state._launched(L.K_SCAN, L.K_SEARCH)
        state.pending = True

    def pf_update(state, args, argdiffs, obs):
        state.orc.update(np.asarray(obs, float))
        state._launched(L.K_STEP)
        state.pending = False

    def get_ess(state):
        if getattr(state, "pending", False):
            state._launched(L.K_GATHER)
            state.pending = False
        return state.orc.effective_sample_size()

    def get_lml_est(state):
        return state.orc.log_ml_estimate()

    for f in (pf_initialize, pf_resample, pf_update, get_ess, get_lml_est):
        setattr(stub, f.__name__, f)
    monkeypatch.setitem(sys.modules, "gpf_amd", stub)
    import torch
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_a, **_k: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *_a, **_k: None)


@pytest.mark.parametrize("argv", [
    ["--gpus", "1", "--steps", "20", "--warmup", "5"],          # the driver's command line (BENCH_r01 crashed on it)
    ["--steps", "1", "--warmup", "0"],
    ["--steps", "3", "--warmup", "40"],
    ["--steps", "100", "--warmup", "2"],
])
def test_bench_main_prints_one_json_line(argv, o, built, monkeypatch):
    _install_stub(monkeypatch, o)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPF_BENCH_FORCE_SHARDED", "GPF_BENCH_ONE_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--particles-per-gpu", "512"] + argv)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    steps = int(argv[argv.index("--steps") + 1])
    assert out["steps"] == steps and out["n_gpus"] == 1 and out["unit"] == "particle-steps/sec"
    assert out["dtype"] == "f64" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert "configs[1]" in out["config"]["workload"] and "model" not in out["config"]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    r = out["roofline"]
    assert r is not None and r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic_source"] is not None and (r["traffic"] is None) == isinstance(r["traffic_source"], str)   # a figure only with its provenance
    c = out["cpu_baseline"]
    assert c is not None and c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert set(out["resample_gather_kernel"]) == {"multinomial", "multinomial_sorted", "stratified"}
    sv = out["multinomial_sorted_variant"]                          # the opt-in sorted multinomial as a named variant beside the unchanged headline
    assert sv is not None and sv["value"] > 0 and "multinomial_sorted" in sv["workload"] and "configs[1]" in out["config"]["workload"]
    ss = out["stratified_sort_particles_variant"]                   # :stratified with the reference's default sort_particles=true
    assert ss is not None and ss["value"] > 0 and "sort_particles=true" in ss["workload"]
    assert np.isfinite(out["log_ml_estimate"]) and np.isfinite(out["log_ml_abs_error"])


# ---- `python bench.py --gpus N` with no WORLD_SIZE: the launcher branch (the driver's scaling command; round 2 exited rc 1 on it)
def _fake_torchrun(tmp_path, body):
    """a stand-in for `python -m torch.distributed.run` on sys.path of the child: records its argv, then runs `body`"""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text("import sys, json, os, time\nargv = sys.argv[1:]\n" + body)
    return str(tmp_path)


def _run_launcher(tmp_path, body, extra=()):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=_fake_torchrun(tmp_path, body))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", *extra],
                          capture_output=True, text=True, env=env, timeout=120)


def test_launcher_starts_ranks_and_relays_the_json_line(tmp_path):
    body = ("print('RCCL version banner')\n"
            "assert '--nproc-per-node' in argv and argv[argv.index('--nproc-per-node') + 1] == '2', argv\n"
            "assert '--master-addr' in argv and argv[argv.index('--master-addr') + 1] == '127.0.0.1'\n"
            "assert argv[-6:] == ['--gpus', '2', '--steps', '20', '--warmup', '5'], argv\n"
            "assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
            "print(json.dumps({'metric': 'particle-steps/sec', 'value': 1.0, 'n_gpus': 2}))\n"
            "print('trailing noise')\n")
    p = _run_launcher(tmp_path, body)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2          # the JSON line is the only (so the last) stdout line
    assert "RCCL version banner" in p.stderr and "trailing noise" in p.stderr


def test_launcher_propagates_a_rank_failure(tmp_path):
    p = _run_launcher(tmp_path, "print('rank 1 died', file=sys.stderr)\nsys.exit(3)\n")
    assert p.returncode == 3 and p.stdout.strip() == ""


def test_launcher_needs_a_json_line(tmp_path):
    p = _run_launcher(tmp_path, "print('no result')\n")
    assert p.returncode != 0 and "without a JSON line" in p.stderr


def test_launcher_kills_a_hung_job(tmp_path):
    import time
    t0 = time.time()
    p = _run_launcher(tmp_path, "time.sleep(600)\n", extra=("--launch-timeout", "3"))
    assert p.returncode == 124 and time.time() - t0 < 60 and "killed" in p.stderr


def test_rank_count_mismatch_is_an_error(monkeypatch, built):
    monkeypatch.setenv("WORLD_SIZE", "1")
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value)
