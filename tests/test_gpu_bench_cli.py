"""The driver's own bench command on the GPU box: `python bench.py --gpus 1 --steps 20 --warmup 5` must exit 0 and its
LAST stdout line must be the JSON line with roofline + cpu_baseline (BENCH_r01 had rc 1 on exactly this command)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("argv", [["--gpus", "1", "--steps", "20", "--warmup", "5"], ["--steps", "1", "--warmup", "0"]])
def test_driver_command_line(built, argv):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["steps"] == int(argv[argv.index("--steps") + 1]) and out["n_gpus"] == 1
    assert out["config"]["particles_per_gpu"] == 1_000_000 and "configs[1]" in out["config"]["workload"]
    assert out["dtype"] == "f64" and out["value"] > 5e7                     # north_star: >= 50 M particle-steps/s on one GPU
    r, c = out["roofline"], out["cpu_baseline"]
    assert r and r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["kernel"]
    assert c and c["cores"] == 1 and c["kind"] == "port" and c["value"] > 0
    assert out["log_ml_abs_error"] < 1.0
