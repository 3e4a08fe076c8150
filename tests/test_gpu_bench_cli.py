"""The driver's own bench command on the GPU box: `python bench.py --gpus 1 --steps 20 --warmup 5` must exit 0 and its
LAST stdout line must be the JSON line with roofline + cpu_baseline (BENCH_r01 had rc 1 on exactly this command)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("argv", [["--gpus", "1", "--steps", "20", "--warmup", "5"], ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]])
def test_driver_command_line(built, argv):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["steps"] == int(argv[argv.index("--steps") + 1]) and out["n_gpus"] == 1
    assert out["config"]["particles_per_gpu"] == 1_000_000 and "configs[1]" in out["config"]["workload"]
    assert out["dtype"] == "f64" and out["value"] > 5e7                     # north_star: >= 50 M particle-steps/s on one GPU
    r, c = out["roofline"], out["cpu_baseline"]
    assert r and r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["kernel"]
    if "--no-cpu-baseline" in argv:                                         # (the 13 s CPU leg once is enough)
        assert c is None
    else:
        assert c and c["cores"] == 1 and c["kind"] == "port" and c["value"] > 0
    assert out["log_ml_abs_error"] < 1.0


def _last_json(stdout):
    return json.loads([ln for ln in stdout.strip().splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_two_ranks_one_device_smoke(built):
    """the driver's multi-GPU command line with two ranks, both on cuda:0 (collectives staged through gloo: RCCL refuses two
    ranks on one device): exercises rank/world plumbing, the sharded legs and the barrier + max-over-ranks timing"""
    env = dict(os.environ, GPF_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--particles-per-gpu", "200000"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["config"]["particles_total"] == 400000
    assert out["scaling"] == "weak" and out["value"] > 0
    assert out["stratified_variant"]["value"] > 0 and out["local_resample_variant"]["value"] > 0
    assert out["shard_engine"].startswith("python") and out["cpu_baseline"] is None


@pytest.mark.gpu
def test_sharded_path_with_rccl_one_rank(built):
    """one rank, the sharded code path, every collective issued by the library on a real (1-rank) RCCL communicator"""
    env = dict(os.environ, GPF_BENCH_FORCE_SHARDED="1", GPF_SHARD_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29733",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 1 and out["shard_engine"].startswith("library") and out["value"] > 5e7
    assert out["roofline"]["kernel"] in ("k_push", "k_push_scan", "k_step", "k_scan")
    assert abs(out["log_ml_abs_error"]) < 1.0


@pytest.mark.gpu
def test_two_ranks_library_engine_over_loopback(built, tmp_path):
    """`bench.py --gpus 2` through the LIBRARY engine (gpf_shard_resample: what the driver's multi-GPU runs take), two ranks on
    cuda:0 with tests/loopback_rccl under it; and the guard: with a transport that cannot be loaded every rank falls back to
    the torch.distributed engine together and says so"""
    lib = tmp_path / "libloopback_rccl.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "loopback_rccl", "loopback_rccl.cpp"),
                    "-o", str(lib)], check=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29735", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--particles-per-gpu", "200000"]
    env = dict(os.environ, GPF_BENCH_ONE_DEVICE="1", GPF_SHARD_ENGINE="library", GPF_RCCL_LIBRARY=str(lib))
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["shard_engine"].startswith("library") and "fell back" not in out["shard_engine"]
    ref = out["log_ml_estimate"]
    # the same run through the python engine: same filter, same estimate, bit for bit
    p2 = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, GPF_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, timeout=900)
    assert p2.returncode == 0, p2.stderr[-3000:]
    assert _last_json(p2.stdout)["log_ml_estimate"] == ref
    # a transport that does not exist: the guard moves both ranks to the python engine
    env_bad = dict(os.environ, GPF_BENCH_ONE_DEVICE="1", GPF_RCCL_LIBRARY=str(tmp_path / "missing.so"))
    env_bad.pop("GPF_SHARD_ENGINE", None)
    cmd_bad = list(cmd); cmd_bad[cmd_bad.index("29735")] = "29737"
    p3 = subprocess.run(cmd_bad, cwd=ROOT, env=dict(env_bad, GPF_BENCH_TRY_LIBRARY="1"), capture_output=True, text=True, timeout=900)
    assert p3.returncode == 0, p3.stderr[-3000:]
    out3 = _last_json(p3.stdout)
    assert out3["shard_engine"].startswith("python") and "fell back" in out3["shard_engine"] and out3["log_ml_estimate"] == ref


@pytest.mark.gpu
def test_plain_command_self_launches_two_ranks(built):
    """the driver's scaling command as the driver types it -- plain `python bench.py --gpus 2 ...`, no torchrun, no WORLD_SIZE:
    bench.py starts the two ranks itself as a child job (round 2: SystemExit, rc 1).  Both ranks on cuda:0 here."""
    env = dict(os.environ, GPF_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--particles-per-gpu", "200000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])                      # the LAST stdout line, nothing after it
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["config"]["particles_total"] == 400000
    assert out["rccl_ranks"] == 0 and out["shard_engine"].startswith("python")      # gloo staging: no RCCL in this run, and the line says so
    assert out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.gpu
def test_two_ranks_line_explains_itself(built, tmp_path):
    """the driver's scaling command runs `--steps 20`: whatever that run does not print is lost (nobody can attach a profiler to it).  With 20 steps a
    sharded library-engine line must still carry (a) BOTH exchange plans of the headline workload over a fixed 100 steps (gpf_comm_set_plan) and which one
    the headline value used, (b) `phases_us` -- microseconds per step of summaries / plan / pack / host wait for the counts / exchange / commit + propagate --
    for the headline and every timed variant, (c) the measured per-link rate of the grouped send / receive and the mailbox round trip
    (gpf_comm_calibrate), (d) rccl_ranks, shard_summaries, and both slab transports (receive windows against grouped send / receive) for the resamplers
    with ascending targets.  Two ranks on cuda:0 over the loopback transport."""
    lib = tmp_path / "libloopback_rccl.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "loopback_rccl", "loopback_rccl.cpp"),
                    "-o", str(lib)], check=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29739", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
           "--particles-per-gpu", "100000"]
    ests = {}
    for plan in ("push", "pull"):
        env = dict(os.environ, GPF_BENCH_ONE_DEVICE="1", GPF_SHARD_ENGINE="library", GPF_RCCL_LIBRARY=str(lib), GPF_SHARD_PLAN=plan)
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        out = _last_json(p.stdout)
        assert out["shard_engine"].startswith("library") and "fell back" not in out["shard_engine"]
        ep = out["exchange_plans"]
        assert ep["timed"] == plan and ep["push"]["value"] > 0 and ep["pull"]["value"] > 0 and ep["push"]["steps"] == 100
        # the third way -- the push plan's rows through the receive windows (GPF_SHARD_EXCHANGE_P2P_ALL) -- is for ranks with a GPU each: this run says why not
        assert "share one GPU" in ep["push_windows"]["error"]
        ests[plan] = out["log_ml_estimate"]
        # what the exchanges really sent (gpf_comm_traffic) beside the worksheet: the i.i.d. multinomial moves (G-1)/G of a shard's rows, the
        # sorted multinomial and the stratified resampler boundary slabs
        links = out["exchange_bytes_per_link"]
        head = links["multinomial_headline"]
        assert head["entry_bytes"] == 24 and abs(head["observed_entries_out_per_step"] - 50_000) < 2_000       # n (G-1)/G = 50 000 at G = 2
        assert abs(head["observed_bytes_per_link_per_step"] - head["predicted_bytes_per_link_per_step"]) < 0.05 * head["predicted_bytes_per_link_per_step"]
        for name in ("stratified", "multinomial_sorted"):
            assert 0 <= links[name]["observed_entries_out_per_step"] < 5_000, (name, links[name])
        assert out["multinomial_sorted_variant"]["value"] > 0 and out["stratified_variant"]["value"] > 0
        assert out["steps"] == 20 and out["shard_summaries"] == "mailbox" and out["rccl_ranks"] == 0      # (the loopback transport is not RCCL, and the line says so)
        # (b) phases of the headline and of every timed variant
        names = {"summaries", "plan", "pack", "host_wait_counts", "exchange", "commit_propagate", "resamples"}
        for ph in (out["phases_us"], out["stratified_variant"]["phases_us"], out["multinomial_sorted_variant"]["phases_us"], ep["push"]["phases_us"], ep["pull"]["phases_us"]):
            assert set(ph) == names and ph["resamples"] == 30 and ph["summaries"] > 0 and ph["commit_propagate"] > 0
        assert out["phases_us"]["exchange"] > 0 and out["phases_us"]["host_wait_counts"] > 0              # the i.i.d. exchange: grouped send / receive behind a host wait
        # (c) the transports, measured
        cal = out["transport_calibration"]
        assert cal["link_GBps_measured"] > 0 and cal["exchange_us"] > 0 and cal["mailbox_rtt_us"] > 0 and cal["entries_per_peer"] == 50_000
        assert cal["slab_exchange"]["entries_per_peer"] == 4096 and cal["slab_exchange"]["exchange_us"] > 0
        # (d) both slab transports; the window exchange neither waits on the host nor opens a group
        sm = out["slab_exchange_modes"]
        assert sm["timed"] == "p2p" and out["stratified_variant"]["exchange"] == "p2p"
        for meth in ("stratified", "multinomial_sorted"):
            w, r = sm["p2p"][meth], sm["rccl"][meth]
            assert w["value"] > 0 and r["value"] > 0
            assert w["phases_us"]["host_wait_counts"] == 0 and w["phases_us"]["exchange"] == 0 and r["phases_us"]["exchange"] > 0
    assert ests["push"] == ests["pull"]                                       # the same filter, bit for bit
