"""Multi-GPU path on CPU: world_size-2 gloo jobs drive genparticlefilters.jl_amd/sharded.py (the real routing
and collectives) with the oracle-backed shard backend; the concatenated shards must equal the single-shard
oracle run BIT FOR BIT (ancestors, rows, weights) -- the sharded spec is independent of the number of shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import shard_worker  # noqa: E402


def free_port():
    """a rendezvous port for one spawned job.  NOT bind(0): that hands out a port of the kernel's ephemeral range (32768-60999), which the kernel may give to
    somebody's outgoing connection -- gloo's own pairs, the next job's store clients -- between this probe and rank 0's listen (seen once in ~2 500 jobs:
    EADDRINUSE).  A random port BELOW that range, checked to be bindable, is only ever taken by another explicit listener."""
    import random
    rng = random.SystemRandom()
    for _ in range(200):
        p = rng.randrange(20000, 32000)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", p))
            except OSError:
                continue
            return p
    raise RuntimeError("no free rendezvous port between 20000 and 32000")


def single(g, o, model_name, method, n_global, T, ess_frac, rejuv):
    model = g.models.by_name(model_name)
    ys = g.models.simulate(model, T)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77, keep_prev=rejuv is not None).initialize(ys[0])
    ess_log, lml_log = [], []
    for t in range(1, T):
        ess = f.effective_sample_size(); ess_log.append(ess)
        if ess_frac is None or ess < ess_frac * n_global:
            # ("stratified_sorted": the reference's default sort_particles = true -- the library engine's replicated plan, tests/test_gpu_sharded.py)
            f.resample(method.replace("_sorted", "") if method == "stratified_sorted" else method, sort_particles=method == "stratified_sorted", check=False)
            if rejuv and rejuv != "keep":
                f.rejuvenate(rejuv, 1)
        f.update(ys[t]); lml_log.append(f.log_ml_estimate())
    return f, np.array(ess_log), np.array(lml_log)


CASES = [
    ("lgssm2", "multinomial", 1001, 5, None, None),
    ("lgssm2", "stratified", 4100, 5, None, None),
    ("lgssm2", "residual", 3000, 5, None, None),
    ("bearings4", "residual", 2000, 6, 0.5, "move"),        # BASELINE config 4 shape: ESS-triggered residual + MH
    ("sv1", "multinomial", 1500, 5, None, "reweight"),       # BASELINE config 5 shape
    ("bearings4", "stratified", 1600, 5, 0.6, "move"),       # ESS-triggered stratified + MH: the deferred packed commit is scattered by the move
    ("bearings4", "multinomial", 1200, 4, None, "keep"),     # rows carry x_{t-1} (keep_prev) but nothing rejuvenates: the
                                                             # resampled population goes straight into the next propagate
    ("lgssm2", "multinomial_sorted", 9001, 5, None, None),   # sorted uniforms across shards: 5 tiles of 2048 global slots, ragged last tile
    ("sv1", "multinomial_sorted", 2500, 5, None, "reweight"),  # BASELINE config 5 shape with the sorted resampler; shard boundaries inside tiles
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-{c[1]}" for c in CASES])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single(g, o, tmp_path, case, world, one_call=False):
    model_name, method, n_global, T, ess_frac, rejuv = case
    port = free_port()
    mp.spawn(shard_worker.run, args=(world, port, model_name, method, n_global, T, ess_frac, rejuv, str(tmp_path), one_call),
             nprocs=world, join=True)
    f, ess_log, lml_log = single(g, o, model_name, method, n_global, T, ess_frac, rejuv)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert [int(p["gid0"]) for p in parts] == sorted(int(p["gid0"]) for p in parts)
    rows = np.concatenate([p["rows"] for p in parts]); lw = np.concatenate([p["lw"] for p in parts])
    parents = np.concatenate([p["parents"] for p in parts])
    assert np.array_equal(parents, f.parents), "sharded ancestors differ from the single-shard run"
    assert np.array_equal(rows, f.rows) and np.array_equal(lw, f.lw)
    for p in parts:                                     # every rank sees the same global summaries
        assert (one_call or np.array_equal(p["ess"], ess_log)) and np.array_equal(p["lml"], lml_log)


@pytest.mark.parametrize("case", [CASES[3], CASES[5]], ids=lambda c: f"{c[0]}-{c[1]}")
def test_sharded_step_ess_equals_single(g, o, tmp_path, case):
    """sharded.pf_step_ess (the README loop's body, README.md:66-77, as one call per rank) through the python engine: the same bits as the
    separate calls and as the single-shard oracle"""
    test_sharded_equals_single(g, o, tmp_path, case, 2, one_call=True)


SORTED_CASES = [("lgssm2", "stratified_sorted", 4100, 5, None, None),            # 4100 is not a multiple of 3: the padded all-gather of the log-weights
                ("bearings4", "stratified_sorted", 1600, 6, 0.6, "move")]


@pytest.mark.parametrize("case", SORTED_CASES, ids=lambda c: f"{c[0]}-{c[1]}")
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sorted_stratified_equals_single(g, o, tmp_path, case, world):
    """pf_resample!(state, :stratified; sort_particles = true) (the reference's default, src/resample.jl:145,156-157) across shards, python engine: every rank
    gathers all log-weights and works the unsharded sort + scan + search out itself (sharded.py's replicated plan), rows through the all-to-all"""
    test_sharded_equals_single(g, o, tmp_path, case, world)


def test_sharded_sorted_stratified_overflow_and_skew(g, o, tmp_path, monkeypatch):
    monkeypatch.setenv("GPF_PUSH_CAPACITY", "7")
    test_sharded_equals_single(g, o, tmp_path, SORTED_CASES[0], 2)
    monkeypatch.delenv("GPF_PUSH_CAPACITY")
    for pattern in ("all_on_first_shard", "middle_band"):
        test_sharded_skewed_weights(g, o, tmp_path, "stratified_sorted", pattern)


def test_sharded_push_overflow_path(g, o, tmp_path, monkeypatch):
    """a send buffer that is too small for the exchange: the counts reveal it and the push is repeated at the right size"""
    monkeypatch.setenv("GPF_PUSH_CAPACITY", "7")
    test_sharded_equals_single(g, o, tmp_path, CASES[2], 2)


def single_skew(g, o, method, n_global, pattern):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77).initialize(ys[0])
    f.lw[:] = shard_worker.skew_weights(n_global, pattern)
    f.resample("stratified" if method == "stratified_sorted" else method, sort_particles=method == "stratified_sorted", check=False)
    f.update(ys[1])
    return f


@pytest.mark.parametrize("pattern", ["all_on_first_shard", "single_particle", "middle_band"])
@pytest.mark.parametrize("method", ["multinomial", "stratified", "residual", "multinomial_sorted"])
def test_sharded_skewed_weights(g, o, tmp_path, method, pattern):
    """shards that own every target (their push overflows the balanced-size send buffer) next to shards that own none"""
    world, n_global = 3, 5000
    mp.spawn(shard_worker.run_skew, args=(world, free_port(), method, n_global, pattern, str(tmp_path)), nprocs=world, join=True)
    f = single_skew(g, o, method, n_global, pattern)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    assert all(float(p["lml"]) == f.log_ml_estimate() for p in parts)


def single_local(g, o, method, n_global, world):
    """the unsharded oracle with pf_resample!(state[shard range], ...) for every shard range (resample.jl:205-218)"""
    from gpf_amd.sharded import shard_range
    model = g.models.lgssm2(); ys = g.models.simulate(model, 5)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77).initialize(ys[0])
    lml = []
    for t in range(1, 5):
        epoch = f.epoch
        for r in range(world):
            gid0, n = shard_range(n_global, r, world)
            f.epoch = epoch                                  # the shards resample concurrently: same epoch, disjoint global ids
            f[gid0:gid0 + n].resample(method, sort_particles=(t % 2 == 0), check=False)
        f.update(ys[t])
        lml.append(f.log_ml_estimate())
    return f, np.array(lml)


@pytest.mark.parametrize("method", ["multinomial", "stratified", "residual"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_local_resample_equals_substate_resamples(g, o, tmp_path, method, world):
    """island mode (the communication-free alternative) == sub-state resamples of the unsharded oracle"""
    n_global = 3001
    mp.spawn(shard_worker.run_local, args=(world, free_port(), method, n_global, str(tmp_path)), nprocs=world, join=True)
    f, lml = single_local(g, o, method, n_global, world)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)      # local to each shard, like the reference's views
    for p in parts:
        assert np.array_equal(p["lml"], lml) and float(p["ess"]) == f.effective_sample_size()


def _check_single(g, o, n_global):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77).initialize(ys[0])
    f.lw[:] = -np.inf
    f.resample("multinomial", check=False)
    return f


def test_sharded_validity_checks(g, o, tmp_path):
    """check = true raises, check = :warn warns and falls back to uniform weights (resample.jl:54-55), NaN on one shard is an
    error on every shard -- all without synchronising the stream (the flags come from pinned memory)"""
    world, n_global = 2, 2500
    mp.spawn(shard_worker.run_check, args=(world, free_port(), n_global, str(tmp_path)), nprocs=world, join=True)
    f = _check_single(g, o, n_global)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    for p in parts:
        assert bool(p["true_raised"]) and bool(p["warned"]) and bool(p["nan_raised"])
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
