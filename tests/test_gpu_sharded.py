"""Sharded HIP path on ONE GPU: two ranks (two processes, both on cuda:0) run the real shard kernels
(gpf_shard_* through the C ABI); the concatenated shards must equal the single-shard oracle bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import shard_worker_gpu  # noqa: E402
from test_sharded_gloo import CASES, _check_single, free_port, single, single_local, single_skew  # noqa: E402
import shard_worker  # noqa: E402
from conftest import soak_grid  # noqa: E402

# The driver's `pytest -m gpu` keeps ONE representative of every code path below; the rest of each matrix (more worlds / transports / patterns / seeds
# of the same path) carries gpu_soak and runs with `-m "gpu or gpu_soak"` (tools/gpu.sh soak).  Every multi-process case costs ~2 s of process start-up.
_cid = lambda v: (f"{v[0]}-{v[1]}-{v[4]}-{v[5]}" if isinstance(v, tuple) else str(v))  # noqa: E731


@pytest.mark.parametrize("case", [pytest.param(c, marks=() if c in (CASES[0], CASES[1], CASES[2], CASES[3], CASES[4], CASES[7]) else (pytest.mark.gpu_soak,)) for c in CASES],
                         ids=[f"{c[0]}-{c[1]}" for c in CASES])
def test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=2, one_call=False):
    model_name, method, n_global, T, ess_frac, rejuv = case
    n_global *= 20                       # a few scan tiles per shard
    mp.spawn(shard_worker_gpu.run, args=(world, free_port(), model_name, method, n_global, T, ess_frac, rejuv, str(tmp_path), "gloo", None, one_call),
             nprocs=world, join=True)
    f, ess_log, lml_log = single(g, o, model_name, method, n_global, T, ess_frac, rejuv)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    rows = np.concatenate([p["rows"] for p in parts]); lw = np.concatenate([p["lw"] for p in parts])
    parents = np.concatenate([p["parents"] for p in parts])
    assert np.array_equal(parents, f.parents)
    assert np.array_equal(rows, f.rows) and np.array_equal(lw, f.lw)
    for p in parts:
        assert (one_call or np.array_equal(p["ess"], ess_log)) and np.array_equal(p["lml"], lml_log)


@pytest.mark.parametrize("case,world,mode", soak_grid([CASES[3], CASES[5], CASES[4], CASES[0], CASES[7]], [2, 3], ["mailbox", "rccl", "python"],
                                                    keep=lambda c, w, m: (m == "mailbox" and ((w == 2 and c in (CASES[3], CASES[5], CASES[7])) or (w == 3 and c == CASES[3]))) or (w == 2 and c == CASES[3])), ids=_cid)
def test_sharded_step_ess_equals_single_oracle(g, o, tmp_path, monkeypatch, loopback_lib, case, world, mode):
    """gpf_shard_step_ess / sharded.pf_step_ess -- one README-loop iteration per call on every rank, the GLOBAL ESS verdict formed by the summary
    reduction after it has exchanged the shard totals through the mailboxes (k_sum_reduce<SHARD>), the propagate speculatively behind it: ESS-triggered
    residual / stratified + MH on the bearings model (BASELINE configs[3]'s loop), resample-every-step cases (threshold 1.1), 2 - 3 ranks on one GPU.
    mailbox: the fast path; rccl: summaries through all-gathers -> the plain sequence inside the call; python: the python engine's sequence.
    Bit-identical to the single-shard oracle run of the same loop."""
    if mode != "python":
        monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
        if mode == "rccl":
            monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")
    else:
        monkeypatch.setenv("GPF_SHARD_ENGINE", "python")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world, one_call=True)
    if mode != "python":
        assert all(str(np.load(os.path.join(tmp_path, f"rank{r}.npz"))["summaries"]) == mode for r in range(world))


@pytest.mark.parametrize("case", [CASES[3], CASES[5]], ids=lambda c: f"{c[0]}-{c[1]}")
def test_sharded_step_ess_collecting_reduction(g, o, tmp_path, monkeypatch, loopback_lib, case):
    """GPF_SHARD_SUM=collect: the global summary through k_sum_reduce<SHARD> (workgroup 0 collects tagged partials) instead of k_sum_shard (accumulator
    lines, the last workgroup to arrive exchanges and publishes) -- the same verdicts, the same bits"""
    monkeypatch.setenv("GPF_SHARD_SUM", "collect")
    test_sharded_step_ess_equals_single_oracle(g, o, tmp_path, monkeypatch, loopback_lib, case, 3, "mailbox")


def test_world1_sharded_step_ess_and_getters_equal_unsharded(g, o):
    """one shard without a communicator IS the unsharded filter: its getters and its step_ess take the unsharded kernels (k_sum_host, gpf_step_ess);
    with a real 1-rank RCCL communicator (GPF_SHARD_FORCE_COLLECTIVES) the mailbox path runs -- covered by test_rccl_collectives_one_rank's getters"""
    from gpf_amd import sharded
    model = g.models.bearings4(); ys = g.models.simulate(model, 30); N = 40_000
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5, keep_prev=True)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=5, keep_prev=True)
    res = []
    for t in range(1, 30):
        ra = sharded.pf_step_ess(a, (t + 1,), (None,), ys[t], ess_threshold=0.5, method="residual", rejuvenate="move", check=False)
        rb = g.pf_step_ess(b, (t + 1,), (None,), ys[t], ess_threshold=0.5, method="residual", rejuvenate="move", check=False, sort_particles=False)
        assert ra == rb
        res.append(ra)
        if t % 5 == 0:
            assert sharded.get_ess(a) == g.get_ess(b) and sharded.get_lml_est(a) == g.get_lml_est(b)
    assert any(res) and not all(res)
    assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights) and np.array_equal(a.local.parents, b.parents)


@pytest.mark.parametrize("method,n_global,world", [pytest.param(m_, n_, w_, marks=() if ((n_ == 10 and m_ in ("multinomial", "residual")) or (n_ == 3 and m_ == "stratified")) else (pytest.mark.gpu_soak,))
                                                   for m_ in ("multinomial", "stratified", "residual", "multinomial_sorted") for n_, w_ in ((10, 3), (3, 3), (257, 2))])
def test_hip_tiny_shards(g, o, tmp_path, method, n_global, world):
    """shards of 1 to a few particles (fewer slots than a workgroup handles, shard totals that differ a lot)"""
    mp.spawn(shard_worker_gpu.run, args=(world, free_port(), "lgssm2", method, n_global, 5, None, None, str(tmp_path)), nprocs=world, join=True)
    f, ess_log, lml_log = single(g, o, "lgssm2", method, n_global, 5, None, None)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    for p in parts:
        assert np.array_equal(p["ess"], ess_log) and np.array_equal(p["lml"], lml_log)


def test_world1_sharded_equals_unsharded(g, o):
    """G = 1 through sharded.py (no process group) equals the plain single-GPU API."""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 5); N = 30_000
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=5)
    for t in range(1, 5):
        for method in ("multinomial", "stratified", "residual", "multinomial_sorted")[(t - 1) % 4:(t - 1) % 4 + 1]:
            sharded.pf_resample(a, method, check=False)
            kw = dict(sort_particles=False) if method == "stratified" else {}
            g.pf_resample(b, method, check=False, **kw)
        assert np.array_equal(a.local.parents, b.parents)
        sharded.pf_update(a, (t + 1,), (None,), ys[t]); g.pf_update(b, (t + 1,), (None,), ys[t])
        assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights)
    assert sharded.get_lml_est(a) == g.get_lml_est(b) and sharded.get_ess(a) == g.get_ess(b)


@pytest.mark.parametrize("N", [1024, 4096, 1 << 17])
def test_world1_uniform_weights_power_of_two(g, o, N):
    """S = 2^62 exactly (all weights equal, N a power of two): the staged targets, the key table and the descriptors of the
    sharded path against the unsharded one"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=5)
    for method in ("multinomial", "multinomial", "stratified", "residual", "multinomial_sorted", "multinomial"):
        sharded.pf_resample(a, method, check=False)
        g.pf_resample(b, method, check=False, **({"sort_particles": False} if method == "stratified" else {}))
        assert np.array_equal(a.local.parents, b.parents), method
    sharded.pf_update(a, (2,), (None,), ys[1]); g.pf_update(b, (2,), (None,), ys[1])
    assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights)


def test_hip_push_overflow_path(g, o, tmp_path, monkeypatch):
    """send buffer smaller than the exchange: the kernel stops at the capacity, the host repeats the push at the right size"""
    monkeypatch.setenv("GPF_PUSH_CAPACITY", "1000")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, CASES[0])
    test_hip_shards_equal_single_oracle(g, o, tmp_path, CASES[1])    # stratified: the served slot range is cut at the capacity
    test_hip_shards_equal_single_oracle(g, o, tmp_path, CASES[7])    # sorted multinomial: the same, from a tile boundary below the range
    test_world1_sharded_equals_unsharded(g, o)


@pytest.mark.parametrize("case,engine", soak_grid(CASES[:3] + [CASES[7]], ["library", "python"], keep=lambda c, e: (e == "library" and c in (CASES[0], CASES[1])) or (e == "python" and c == CASES[0])), ids=_cid)
def test_rccl_collectives_one_rank(g, o, tmp_path, case, engine):
    """the REAL collectives on RCCL in a 1-rank group: the call path the multi-GPU runs take, as far as a 1-GPU box can
    exercise it.  engine = library: gpf_shard_resample -- ncclAllGather and the grouped ncclSend / ncclRecv exchange issued by
    libgpf on its own communicator (gpf_comm_create), what a Julia host gets for one ccall;  engine = python: sharded.py
    composes the phases with torch.distributed (all_gather_into_tensor, all_to_all_single with split sizes)"""
    model_name, method, n_global, T, ess_frac, rejuv = case
    n_global *= 20
    mp.spawn(shard_worker_gpu.run, args=(1, free_port(), model_name, method, n_global, T, ess_frac, rejuv, str(tmp_path), "nccl", engine),
             nprocs=1, join=True)
    f, ess_log, lml_log = single(g, o, model_name, method, n_global, T, ess_frac, rejuv)
    p = np.load(os.path.join(tmp_path, "rank0.npz"))
    assert np.array_equal(p["parents"], f.parents) and np.array_equal(p["rows"], f.rows) and np.array_equal(p["lw"], f.lw)
    assert np.array_equal(p["ess"], ess_log) and np.array_equal(p["lml"], lml_log)


@pytest.mark.parametrize("world,engine", soak_grid([2, 3], ["library", "python"], keep=lambda w, e: (w, e) in ((2, "library"), (3, "python"))))
def test_sorted_multinomial_many_tiles(g, o, tmp_path, monkeypatch, loopback_lib, world, engine):
    """more than SP_DIRECT_TILES = 1024 tiles of 2048 global slots (the shape of 8 shards of 10^6): the tiles' starting points come from
    k_sorted_tiles and k_sorted_plan searches them in place (window around lo / S, 256-ary rounds behind it) instead of scanning the tile
    totals in LDS; the shard boundaries fall inside tiles"""
    if engine == "library":
        monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
        if world == 3:
            # THREE ranks of 767 K particles on ONE GPU: two ranks' weight scans (every workgroup waits for all (max, flags) pushes) can fill the device while
            # the third rank's one-wave push kernel still waits for a slot -- nobody finishes until the bounded wait gives up (4 of 6 runs on one box; the
            # oversubscription caveat of DESIGN.md 6.7, not a state ranks with a GPU each can reach).  The summaries of this case go through RCCL instead.
            monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")
    monkeypatch.setenv("GPF_SHARD_ENGINE", engine)
    test_hip_shards_equal_single_oracle(g, o, tmp_path, ("lgssm2", "multinomial_sorted", 115_001, 3, None, None), world=world)   # (x 20 inside)


@pytest.fixture(scope="module")
def loopback_lib(tmp_path_factory):
    """tests/loopback_rccl: the nine RCCL entry points libgpf calls, over files in /dev/shm (several ranks on ONE GPU)"""
    import subprocess
    out = tmp_path_factory.mktemp("loopback") / "libloopback_rccl.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(HERE, "loopback_rccl", "loopback_rccl.cpp"),
                    "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("case,world", soak_grid(CASES, [2, 3], keep=lambda c, w: (w == 2 and c not in (CASES[6], CASES[8], CASES[5])) or (w == 3 and c == CASES[3])), ids=_cid)
def test_library_engine_several_ranks_over_loopback(g, o, tmp_path, monkeypatch, loopback_lib, case, world):
    """gpf_shard_resample / gpf_shard_effective_sample_size / gpf_shard_log_ml_estimate -- the library engine, its all-gathers and
    its grouped send / receive exchange with real counts and offsets -- with 2 and 3 ranks.  Real RCCL refuses two ranks on one
    device and the build environment has no multi-GPU box, so the transport under the library is the loopback stand-in; every
    line of libgpf's own multi-rank code runs.  Result: bit-identical to the single-shard oracle."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world)


@pytest.mark.parametrize("case,one_call", soak_grid([CASES[0], CASES[1], CASES[3], CASES[7]], [False, True],
                                                  keep=lambda c, oc: (c, oc) in ((CASES[0], False), (CASES[1], False), (CASES[3], True))), ids=_cid)
def test_library_engine_fused_max_round(g, o, tmp_path, monkeypatch, loopback_lib, case, one_call):
    """GPF_SHARD_FUSE_MF=1: the (max, flags) mailbox round rides in its consumer's launch (workgroup 0 of the weight scan / of k_sum_shard pushes, every
    workgroup waits) instead of k_pack_mflags' own launch -- the form for ranks that have a GPU each (faster by 1 - 2 us per step on one rank); here 2 ranks
    on one GPU at sizes whose launches fit side by side.  The same bits."""
    monkeypatch.setenv("GPF_SHARD_FUSE_MF", "1")
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=2, one_call=one_call)


@pytest.mark.parametrize("case", [CASES[3], CASES[5]], ids=lambda c: f"{c[0]}-{c[1]}")
def test_library_engine_getters_through_the_scan(g, o, tmp_path, monkeypatch, loopback_lib, case):
    """GPF_SHARD_GETTERS=scan: the sharded ESS / log-ML getters through the weight scan, copies and a stream synchronisation (the round-4 form)
    instead of the one-launch reduction that exchanges the totals through the mailboxes (k_sum_reduce<SHARD>) -- the same values"""
    monkeypatch.setenv("GPF_SHARD_GETTERS", "scan")
    test_library_engine_several_ranks_over_loopback(g, o, tmp_path, monkeypatch, loopback_lib, case, 2)


@pytest.mark.parametrize("world,mode", soak_grid([2, 3], ["mailbox", "rccl"], keep=lambda w, m: (w, m) == (2, "rccl")))
def test_library_engine_summary_transport(g, o, tmp_path, monkeypatch, loopback_lib, world, mode):
    """the two ways the (max, flags) / {S, Q} / residual summaries travel between ranks: shard mailboxes (hipIpc-mapped device
    memory, peer stores from the producing kernels, waits in the consuming ones -- the default) and RCCL all-gathers
    (GPF_SHARD_SUMMARY=rccl, also the automatic fallback).  Both must be the mode they claim and give the oracle's result for
    all three resamplers (CASES[0..2]) plus the ESS-triggered bearings run with rejuvenation."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    if mode == "rccl":
        monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")
    else:
        monkeypatch.delenv("GPF_SHARD_SUMMARY", raising=False)
    for case in CASES[:4]:
        test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world)
        for r in range(world):
            assert str(np.load(os.path.join(tmp_path, f"rank{r}.npz"))["summaries"]) == mode


@pytest.mark.parametrize("world,summaries", soak_grid([2, 3], ["mailbox", "rccl"], keep=lambda w, m: (w, m) == (2, "mailbox")))
def test_library_engine_pull_plan(g, o, tmp_path, monkeypatch, loopback_lib, world, summaries):
    """GPF_SHARD_PLAN=pull (gpf.h gpf_comm_set_plan): every shard evaluates its own slots only, requests go to the owners of the
    targets, rows come back -- two exchanges with their own counts and offsets.  The same bits as the push plan and the oracle for
    multinomial, residual (deterministic head + i.i.d. tail), stratified (closed-form plan, unaffected) and the ESS-triggered
    bearings run with rejuvenation; also with a send buffer that is too small (the repeated pass 2)."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_PLAN", "pull")
    if summaries == "rccl":
        monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")
    for case in CASES[:4]:
        test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world)
        for r in range(world):
            assert str(np.load(os.path.join(tmp_path, f"rank{r}.npz"))["plan"]) == "pull"
    monkeypatch.setenv("GPF_PUSH_CAPACITY", "1000")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, CASES[0], world=world)
    test_hip_shards_equal_single_oracle(g, o, tmp_path, CASES[2], world=world)


@pytest.mark.parametrize("method,pattern", soak_grid(["multinomial", "residual"], ["all_on_first_shard", "single_particle", "middle_band"],
                                                    keep=lambda m, p: (m, p) in (("multinomial", "all_on_first_shard"), ("residual", "single_particle"))))
def test_pull_plan_skewed_weights(g, o, tmp_path, monkeypatch, loopback_lib, method, pattern):
    """every request to one shard (the others receive no request at all and serve nothing), pull plan"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_PLAN", "pull")
    test_hip_shards_skewed_weights(g, o, tmp_path, method, pattern)


@pytest.mark.parametrize("case", [CASES[0], CASES[2]], ids=["multinomial", "residual"])
def test_pull_plan_on_rccl_one_rank(g, o, tmp_path, monkeypatch, case):
    """the pull plan's collectives on the REAL RCCL (1-rank communicator: the all-gather of the request counts, the request
    send / receive to itself as ncclUint64)"""
    monkeypatch.setenv("GPF_SHARD_PLAN", "pull")
    test_rccl_collectives_one_rank(g, o, tmp_path, case, "library")
    assert str(np.load(os.path.join(tmp_path, "rank0.npz"))["plan"]) == "pull"


def test_plan_switch_between_resamples(g, o):
    """gpf_comm_set_plan between resamples of one filter (world 1, no communicator): push and pull alternate, all three resamplers,
    against the unsharded filter; the python engine refuses the setting"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 8); N = 50_001
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=5)
    if not a.backend.lib_comm:
        with pytest.raises(g.ErrorException):
            a.backend.set_plan("pull")
        pytest.skip("python engine")
    assert a.backend.plan() == "push"
    with pytest.raises(g.ErrorException):
        a.backend.set_plan("sideways")
    for t in range(1, 8):
        a.backend.set_plan("pull" if t % 2 else "push"); assert a.backend.plan() == ("pull" if t % 2 else "push")
        method = ("multinomial", "residual", "stratified")[t % 3]
        sharded.pf_resample(a, method, check=False)
        g.pf_resample(b, method, check=False, **({"sort_particles": False} if method == "stratified" else {}))
        assert np.array_equal(a.local.parents, b.parents), (t, method)
        sharded.pf_update(a, (t + 1,), (None,), ys[t]); g.pf_update(b, (t + 1,), (None,), ys[t])
        assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights)
    assert sharded.get_lml_est(a) == g.get_lml_est(b)


_SKEW_KEPT = (("multinomial", "all_on_first_shard"), ("stratified", "all_on_first_shard"), ("residual", "single_particle"),
              ("multinomial_sorted", "all_on_first_shard"), ("stratified", "middle_band"), ("multinomial_sorted", "single_particle"))


@pytest.mark.parametrize("method,pattern", soak_grid(["multinomial", "stratified", "residual", "multinomial_sorted"], ["all_on_first_shard", "single_particle", "middle_band"],
                                                    keep=lambda m, p: (m, p) in _SKEW_KEPT[:4]))
def test_library_engine_skewed_weights_over_loopback(g, o, tmp_path, monkeypatch, loopback_lib, method, pattern, n_global=300_000):
    """zero-length sends, one shard serving everything (send-buffer overflow and the repeated push) through the library engine"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    test_hip_shards_skewed_weights(g, o, tmp_path, method, pattern, n_global)


@pytest.mark.parametrize("method,pattern", soak_grid(["multinomial", "stratified", "residual", "multinomial_sorted"], ["all_on_first_shard", "single_particle", "middle_band"],
                                                    keep=lambda m, p: (m, p) in _SKEW_KEPT[:2]))
def test_hip_shards_skewed_weights(g, o, tmp_path, method, pattern, n_global=300_000):
    """one shard owns every target (its push exceeds the balanced-size send buffer: the overflow path runs for real),
    the others own none (zero-length sends)"""
    world = 3                             # 2 n_local + 64 Ki < n_global: the shard that owns everything overflows its send buffer
    mp.spawn(shard_worker_gpu.run_skew, args=(world, free_port(), method, n_global, pattern, str(tmp_path)), nprocs=world, join=True)
    f = single_skew(g, o, method, n_global, pattern)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    assert all(float(p["lml"]) == f.log_ml_estimate() for p in parts)


@pytest.mark.parametrize("method", ["multinomial", "stratified", "residual"])
def test_hip_local_resample_equals_substate_resamples(g, o, tmp_path, method):
    """island mode on the GPU: every shard resamples through a view of itself; equals the oracle's sub-state resamples"""
    world, n_global = 2, 60_001
    mp.spawn(shard_worker_gpu.run_local, args=(world, free_port(), method, n_global, str(tmp_path)), nprocs=world, join=True)
    f, lml = single_local(g, o, method, n_global, world)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    for p in parts:
        assert np.array_equal(p["lml"], lml) and float(p["ess"]) == f.effective_sample_size()


def test_hip_sharded_validity_checks(g, o, tmp_path):
    world, n_global = 2, 50_000
    mp.spawn(shard_worker.run_check, args=(world, free_port(), n_global, str(tmp_path), True), nprocs=world, join=True)
    f = _check_single(g, o, n_global)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    for p in parts:
        assert bool(p["true_raised"]) and bool(p["warned"]) and bool(p["nan_raised"])
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)


@pytest.mark.parametrize("seed,world,n_global,engine", [pytest.param(s_, 2 + s_ % 2, [6000, 6001, 40_000, 2048, 1024, 9999][s_ % 6], e_,
                                                                     marks=() if (s_ < 2 or (s_ < 3 and e_ == "library")) else (pytest.mark.gpu_soak,))
                                                        for s_ in range(int(os.environ.get("GPF_FUZZ_SHARD_SEEDS", "4"))) for e_ in ("library", "library-pull", "python")])
def test_sharded_random_api_sequences(g, o, tmp_path, monkeypatch, loopback_lib, seed, world, n_global, engine):
    """random sequences of updates, global resamples (all four), rejuvenation, global getters, one-call loop iterations (pf_step_ess), island resamples and adversarial
    weight vectors on a sharded filter (2 - 3 ranks on one GPU; library engine over the loopback transport / python engine over
    gloo) against ONE oracle filter: the global resample is the unsharded one bit for bit, the island resample is the sub-state
    resample of each shard's range"""
    if engine.startswith("library"):
        monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
        monkeypatch.setenv("GPF_SHARD_PLAN", "pull" if engine == "library-pull" else "push")
    else:
        monkeypatch.setenv("GPF_SHARD_ENGINE", "python")
    T = 30
    mp.spawn(shard_worker_gpu.run_fuzz, args=(world, free_port(), seed, n_global, T, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    model = g.models.bearings4(); ys = g.models.simulate(model, T + 2)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77, keep_prev=True).initialize(ys[0])
    bounds = np.concatenate([[0], np.cumsum([int(p["n"]) for p in parts])])
    t, scal = 1, []
    for op, method, kind, salt in shard_worker_gpu.fuzz_ops(seed, T):
        if op == "update":
            f.update(ys[t]); t += 1
        elif op == "resample":
            tempered = engine.startswith("library") and bool(salt & 4)
            f.resample(method, check=False, priority_alpha=0.5 if tempered else None,
                       **({"sort_particles": not tempered and bool(salt & 16)} if method == "stratified" else {}))
        elif op == "rejuvenate":
            f.rejuvenate("move", 1)
        elif op == "getters":
            scal.append((f.effective_sample_size(), f.log_ml_estimate()))
        elif op == "step_ess":
            if f.effective_sample_size() < shard_worker_gpu.STEP_ESS_THRESHOLDS[salt % 3] * n_global:
                f.resample(method, check=False, **({"sort_particles": False} if method == "stratified" else {}))
                if salt & 8:
                    f.rejuvenate("move", 1)
            f.update(ys[t]); t += 1
        elif op == "local":
            epoch = f.epoch
            for r in range(world):
                f.epoch = epoch                                  # the shards resample concurrently: same epoch, disjoint global ids
                f[int(bounds[r]):int(bounds[r + 1])].resample(method, check=False, priority_alpha=0.5 if salt & 2 else None,
                                                              **({"sort_particles": bool(salt & 1)} if method == "stratified" else {}))
        else:
            f.lw = shard_worker_gpu.fuzz_weights(kind, n_global, salt).copy()
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw, equal_nan=True)
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    ref = np.array(scal, dtype=np.float64).reshape(-1, 2)
    for p in parts:
        assert np.array_equal(p["scal"], ref, equal_nan=True)
        assert float(p["lml"]) == f.log_ml_estimate() or (np.isnan(float(p["lml"])) and np.isnan(f.log_ml_estimate()))


def test_comm_destroy_after_resample_keeps_the_population(g, o):
    """resample -> gpf_comm_destroy -> update: the deferred commit points into the communicator's exchange buffers; destroying
    the communicator must scatter it first (round-2 advisor: use-after-free through the public ABI)"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 4); N = 50_000
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=9)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=9)
    assert a.backend.lib_comm
    for method in ("multinomial", "stratified"):
        sharded.pf_resample(a, method, check=False)
        g.pf_resample(b, method, check=False, **({"sort_particles": False} if method == "stratified" else {}))
        a.backend._ck(a.backend.L.gpf_comm_destroy(a.backend.h))
        # scribble over freed device memory: a stale pointer would now read garbage
        import torch
        junk = [torch.full((N * 3,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(4)]
        torch.cuda.synchronize()
        sharded.pf_update(a, (2,), (None,), ys[1]); g.pf_update(b, (2,), (None,), ys[1])
        assert np.array_equal(a.local.parents, b.parents)
        assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights)
        del junk
        a.backend._ck(a.backend.L.gpf_comm_create(a.backend.h, None, 0, 1))
    assert g.get_lml_est(b) == sharded.get_lml_est(a)


def _tempered_oracle(g, o, method, n_global, T):
    model = g.models.bearings4(); ys = g.models.simulate(model, T + 1)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77, keep_prev=True).initialize(ys[0])
    kw = {"sort_particles": False} if method == "stratified" else {}
    scal = []
    for t in range(1, T):
        f.resample(method, priority_alpha=0.5 if t % 2 else 0.25, check=False, **kw)
        scal.append((f.effective_sample_size(), f.log_ml_estimate()))
        if t == 2:
            f.rejuvenate("move", 1)
        if t == 3:
            f.resample(method, check=False, **kw)
        f.update(ys[t])
    return f, np.array(scal)


@pytest.mark.parametrize("method,world,mode", soak_grid(["multinomial", "stratified", "residual", "multinomial_sorted"], [2, 3], ["mailbox", "rccl"],
                                                       keep=lambda m, w, md: ((w, md) == (2, "mailbox") and m != "multinomial_sorted") or (w, md, m) == (3, "rccl", "multinomial_sorted")))
def test_sharded_tempered_resample(g, o, tmp_path, monkeypatch, loopback_lib, method, world, mode):
    """priority_fn = w -> alpha w across shards (src/resample.jl:51-52,57,198-200; test/resample.jl:15): ancestors from the
    priorities' global CDF, log-ML from the raw weights, weights from the global logsumexp of log_ws -- three summary rounds and
    one more double per exchanged entry; bit-identical to the unsharded oracle's resample(method, priority_alpha=...)"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    if mode == "rccl":
        monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")
    n_global, T = 30_011, 6
    mp.spawn(shard_worker_gpu.run_tempered, args=(world, free_port(), method, n_global, T, str(tmp_path)), nprocs=world, join=True)
    f, scal = _tempered_oracle(g, o, method, n_global, T)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    for p in parts:
        assert np.array_equal(p["scal"], scal) and float(p["lml"]) == f.log_ml_estimate() and str(p["summaries"]) == mode


@pytest.mark.parametrize("method,world", soak_grid(["multinomial", "residual"], [2, 3], keep=lambda m, w: (m, w) == ("residual", 3)))
def test_sharded_tempered_resample_pull_plan(g, o, tmp_path, monkeypatch, loopback_lib, method, world):
    """tempering through the pull plan: the answered entries carry log_ws like the pushed ones"""
    monkeypatch.setenv("GPF_SHARD_PLAN", "pull")
    test_sharded_tempered_resample(g, o, tmp_path, monkeypatch, loopback_lib, method, world, "mailbox")
    assert all(str(np.load(os.path.join(tmp_path, f"rank{r}.npz"))["plan"]) == "pull" for r in range(world))


@pytest.mark.parametrize("method", ["multinomial", "stratified", "residual", "multinomial_sorted"])
def test_world1_tempered_equals_unsharded(g, o, method):
    """one shard, no communicator: the tempered sharded resample against the plain device API"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 4); N = 40_000
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=5)
    kw = {"sort_particles": False} if method == "stratified" else {}
    for t in range(1, 4):
        sharded.pf_resample(a, method, priority_fn=g.Tempering(0.5), check=False)
        g.pf_resample(b, method, priority_fn=g.Tempering(0.5), check=False, **kw)
        assert np.array_equal(a.local.parents, b.parents) and np.array_equal(a.local.log_weights, b.log_weights)
        assert sharded.get_lml_est(a) == g.get_lml_est(b)
        sharded.pf_update(a, (t + 1,), (None,), ys[t]); g.pf_update(b, (t + 1,), (None,), ys[t])
        assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights)
    with pytest.raises(g.ErrorException):
        sharded.pf_resample(a, method, priority_fn=lambda w: 0.3 * w)


# ---- the window exchange (gpf.h gpf_comm_set_exchange): boundary slabs of the resamplers with ascending targets as peer stores into
#      the destination ranks' slot-addressed receive windows -- no host wait, no ncclGroup
WINDOW_CASES = [CASES[1], CASES[7], CASES[8], CASES[5],
                ("bearings4", "stratified", 1600, 5, None, "keep"),          # rows of 8 doubles (x_t and x_{t-1}) straight from the window into the next propagate
                ("bearings4", "multinomial_sorted", 2100, 5, 0.7, "move")]


@pytest.mark.parametrize("case,world,mode", soak_grid(WINDOW_CASES, [2, 3], ["p2p", "rccl"],
                                                    keep=lambda c, w, m: (m == "p2p" and ((w == 2 and c in (WINDOW_CASES[1], WINDOW_CASES[2], WINDOW_CASES[3], WINDOW_CASES[5])) or (w == 3 and c in (WINDOW_CASES[0], WINDOW_CASES[4])))) or
                                                                         (m == "rccl" and w == 2 and c == WINDOW_CASES[1])), ids=_cid)
def test_window_exchange_equals_single_oracle(g, o, tmp_path, monkeypatch, loopback_lib, case, world, mode):
    """stratified / sorted multinomial across 2 - 3 ranks on one GPU through the library engine, the rows of the boundary slabs stored by the serving
    rank's merge kernel straight into the holding rank's receive window (hipIpc-mapped, sealed entries) and read there by that rank's next propagate
    (k_step<GATHER> with PackedCommit::ring) or materialised commit (k_commit_ring) -- against the grouped send / receive (GPF_SHARD_EXCHANGE=rccl) and
    the single-shard oracle: the same bits, and each run is in the mode it claims"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_EXCHANGE", mode)
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert all(str(p["exchange"]) == mode for p in parts)
    # what crossed the shard boundaries is the same either way, and every entry sent was received by somebody
    sent, recv = sum(int(p["traffic"][1]) for p in parts), sum(int(p["traffic"][2]) for p in parts)
    assert sent == recv and int(parts[0]["traffic"][0]) > 0
    D = {"lgssm2": 2, "sv1": 1, "bearings4": 4}[case[0]]
    W = ((2 * D if case[5] is not None else D) + 1) & ~1                          # (rows carry x_{t-1} whenever the run could rejuvenate)
    assert all(int(p["traffic"][3]) == 8 * (W + (2 if mode == "p2p" else 1)) for p in parts)       # [row | ancestor | seal] against [row | slot, ancestor]


@pytest.mark.parametrize("case,one_call", soak_grid([CASES[5], WINDOW_CASES[-1]], [False, True], keep=lambda c, oc: oc), ids=_cid)
def test_window_exchange_in_the_one_call_loop(g, o, tmp_path, monkeypatch, loopback_lib, case, one_call):
    """BASELINE configs[3]'s loop shape with a resampler whose exchange goes through the windows: gpf_shard_step_ess (verdict on the device, speculative
    propagate) around gpf_shard_resample's window exchange and the MH sweep that materialises it"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=3, one_call=one_call)
    assert all(str(np.load(os.path.join(tmp_path, f"rank{r}.npz"))["exchange"]) == "p2p" for r in range(3))


@pytest.mark.parametrize("method,n_global,world", [pytest.param(m_, n_, w_, marks=() if ((n_ == 10 and m_ == "multinomial_sorted") or (n_ == 4099 and m_ == "stratified") or (n_ == 3 and m_ == "stratified")) else (pytest.mark.gpu_soak,))
                                                   for m_ in ("stratified", "multinomial_sorted") for n_, w_ in ((10, 3), (3, 3), (257, 2), (4099, 3))])
def test_window_exchange_tiny_shards(g, o, tmp_path, monkeypatch, loopback_lib, method, n_global, world, expect="p2p"):
    """shards of 1 to a few particles: own ranges that are empty, a shard served entirely by its neighbours"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib)
    mp.spawn(shard_worker_gpu.run, args=(world, free_port(), "lgssm2", method, n_global, 5, None, None, str(tmp_path), "gloo", "library"), nprocs=world, join=True)
    f, ess_log, lml_log = single(g, o, "lgssm2", method, n_global, 5, None, None)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert all(str(p["exchange"]) == expect for p in parts)
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    for p in parts:
        assert np.array_equal(p["ess"], ess_log) and np.array_equal(p["lml"], lml_log)


@pytest.mark.parametrize("method,pattern", soak_grid(["stratified", "multinomial_sorted"], ["all_on_first_shard", "single_particle", "middle_band"],
                                                    keep=lambda m, p: (m, p) in (("stratified", "all_on_first_shard"), ("multinomial_sorted", "middle_band"))))
def test_grouped_exchange_skewed_weights_over_loopback(g, o, tmp_path, monkeypatch, loopback_lib, method, pattern):
    """the skew cases of test_library_engine_skewed_weights_over_loopback (which now run through the windows: a slab as large as a whole shard needs no
    capacity there) through the grouped send / receive with its overflowing send buffer and repeated push"""
    monkeypatch.setenv("GPF_SHARD_EXCHANGE", "rccl")
    test_library_engine_skewed_weights_over_loopback(g, o, tmp_path, monkeypatch, loopback_lib, method, pattern)


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_mode_switch_between_resamples(g, o, tmp_path, monkeypatch, loopback_lib, world):
    """gpf_comm_set_exchange between the resamples of one filter: windows and grouped send / receive alternate (the window's sequence numbers skip
    the grouped rounds), deferred and materialised commits alternate; against one oracle filter"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    n_global, T = 50_003, 14
    mp.spawn(shard_worker_gpu.run_exchange_switch, args=(world, free_port(), "bearings4", n_global, T, str(tmp_path)), nprocs=world, join=True)
    model = g.models.bearings4(); ys = g.models.simulate(model, T + 1)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77, keep_prev=True).initialize(ys[0])
    lml = []
    for t in range(1, T):
        meth = shard_worker_gpu.EXCHANGE_SWITCH_METHODS[t % 4]
        f.resample(meth, check=False, **({"sort_particles": False} if meth == "stratified" else {}))
        if t % 4 == 0:
            lml.append(f.log_ml_estimate())
        if t % 5 == 0:
            f.rejuvenate("move", 1)
        f.update(ys[t])
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents)
    assert np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
    assert np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw)
    for p in parts:
        assert np.array_equal(p["lml"], np.array(lml)) and float(p["lml_end"]) == f.log_ml_estimate()


def test_exchange_mode_api(g, o):
    """one shard: without a communicator there are no windows (and nothing to exchange) -- the mode reads rccl and p2p is refused; the python engine has no switch"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    a = sharded.pf_initialize(model, (1,), ys[0], 10_000, seed=5)
    if not a.backend.lib_comm:
        with pytest.raises(g.ErrorException):
            a.backend.set_exchange("p2p")
        return
    assert a.backend.exchange() == "rccl"
    with pytest.raises(g.ErrorException):
        a.backend.set_exchange("p2p")
    with pytest.raises(g.ErrorException):
        a.backend.set_exchange("carrier pigeon")
    a.backend.set_exchange("rccl")


@pytest.mark.parametrize("case,one_call", soak_grid([CASES[3], CASES[5], CASES[0]], [False, True], keep=lambda c, oc: (c == CASES[3] and oc) or (c == CASES[5] and oc)), ids=_cid)
def test_resample_behind_an_ess_read_with_and_without_summary_reuse(g, o, tmp_path, monkeypatch, loopback_lib, case, one_call):
    """An ESS read in front of a resample (README.md:68-70) has exchanged (max, flags) and {S, limbs} already (k_sum_shard): gpf_shard_resample reuses that
    round -- no second (max, flags) exchange, and for :residual no weight scan at all (k_scan_residual2<DIRECT> with the global S from the host and the
    maximum folded from the gathered pairs).  GPF_SHARD_REUSE_SUMMARY=0 keeps the plain sequence: both equal the single-shard oracle bit for bit, through
    the separate calls (get_ess; pf_resample!; ...) and through the one call (gpf_shard_step_ess), 3 ranks on one GPU."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    for reuse in ("1", "0"):
        monkeypatch.setenv("GPF_SHARD_REUSE_SUMMARY", reuse)
        test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=3, one_call=one_call)


IID_WINDOW_CASES = [CASES[0], CASES[2], CASES[3], CASES[4], CASES[6], CASES[1]]      # (the last one: ascending targets keep using the windows under p2p_all)


@pytest.mark.parametrize("case,world,one_call", soak_grid(IID_WINDOW_CASES, [2, 3], [False, True],
                                                        keep=lambda c, w, oc: (not oc and ((w == 2 and c in (CASES[0], CASES[3], CASES[4])) or (w == 3 and c == CASES[2]))) or (oc and w == 3 and c == CASES[3])), ids=_cid)
def test_iid_rows_through_the_windows(g, o, tmp_path, monkeypatch, loopback_lib, case, world, one_call):
    """GPF_SHARD_EXCHANGE=p2p_all (gpf.h GPF_SHARD_EXCHANGE_P2P_ALL): the i.i.d. resamplers' rows -- :multinomial, :residual's tail and head -- also go
    straight from the look-up kernels (k_push_multi / k_push) into the window slot of the rank that holds the slot, the commit reads the window wherever the
    own-slot search left -1 (k_step<GATHER> masked, k_move_step, k_commit_ring): no host wait, no ncclGroup for them either.  Bit-identical to the oracle."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_EXCHANGE", "p2p_all")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world, one_call=one_call)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert all(str(p["exchange"]) == "p2p_all" for p in parts)
    sent, recv = sum(int(p["traffic"][1]) for p in parts), sum(int(p["traffic"][2]) for p in parts)
    assert sent == recv and sent > 0 and int(parts[0]["traffic"][0]) > 0


@pytest.mark.parametrize("method,pattern", soak_grid(["multinomial", "residual"], ["all_on_first_shard", "single_particle", "middle_band"],
                                                    keep=lambda m, p: (m, p) == ("residual", "all_on_first_shard")))
def test_iid_rows_through_the_windows_skewed(g, o, tmp_path, monkeypatch, loopback_lib, method, pattern):
    """one shard serving every slot of the others (a whole shard's worth of window entries from one peer), others serving nothing.
    Sized for ranks that SHARE a GPU (this test): every slot of the served ranks waits in their propagate for a 1024-thread look-up kernel of the serving
    rank, and on one device the waiting workgroups hold the registers that kernel needs -- at 10^5 slots per rank it never gets a CU (a window wait that
    times out after seconds, found here); ranks with a GPU each -- the only configuration RCCL accepts -- wait on their own device for a peer's."""
    monkeypatch.setenv("GPF_SHARD_EXCHANGE", "p2p_all")
    test_library_engine_skewed_weights_over_loopback(g, o, tmp_path, monkeypatch, loopback_lib, method, pattern, n_global=45_000)


@pytest.mark.parametrize("method", ["multinomial", pytest.param("residual", marks=pytest.mark.gpu_soak)])
def test_iid_rows_through_the_windows_tiny_shards(g, o, tmp_path, monkeypatch, loopback_lib, method):
    monkeypatch.setenv("GPF_SHARD_EXCHANGE", "p2p_all")
    test_window_exchange_tiny_shards(g, o, tmp_path, monkeypatch, loopback_lib, method, 10, 3, expect="p2p_all")


@pytest.mark.parametrize("world,exchange", soak_grid([2, 3], ["p2p", "rccl"], keep=lambda w, e: (w, e) == (3, "p2p")))
def test_stratified_plan_outside_the_weight_scan(g, o, tmp_path, monkeypatch, loopback_lib, world, exchange):
    """Since round 6 the plan of a sharded stratified resample (served slot range, own range, exchange counts) is derived by the workgroup of the weight scan
    that ends up with the shard total (k_scan MODE 3, ScanExtras::splan) -- every other stratified test of the library engine with mailboxes runs that way.
    GPF_SHARD_PLAN_IN_SCAN=0 keeps the separate k_strat_plan launch (also what RCCL-carried summaries and the per-phase hosts use): the same bits."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_PLAN_IN_SCAN", "0"); monkeypatch.setenv("GPF_SHARD_EXCHANGE", exchange)
    for case in (CASES[1], CASES[5]):
        test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world)


SORTED_CASES = [("lgssm2", "stratified_sorted", 4100, 5, None, None),            # x 20 particles: 82 000, not a multiple of 3 (the padded all-gather)
                ("bearings4", "stratified_sorted", 1600, 6, 0.6, "move"),      # ESS-triggered + MH: the move scatters the deferred commit; rows carry x_{t-1}
                ("sv1", "stratified_sorted", 1500, 4, None, "reweight")]


@pytest.mark.parametrize("case,world,mode", soak_grid(SORTED_CASES, [2, 3], ["mailbox", "rccl"],
                                                    keep=lambda c, w, m: (c, w, m) in ((SORTED_CASES[0], 3, "mailbox"), (SORTED_CASES[1], 2, "mailbox"))), ids=_cid)
def test_sorted_stratified_across_shards(g, o, tmp_path, monkeypatch, loopback_lib, case, world, mode):
    """pf_resample!(state, :stratified; sort_particles = true) -- the reference's default (src/resample.jl:145,156-157) -- on 2 - 3 shards through
    gpf_shard_resample_sorted: every rank gathers all log-weights and runs the unsharded sort + scan + search on them (the replicated plan), rows travel as
    packed entries, own hits in place.  Bit-identical to the single-shard oracle, for shard sizes that divide n_global and that do not."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    if mode == "rccl":
        monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, case, world=world)


@pytest.mark.gpu_soak
def test_sorted_stratified_across_shards_wide_bucket_sort(g, o, tmp_path, monkeypatch, loopback_lib):
    """1.3 M global particles on 3 ranks: every rank's planner sorts them with the bucket sort's wide form (above 1 179 648 keys)"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_SUMMARY", "rccl")               # (three large ranks on ONE GPU: see test_sorted_multinomial_many_tiles)
    test_hip_shards_equal_single_oracle(g, o, tmp_path, ("lgssm2", "stratified_sorted", 65_000, 3, None, None), world=3)


@pytest.mark.parametrize("variant", [pytest.param("own_off", marks=pytest.mark.gpu_soak), "small_send_buffer"])
def test_sorted_stratified_across_shards_packed_paths(g, o, tmp_path, monkeypatch, loopback_lib, variant):
    """the same with every entry packed (GPF_SHARD_OWN=0: own hits travel through the exchange buffer too) and with a send buffer smaller than the exchange
    (the pack kernel stops at the capacity, the host repeats it at the right size)"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    if variant == "own_off":
        monkeypatch.setenv("GPF_SHARD_OWN", "0")
    else:
        monkeypatch.setenv("GPF_PUSH_CAPACITY", "1000")
    test_hip_shards_equal_single_oracle(g, o, tmp_path, SORTED_CASES[0], world=3)


@pytest.mark.parametrize("pattern", ["all_on_first_shard", pytest.param("single_particle", marks=pytest.mark.gpu_soak), pytest.param("middle_band", marks=pytest.mark.gpu_soak)])
def test_sorted_stratified_across_shards_skewed(g, o, tmp_path, monkeypatch, loopback_lib, pattern):
    """all mass on one shard / one particle / a band of EQUAL weights (long runs of equal sort keys: ties by index): one shard serves everything"""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    test_hip_shards_skewed_weights(g, o, tmp_path, "stratified_sorted", pattern, n_global=100_000)


def test_world1_sorted_stratified_equals_unsharded(g, o):
    """one shard through sharded.py: gpf_shard_resample_sorted against gpf_resample(..., sort_particles = 1) on the plain filter, between updates and after
    a getter; the python engine refuses"""
    from gpf_amd import sharded
    model = g.models.lgssm2(); ys = g.models.simulate(model, 6); N = 50_000
    a = sharded.pf_initialize(model, (1,), ys[0], N, seed=5)
    b = g.pf_initialize(model, (1,), ys[0], N, seed=5)
    for t in range(1, 6):
        sharded.pf_resample(a, "stratified", sort_particles=t != 3, check=False)
        g.pf_resample(b, "stratified", sort_particles=t != 3, check=False)
        assert np.array_equal(a.local.parents, b.parents), t
        if t == 4:
            assert sharded.get_lml_est(a) == g.get_lml_est(b)
        sharded.pf_update(a, (t + 1,), (None,), ys[t]); g.pf_update(b, (t + 1,), (None,), ys[t])
        assert np.array_equal(a.local.traces, b.traces) and np.array_equal(a.local.log_weights, b.log_weights)
    assert sharded.get_lml_est(a) == g.get_lml_est(b) and sharded.get_ess(a) == g.get_ess(b)
    with pytest.raises(g.ErrorException, match="priority_fn"):
        sharded.pf_resample(a, "stratified", sort_particles=True, priority_fn=g.Tempering(0.5), check=False)


@pytest.mark.parametrize("what,summaries,exchange", [("fail_mailbox", "rccl", "rccl"), ("fail_windows", "mailbox", "rccl")])
def test_comm_self_test_failure_falls_back_on_every_rank(g, o, tmp_path, monkeypatch, loopback_lib, what, summaries, exchange):
    """gpf_comm_create tries mailboxes and receive windows out before anything relies on them (a few dependent mailbox rounds; one window entry to and from
    every peer).  One rank reporting a failed test (GPF_SHARD_SELFTEST=fail_*: rank 0 votes no) and EVERY rank keeps the RCCL all-gathers / the grouped
    ncclSend / ncclRecv -- agreed through an all-gather, so no rank is left waiting on a transport its peers gave up.  The same bits."""
    monkeypatch.setenv("GPF_RCCL_LIBRARY", loopback_lib); monkeypatch.setenv("GPF_SHARD_ENGINE", "library")
    monkeypatch.setenv("GPF_SHARD_SELFTEST", what)
    test_hip_shards_equal_single_oracle(g, o, tmp_path, CASES[1], world=3)
    for r in range(3):
        p = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert str(p["summaries"]) == summaries and str(p["exchange"]) == exchange
