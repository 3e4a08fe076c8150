"""Worker for tests/test_gpu_sharded.py: one rank of a 2-process job, BOTH ranks on cuda:0 with the HIP shard
backend; collectives staged through gloo (two ranks cannot share one GPU under RCCL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, port, model_name, method, n_global, T, ess_frac, rejuv, out_dir, backend="gloo", engine=None, one_call=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    if engine:
        os.environ["GPF_SHARD_ENGINE"] = engine            # "library": gpf_shard_resample (libgpf's own RCCL communicator); "python": sharded.py composes the phases
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        os.environ["GPF_SHARD_FORCE_COLLECTIVES"] = "1"     # read at import of gpf_amd.sharded
    import torch
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.by_name(model_name)
        ys = g.models.simulate(model, T)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, keep_prev=rejuv is not None, device=0)
        assert st.backend.lib_comm == (os.environ.get("GPF_SHARD_ENGINE", "python" if backend == "gloo" else "library") == "library")
        ess_log, lml_log = [], []
        for t in range(1, T):
            if one_call:                                         # the loop body as ONE call per rank (gpf_shard_step_ess): no separate ESS read
                ess_log.append(float("nan"))
                sharded.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=1.1 if ess_frac is None else ess_frac, method=method,
                                    rejuvenate=None if rejuv in (None, "keep") else rejuv, check=False)
                lml_log.append(sharded.get_lml_est(st))
                continue
            ess = sharded.get_ess(st); ess_log.append(ess)
            if ess_frac is None or ess < ess_frac * n_global:
                if method == "stratified_sorted":                     # the reference's default order of the strata: gpf_shard_resample_sorted
                    sharded.pf_resample(st, "stratified", sort_particles=True, check=False)
                else:
                    sharded.pf_resample(st, method, check=False)
                if rejuv and rejuv != "keep":
                    sharded.pf_rejuvenate(st, None, (), 1, method=rejuv)
            sharded.pf_update(st, (t + 1,), (None,), ys[t])
            lml_log.append(sharded.get_lml_est(st))
        loc = st.local
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents,
                 gid0=st.gid0, ess=np.array(ess_log), lml=np.array(lml_log), summaries=st.backend.summary_mode(), plan=st.backend.plan(),
                 exchange=st.backend.exchange(), traffic=np.array(st.backend.traffic() if st.backend.lib_comm else (0, 0, 0, 0)))
    finally:
        dist.destroy_process_group()


EXCHANGE_SWITCH_METHODS = ("stratified", "multinomial_sorted", "multinomial", "residual")


def run_exchange_switch(rank, world, port, model_name, n_global, T, out_dir):
    """gpf_comm_set_exchange between the resamples of one sharded filter (library engine): the receive windows and the grouped send / receive
    alternate, stratified and sorted multinomial alternate, a getter or a rejuvenation now and then forces the materialised commit"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.by_name(model_name); ys = g.models.simulate(model, T + 1)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, keep_prev=True, device=0)
        assert st.backend.lib_comm and st.backend.exchange() == "p2p"
        lml = []
        for t in range(1, T):
            st.backend.set_exchange(("p2p", "p2p_all", "rccl")[t % 3])
            sharded.pf_resample(st, EXCHANGE_SWITCH_METHODS[t % 4], check=False)
            if t % 4 == 0:
                lml.append(sharded.get_lml_est(st))                   # (materialises the deferred commit out of the window / the receive buffer)
            if t % 5 == 0:
                sharded.pf_rejuvenate(st, None, (), 1, method="move")
            sharded.pf_update(st, (t + 1,), (None,), ys[t])
        loc = st.local
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents, gid0=st.gid0,
                 lml=np.array(lml), lml_end=sharded.get_lml_est(st), traffic=np.array(st.backend.traffic()))
    finally:
        dist.destroy_process_group()


def skew_weights(n_global, pattern):
    """global log-weight vectors that put all (or nothing) of the mass on single shards"""
    i = np.arange(n_global, dtype=np.float64)
    if pattern == "all_on_first_shard":
        return np.where(i < n_global // 7, -0.001 * i, -np.inf)
    if pattern == "single_particle":
        return np.where(i == n_global - 2, 0.0, -800.0)
    if pattern == "middle_band":
        return np.where((i > 0.45 * n_global) & (i < 0.55 * n_global), 0.0, -40.0)
    raise ValueError(pattern)


def run_skew(rank, world, port, method, n_global, pattern, out_dir):
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.lgssm2()
        ys = g.models.simulate(model, 3)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, device=0)
        assert st.backend.lib_comm == (os.environ.get("GPF_SHARD_ENGINE", "python") == "library")
        loc = st.local
        loc.log_weights = skew_weights(n_global, pattern)[st.gid0:st.gid0 + st.n_local]
        if method == "stratified_sorted":
            sharded.pf_resample(st, "stratified", sort_particles=True, check=False)
        else:
            sharded.pf_resample(st, method, check=False)
        sharded.pf_update(st, (2,), (None,), ys[1])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents, gid0=st.gid0,
                 lml=sharded.get_lml_est(st))
    finally:
        dist.destroy_process_group()


def run_local(rank, world, port, method, n_global, out_dir):
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.lgssm2(); ys = g.models.simulate(model, 5)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, device=0)
        lml = []
        for t in range(1, 5):
            sharded.pf_resample(st, method, check=False, local=True, sort_particles=(t % 2 == 0))
            sharded.pf_update(st, (t + 1,), (None,), ys[t])
            lml.append(sharded.get_lml_est(st))
        loc = st.local
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents, gid0=st.gid0,
                 n=st.n_local, lml=np.array(lml), ess=sharded.get_ess(st))
    finally:
        dist.destroy_process_group()


STEP_ESS_THRESHOLDS = (0.0, 0.5, 1.1)


def fuzz_ops(seed, T):
    """the operation list of one sharded fuzz run (the same on every rank and in the parent's oracle run)"""
    rng = np.random.default_rng(7000 + seed)
    ops = []
    for _ in range(T):
        op = str(rng.choice(["update", "resample", "rejuvenate", "getters", "local", "set_weights", "step_ess"], p=[0.25, 0.25, 0.1, 0.1, 0.1, 0.1, 0.1]))
        ops.append((op, str(rng.choice(["multinomial", "stratified", "residual", "multinomial_sorted"])), str(rng.choice(["equal", "one heavy", "some -inf", "wide"])),
                    int(rng.integers(1 << 30))))
    return ops


def fuzz_weights(kind, n_global, salt):
    r = np.random.default_rng(salt)
    i = np.arange(n_global, dtype=np.float64)
    return {"equal": np.full(n_global, -3.25), "one heavy": np.where(i == salt % n_global, 0.0, -745.0),
            "some -inf": np.where(r.random(n_global) < 0.7, -np.inf, -r.random(n_global)), "wide": -700.0 * r.random(n_global)}[kind]


def run_fuzz(rank, world, port, seed, n_global, T, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.bearings4(); ys = g.models.simulate(model, T + 2)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, keep_prev=True, device=0)
        t, scal = 1, []
        for op, method, kind, salt in fuzz_ops(seed, T):
            if op == "update":
                sharded.pf_update(st, (t + 1,), (None,), ys[t]); t += 1
            elif op == "resample":
                # (library engine: every other global resample is tempered, priority_fn = w -> w / 2)
                tempered = st.backend.lib_comm and bool(salt & 4)
                # (every other untempered stratified one with the reference's default sort_particles = true: the replicated plan, either engine)
                sharded.pf_resample(st, method, check=False, priority_fn=g.Tempering(0.5) if tempered else None,
                                    sort_particles=method == "stratified" and not tempered and bool(salt & 16))
            elif op == "rejuvenate":
                sharded.pf_rejuvenate(st, None, (), 1, method="move")
            elif op == "getters":
                scal.append((sharded.get_ess(st), sharded.get_lml_est(st)))
            elif op == "step_ess":                           # the README loop's body as one call: never / ESS < N/2 / always resampling, with or without the MH sweep
                sharded.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=STEP_ESS_THRESHOLDS[salt % 3], method=method,
                                    rejuvenate="move" if salt & 8 else None, check=False); t += 1
            elif op == "local":
                sharded.pf_resample(st, method, check=False, local=True, sort_particles=bool(salt & 1),
                                    priority_fn=g.Tempering(0.5) if salt & 2 else None)        # (with a priority: through a view of the shard)
            else:
                st.local.log_weights = fuzz_weights(kind, n_global, salt)[st.gid0:st.gid0 + st.n_local]
        loc = st.local
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents, n=st.n_local,
                 scal=np.array(scal, dtype=np.float64).reshape(-1, 2), lml=sharded.get_lml_est(st))
    finally:
        dist.destroy_process_group()


def run_tempered(rank, world, port, method, n_global, T, out_dir):
    """pf_resample!(state, method; priority_fn = w -> w / 2) across shards (library engine), with updates and the global getters between"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.bearings4(); ys = g.models.simulate(model, T + 1)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, keep_prev=True, device=0)
        scal = []
        for t in range(1, T):
            sharded.pf_resample(st, method, priority_fn=g.Tempering(0.5 if t % 2 else 0.25), check="warn")
            scal.append((sharded.get_ess(st), sharded.get_lml_est(st)))
            if t == 2:
                sharded.pf_rejuvenate(st, None, (), 1, method="move")
            if t == 3:
                sharded.pf_resample(st, method, check=False)              # a plain resample right after a tempered one
            sharded.pf_update(st, (t + 1,), (None,), ys[t])
        loc = st.local
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents, n=st.n_local,
                 scal=np.array(scal), lml=sharded.get_lml_est(st), summaries=st.backend.summary_mode(), plan=st.backend.plan())
    finally:
        dist.destroy_process_group()
