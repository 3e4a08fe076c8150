"""Worker for tests/test_sharded_gloo.py: one rank of a world_size-2 gloo job on CPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(rank, world, port, model_name, method, n_global, T, ess_frac, rejuv, out_dir, one_call=False):
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    from oracle_shard_backend import OracleShardBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.by_name(model_name)
        ys = g.models.simulate(model, T)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, keep_prev=rejuv is not None,
                                   backend_factory=OracleShardBackend)
        ess_log, lml_log = [], []
        for t in range(1, T):
            if one_call:                                     # sharded.pf_step_ess: the loop body as one call (python engine: the same sequence)
                ess_log.append(float("nan"))
                sharded.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=1.1 if ess_frac is None else ess_frac, method=method,
                                    rejuvenate=None if rejuv in (None, "keep") else rejuv, check=False)
                lml_log.append(sharded.get_lml_est(st))
                continue
            ess = sharded.get_ess(st)
            ess_log.append(ess)
            if ess_frac is None or ess < ess_frac * n_global:
                if method == "stratified_sorted":             # the reference's default order of the strata: the replicated plan, phase by phase
                    sharded.pf_resample(st, "stratified", sort_particles=True, check=False)
                else:
                    sharded.pf_resample(st, method, check=False)
                if rejuv and rejuv != "keep":
                    sharded.pf_rejuvenate(st, None, (), 1, method=rejuv)
            sharded.pf_update(st, (t + 1,), (None,), ys[t])
            lml_log.append(sharded.get_lml_est(st))
        b = st.backend
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=b.rows, lw=b.lw, parents=b.parents, gid0=b.gid0,
                 ess=np.array(ess_log), lml=np.array(lml_log))
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
    """one resample + update from a skewed weight vector (some shards own all targets, some none)"""
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    from oracle_shard_backend import OracleShardBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.lgssm2()
        ys = g.models.simulate(model, 3)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, backend_factory=OracleShardBackend)
        b = st.backend
        b.lw[:] = skew_weights(n_global, pattern)[b.gid0:b.gid0 + b.n]
        if method == "stratified_sorted":
            sharded.pf_resample(st, "stratified", sort_particles=True, check=False)
        else:
            sharded.pf_resample(st, method, check=False)
        sharded.pf_update(st, (2,), (None,), ys[1])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=b.rows, lw=b.lw, parents=b.parents, gid0=b.gid0,
                 lml=sharded.get_lml_est(st))
    finally:
        dist.destroy_process_group()


def run_local(rank, world, port, method, n_global, out_dir):
    """island mode: every shard resamples locally (sub-state semantics), no exchange"""
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    from oracle_shard_backend import OracleShardBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.lgssm2(); ys = g.models.simulate(model, 5)
        st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, backend_factory=OracleShardBackend)
        lml = []
        for t in range(1, 5):
            sharded.pf_resample(st, method, check=False, local=True, sort_particles=(t % 2 == 0))
            sharded.pf_update(st, (t + 1,), (None,), ys[t])
            lml.append(sharded.get_lml_est(st))
        b = st.backend
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=b.rows, lw=b.lw, parents=b.parents, gid0=b.gid0, n=b.n,
                 lml=np.array(lml), ess=sharded.get_ess(st))
    finally:
        dist.destroy_process_group()


def run_check(rank, world, port, n_global, out_dir, gpu=False):
    """validity checks of a sharded resample: all weights -Inf (uniform fallback, warning or error), NaN (always an error)"""
    import warnings
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
        if gpu:
            st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, device=0)
            setw = lambda v: setattr(st.local, "log_weights", np.full(st.n_local, v))
            get = lambda: (st.local.traces, st.local.log_weights, st.local.parents)
        else:
            from oracle_shard_backend import OracleShardBackend
            st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, backend_factory=OracleShardBackend)
            def setw(v): st.backend.lw[:] = v
            get = lambda: (st.backend.rows, st.backend.lw, st.backend.parents)
        res = {}
        setw(-np.inf)
        try:
            sharded.pf_resample(st, "multinomial", check=True); res["true_raised"] = False
        except g.ErrorException:
            res["true_raised"] = True
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            sharded.pf_resample(st, "multinomial", check="warn")
        res["warned"] = any("Invalid weights" in str(w.message) for w in wl)
        rows, lw, parents = get()
        if rank == 0:
            setw(float("nan"))                       # NaN on ONE shard only: every shard must see the global flag
        try:
            sharded.pf_resample(st, "multinomial", check="warn"); res["nan_raised"] = False
        except g.ErrorException:
            res["nan_raised"] = True
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=rows, lw=lw, parents=parents, **res)
    finally:
        dist.destroy_process_group()
