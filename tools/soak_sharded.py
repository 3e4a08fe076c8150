"""Long sharded run on ONE GPU: `world` ranks on cuda:0 through the library engine over tests/loopback_rccl, every resampler in turn,
against the single-shard CPU oracle bit for bit at the end.  usage: python tools/soak_sharded.py [world] [n_global] [T]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, port, n_global, T, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import gpf_amd as g
    from gpf_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = g.models.lgssm2(); ys = g.models.simulate(model, T)
    st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=77, device=0)
    assert st.backend.lib_comm
    methods = ("stratified", "multinomial", "residual")
    for t in range(1, T):
        sharded.pf_resample(st, methods[t % 3], check=False)
        sharded.pf_update(st, (t + 1,), (None,), ys[t])
    loc = st.local
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=loc.traces, lw=loc.log_weights, parents=loc.parents, lml=sharded.get_lml_est(st))
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n_global = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    tmp = tempfile.mkdtemp()
    lib = os.path.join(tmp, "libloopback_rccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "loopback_rccl", "loopback_rccl.cpp"), "-o", lib], check=True)
    os.environ["GPF_RCCL_LIBRARY"] = lib; os.environ["GPF_SHARD_ENGINE"] = "library"
    mp.spawn(worker, args=(world, 29761, n_global, T, tmp), nprocs=world, join=True)
    import gpf_amd as g
    from oracle import oracle as o
    model = g.models.lgssm2(); ys = g.models.simulate(model, T)
    f = o.OracleFilter(model.model_id, model.params, n_global, 77).initialize(ys[0])
    methods = ("stratified", "multinomial", "residual")
    for t in range(1, T):
        kw = dict(sort_particles=False) if methods[t % 3] == "stratified" else {}
        f.resample(methods[t % 3], check=False, **kw); f.update(ys[t])
    parts = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    ok = (np.array_equal(np.concatenate([p["parents"] for p in parts]), f.parents) and np.array_equal(np.concatenate([p["rows"] for p in parts]), f.rows)
          and np.array_equal(np.concatenate([p["lw"] for p in parts]), f.lw) and all(float(p["lml"]) == f.log_ml_estimate() for p in parts))
    print("soak", world, "ranks", n_global, "particles", T, "steps:", "BIT-IDENTICAL to the single-shard oracle" if ok else "MISMATCH", "log-ML", f.log_ml_estimate())
    sys.exit(0 if ok else 1)
