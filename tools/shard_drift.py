"""per-100-step wall time of the sharded loop on one rank (drift / host-bound check)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import gpf_amd as g
from gpf_amd import sharded
use_pg = os.environ.get("USE_PG") == "1"
if use_pg:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = g.models.lgssm2(); ys = g.models.simulate(model, 1300)
st = sharded.pf_initialize(model, (1,), ys[0], 1_000_000, seed=1)
for t in range(1, 20):
    sharded.pf_resample(st, "multinomial", check=False); sharded.pf_update(st, (t,), (None,), ys[t])
st.synchronize()
import gc
if os.environ.get('NOGC') == '1':
    gc.freeze(); gc.disable()
out = []
for blk in range(12):
    t0 = time.perf_counter()
    for t in range(20 + blk * 100, 120 + blk * 100):
        sharded.pf_resample(st, "multinomial", check=False); sharded.pf_update(st, (t,), (None,), ys[t])
    st.synchronize()
    out.append(round((time.perf_counter() - t0) / 100 * 1e6, 1))
print("pg" if use_pg else "nopg", out)
