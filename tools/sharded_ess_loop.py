"""One-rank sharded ESS-triggered loop (BASELINE configs[3] shape: bearings-only, residual resample + MH when ESS < N/2) through the library
engine -- what the 4-GPU config costs per rank before wire time:  python3 tools/sharded_ess_loop.py [steps] [N] [calls|one_call] [ess fraction]
calls: get_ess + pf_resample + pf_rejuvenate + pf_update per step; one_call: sharded.pf_step_ess (gpf_shard_step_ess).
GPF_SHARD_FORCE_COLLECTIVES=1: a real one-rank RCCL communicator (mailbox path) instead of the one-shard shortcut; GPF_SHARD_GETTERS=scan: the
scan + copy getters."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402
from gpf_amd import sharded  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
model = g.models.bearings4()
ys = g.models.simulate(model, steps + 12)
st = sharded.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=True)
mode = sys.argv[3] if len(sys.argv) > 3 else "calls"
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5       # 0: never resample (the cost of the gate alone), 1.1: every step
n_res = 0


def step(t):
    global n_res
    if mode == "one_call":
        n_res += sharded.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=frac, method="residual", rejuvenate="move", check=False)
        return
    if sharded.get_ess(st) < frac * N:
        n_res += 1
        sharded.pf_resample(st, "residual", check=False)
        sharded.pf_rejuvenate(st, None, (), 1, method="move")
    sharded.pf_update(st, (t + 1,), (None,), ys[t])


for t in range(1, 11):
    step(t)
st.backend.synchronize(); n_res = 0
t0 = time.perf_counter()
for t in range(11, 11 + steps):
    step(t)
st.backend.synchronize()
print(mode, st.backend.summary_mode(), "us/step", round((time.perf_counter() - t0) / steps * 1e6, 2), "resampled", n_res, "of", steps, "log-ML", sharded.get_lml_est(st))
