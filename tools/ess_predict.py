"""VERDICT r05 item 6 (speculative resample chain in gpf_step_ess), costed from a measured trace instead of built.
A speculative chain pays only if the branch is PREDICTED: enqueueing resample -> move -> propagate behind every verdict costs three aborted
launches (~4.1 us each, profiles/r05_step_ess.txt) on the ~79 % of config 4's steps that do not resample.  This script runs BASELINE config 4
(bearings, N = 1e6, ESS < N/2, residual + MH) for T steps, records the ESS the verdict is formed from, and scores host-side predictors of
"this step resamples" that use only what the host already holds (the ESS of the previous steps):
    python3 tools/ess_predict.py [T] [gap_us] [abort_us]
gap_us: the host round trip a correct prediction hides (measured: tools/gpu.sh gaps:config_loop...), abort_us: one aborted launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
GAP = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
ABORT = float(sys.argv[3]) if len(sys.argv) > 3 else 4.1
N = 1_000_000
model = g.models.bearings4(); ys = g.models.simulate(model, T + 2)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=True)
ess, res = [], []
for t in range(1, T + 1):
    e = g.get_ess(st); ess.append(e)
    go = e < 0.5 * N; res.append(go)
    if go:
        g.pf_resample(st, "residual", check=False); g.pf_rejuvenate(st, None, (), 1, method="move")
    g.pf_update(st, (t + 1,), (None,), ys[t])
n_res = sum(res)
print(f"T = {T}, resampling steps {n_res} ({100 * n_res / T:.1f} %)")


def score(name, pred):
    tp = sum(1 for p, r in zip(pred, res) if p and r); fp = sum(1 for p, r in zip(pred, res) if p and not r)
    # a correct "resample" prediction hides the round trip AND the aborted propagate; a false one costs the three aborted launches of the chain
    gain = (tp * (GAP + ABORT) - fp * 3 * ABORT) / T
    print(f"{name:58s} predicted {tp:3d} of {n_res} resampling steps, {fp:3d} false alarms -> {gain:+.2f} us per step")


score("always speculate the resample chain", [True] * T)
score("oracle (knows the verdict: the upper bound)", list(res))
INF = float("inf")
# what the host knows when it enqueues step t: the ESS of the earlier steps SINCE THE LAST RESAMPLE (a step right behind a resample never resamples here)
last = [INF] + [INF if res[t - 1] else ess[t - 1] for t in range(1, T)]
prev = [INF, INF] + [INF if (res[t - 1] or res[t - 2]) else ess[t - 2] for t in range(2, T)]
for margin in (1.0, 1.1, 1.2, 1.35, 1.5):
    score(f"last ESS < {margin:.2f} x threshold", [l < margin * 0.5 * N for l in last])
for margin in (0.9, 1.0, 1.1, 1.2):
    score(f"geometric extrapolation last^2 / prev < {margin:.2f} x threshold", [(l * l / p if p != INF and l != INF and p > 0 else INF) < margin * 0.5 * N for l, p in zip(last, prev)])
