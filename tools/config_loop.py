"""Filter loop of one BASELINE.json config other than the headline (profiling target for rocprofv3):
    python3 tools/config_loop.py config4|config4g|config5|residual [steps] [N]
config4: bearings-only, residual resample when ESS < N/2, then one MH sweep;  config5: stochastic volatility, multinomial resample
+ one move-reweight sweep every step;  residual: LG-SSM, residual resample every step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "config4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
CFG = {"config4": ("bearings4", 1_000_000, "residual", "move", 0.5),
       "config4g": ("bearings4", 1_000_000, "residual", "move", 0.5),      # the same loop through pf_step_ess (one call per step)
       "config5": ("sv1", 2_000_000, "multinomial", "reweight", None),
       "residual": ("lgssm2", 1_000_000, "residual", None, None)}
model_name, N, method, rejuv, ess_frac = CFG[which]
if len(sys.argv) > 3:
    N = int(sys.argv[3])
model = g.models.by_name(model_name)
ys = g.models.simulate(model, steps + 1)
st = g.pf_initialize(model, (1,), ys[0], N, seed=1, keep_prev=rejuv is not None)
n_res = 0
for t in range(1, steps + 1):
    if which == "config4g":
        n_res += g.pf_step_ess(st, (t + 1,), (None,), ys[t], ess_threshold=ess_frac, method=method, rejuvenate=rejuv, check=False)
        continue
    if ess_frac is None or g.get_ess(st) < ess_frac * N:
        n_res += 1
        g.pf_resample(st, method, check=False)
        if rejuv:
            g.pf_rejuvenate(st, None, (), 1, method=rejuv)
    g.pf_update(st, (t + 1,), (None,), ys[t])
st.synchronize()
print("resampled", n_res, "of", steps, "log-ML", g.get_lml_est(st))
