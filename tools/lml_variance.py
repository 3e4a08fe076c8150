"""BASELINE config 5: stochastic volatility, multinomial resample + move-reweight rejuvenation every step; spread of the
log-ML estimate over seeds on the GPU at full size, and the same seeds at a small size on GPU and CPU oracle (equal bit for bit)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import gpf_amd as g
from oracle import oracle as o

model = g.models.sv1(); T = int(os.environ.get("T", 200)); ys = g.models.simulate(model, T)


def run_gpu(N, seed):
    st = g.pf_initialize(model, (1,), ys[0], N, seed=seed, keep_prev=True)
    for t in range(1, T):
        g.pf_resample(st, "multinomial", check=False); g.pf_rejuvenate(st, g.move_reweight, (), 1, method="reweight")
        g.pf_update(st, (t + 1,), (None,), ys[t])
    v = g.get_lml_est(st); st.close(); return v


def run_cpu(N, seed):
    f = o.OracleFilter(model.model_id, model.params, N, seed, keep_prev=True).initialize(ys[0])
    for t in range(1, T):
        f.resample("multinomial", check=False); f.rejuvenate("reweight", 1); f.update(ys[t])
    return f.log_ml_estimate()


seeds = list(range(1, 17))
big = np.array([run_gpu(2_000_000, s) for s in seeds])
small_gpu = np.array([run_gpu(20_000, s) for s in seeds[:6]])
small_cpu = np.array([run_cpu(20_000, s) for s in seeds[:6]])
print(json.dumps(dict(T=T, N_big=2_000_000, lml_mean=float(big.mean()), lml_std=float(big.std(ddof=1)),
                      N_small=20_000, small_std_gpu=float(small_gpu.std(ddof=1)), small_std_cpu=float(small_cpu.std(ddof=1)),
                      small_gpu_equals_cpu=bool(np.array_equal(small_gpu, small_cpu)))))
