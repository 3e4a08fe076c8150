"""Many small filters: the README loop (README.md:60-79: update, residual resample when ESS < N/2, one MH move) on filters of 100 particles.
  (a) ONE filter through the single-filter API, the ESS read on the host every step (what a reference user's loop does);
  (b) ONE filter, resampling every step without the ESS read (no host round trip);
  (c) B filters in one state: pf_update / pf_resample_blocks(ess_frac = 0.5) / pf_rejuvenate over all blocks, ESS decided on the device.
Prints us per step and filter-steps per second."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gpf_amd as g

N, T = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 300
m = g.models.by_name(sys.argv[2] if len(sys.argv) > 2 else "object_motion"); ys = g.models.simulate(m, T + 12)


def loop_single(ess_read):
    st = g.pf_initialize(m, (1,), ys[0], N, seed=1, keep_prev=True)
    for phase, steps in (("warm", 10), ("timed", T)):
        st.synchronize(); t0 = time.perf_counter()
        for t in range(1, steps + 1):
            g.pf_update(st, (t + 1,), (None,), ys[t])
            if not ess_read or g.get_ess(st) < N / 2:
                g.pf_resample(st, "residual", check=False)
                g.pf_rejuvenate(st, None, (), 1, method="move")
        st.synchronize(); dt = time.perf_counter() - t0
    st.close()
    return dt / T * 1e6


def loop_blocks(B):
    st = g.pf_initialize(m, (1,), ys[0], N * B, seed=1, keep_prev=True)
    for phase, steps in (("warm", 10), ("timed", T)):
        st.synchronize(); t0 = time.perf_counter()
        for t in range(1, steps + 1):
            g.pf_update(st, (t + 1,), (None,), ys[t])
            st._L.gpf_resample_blocks(st._h, 1, N, float("nan"), 1, 0.5, 0, None, None)     # (check = false, no counts asked for: fully asynchronous)
            g.pf_rejuvenate(st, None, (), 1, method="move")
        st.synchronize(); dt = time.perf_counter() - t0
    ess, lml = g.block_stats(st, N)
    st.close()
    return dt / T * 1e6, float(lml.mean()), float(lml.std())


def loop_blocks_own_data(B):
    """every filter on its own observations: per-block initialise / update / rejuvenate (only the blocks that resampled)"""
    import numpy as np
    rng = np.random.default_rng(3)
    base = np.asarray(ys)
    st = g.pf_initialize_blocks(m, (1,), base[0][None, :] + 0.1 * rng.standard_normal((B, base.shape[1])), N * B, N, seed=1, keep_prev=True)
    obs = [base[t][None, :] + 0.1 * rng.standard_normal((B, base.shape[1])) for t in range(1, 12)]
    for phase, steps in (("warm", 10), ("timed", T)):
        st.synchronize(); t0 = time.perf_counter()
        for t in range(1, steps + 1):
            g.pf_update_blocks(st, (t + 1,), (None,), obs[t % 11], N)
            st._L.gpf_resample_blocks(st._h, 1, N, float("nan"), 1, 0.5, 0, None, None)
            st._L.gpf_rejuvenate_blocks(st._h, 0, 1, 1, None)
        st.synchronize(); dt = time.perf_counter() - t0
    st.close()
    return dt / T * 1e6


if not os.environ.get("SMALL_B"):
    a, b = loop_single(True), loop_single(False)
    print(json.dumps({"case": "one filter of 100, ESS read on the host every step", "us_per_step": round(a, 2), "filter_steps_per_s": round(1e6 / a, 1)}))
    print(json.dumps({"case": "one filter of 100, resample + move every step, no host read", "us_per_step": round(b, 2), "filter_steps_per_s": round(1e6 / b, 1)}))
for B in [int(x) for x in os.environ.get("SMALL_B", "1,100,1000,10000,20000").split(",")]:
    us, mean, std = loop_blocks(B)
    print(json.dumps({"case": f"{B} filters of 100 in one state, pf_resample_blocks(ess_frac = 0.5)", "us_per_step": round(us, 2),
                      "filter_steps_per_s": round(B * 1e6 / us, 1), "particle_steps_per_s": round(B * N * 1e6 / us, 1),
                      "log_ml_mean_over_filters": mean, "log_ml_std_over_filters": std}))
for B in [int(x) for x in os.environ.get("SMALL_B", "1000,10000").split(",")]:
    us = loop_blocks_own_data(B)
    print(json.dumps({"case": f"{B} filters of 100, each on its own data (update / resample / rejuvenate block-wise, host hands over {B} observation vectors per step)",
                      "us_per_step": round(us, 2), "filter_steps_per_s": round(B * 1e6 / us, 1)}))
