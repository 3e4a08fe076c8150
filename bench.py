#!/usr/bin/env python3
"""bench.py -- headline benchmark of the particle-filter hot path (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 2-D linear-Gaussian SSM, N = 1e6 particles per GPU, bootstrap
proposal, multinomial resampling EVERY step, T = --steps time steps.  One bench "step" = one
filter time step over all particles = pf_resample!(state, :multinomial) + pf_update!(state, ...).
Metric: particle-steps/sec = (particles on all GPUs) * steps / wall seconds, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the extra objects `roofline` (dominant kernel, HIP-event timed on
the handle's stream) and `cpu_baseline` (the C oracle = a port of the reference algorithm, timed on
this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 1_000_000
AUX_STEPS = 30             # steps of the stand-alone gather leg
CPU_STEPS = 80             # steps of the cpu_baseline sample: ~13 s on one host core at N = 1e6 (a bounded sample of the same workload)
SEED = 1
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def kernel_sources_sha16() -> str:
    """fingerprint of the kernel sources of this build (tools/pmc_to_json.py stores the same when it records PMC traffic)"""
    import hashlib
    root = os.path.join(ROOT, "genparticlefilters.jl_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hpp", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(d: int, W: int):
    """Algorithmic bytes per particle and launch: every input read once,
    every output written once.  d = state columns, rows are W doubles."""
    row = 8 * W
    return {
        "k_step": 4 + row + row + 8,       # fused gather form, the one the hot loop launches: R ancestor, R row (random),
                                           # W row, W lw (incoming weights are 0 after a resample: not read)   (16d + 12)
        "k_max_partial": 8,                # R lw
        "k_scan": 8 + 8,                   # R lw, W cdf
        "k_search": 8 + 4,                 # R cdf cell, W ancestor (k_search_multi reads 2-byte offsets instead: fewer bytes, same figure kept)
        "k_gather": 4 + row + row + 8,     # R ancestor, R row, W row, W lw  (16d + 12)
    }


def _free_port() -> int:
    """a rendezvous port for the self-launched job: a random bindable port BELOW the kernel's ephemeral range (a bind(0) port can be handed to an outgoing
    connection before rank 0 listens on it)"""
    import random
    import socket
    rng = random.SystemRandom()
    for _ in range(200):
        p = rng.randrange(20000, 32000)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", p))
            except OSError:
                continue
            return p
    raise RuntimeError("no free rendezvous port between 20000 and 32000")


def launch_ranks(n: int, argv, timeout_s: float) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (how the driver starts a scaling run):
    start the N ranks as a CHILD job -- this process has not imported torch and never touches a GPU, so nothing that has
    initialised HIP is replaced -- relay its output, print the job's JSON line as the LAST stdout line and return the
    job's exit code.  Any rank failing makes torch.distributed.run (and so this process) exit non-zero.  The child tree is
    killed and the exit code is non-zero if no JSON line arrives within `timeout_s`."""
    import signal
    import subprocess
    import threading
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT, start_new_session=True)
    lines = []

    def pump():
        for ln in child.stdout:
            lines.append(ln.rstrip("\n"))
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    timed_out = False
    try:
        rc = child.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(child.pid, signal.SIGKILL)               # the exact process group this function started
        except ProcessLookupError:
            pass
        child.wait()
        rc = 124
    th.join(timeout=10)
    js = [ln for ln in lines if ln.startswith("{") and '"metric"' in ln]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)                         # RCCL banners etc.: kept, but off the JSON channel
    if timed_out:
        print(f"[bench] no result after {timeout_s:.0f} s: killed the {n}-rank job", file=sys.stderr, flush=True)
        return rc
    if rc == 0 and not js:
        print("[bench] the rank job exited 0 without a JSON line", file=sys.stderr, flush=True)
        return 1
    if js:
        sys.stdout.write(js[-1] + "\n")
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline loop (no stand-alone gather leg, no named variants, no CPU baseline): what the PMC passes profile, so that "
                         "a kernel's mean counters are not mixed with the variants' launches of the same kernel")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("GPF_BENCH_LAUNCH_TIMEOUT", "1500")),
                    help="wall-clock bound (s) of the self-launched multi-rank job")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # BEFORE torch is imported or any GPU call is made
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))

    import numpy as np
    import torch
    import gpf_amd as g

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or run plain `python bench.py --gpus N`)")
    # functional check of the multi-rank code path on a 1-GPU box: every rank on cuda:0, collectives staged through gloo
    one_device = os.environ.get("GPF_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    force_sharded = os.environ.get("GPF_BENCH_FORCE_SHARDED") == "1"       # exercise the sharded path at world 1
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    model = g.models.lgssm2()
    K, Wm = args.steps, args.warmup
    # every leg below indexes observations by its own loop counter: never fewer rows than any leg touches
    # (the gather and cpu_baseline legs run a fixed 30 steps whatever --steps says)
    n_obs = max(K + Wm + 6, AUX_STEPS + 2, CPU_STEPS + 2)       # (+4: the guarded first steps of the sharded engine)
    ys = g.models.simulate(model, n_obs)
    n_local = args.particles_per_gpu
    n_global = n_local * world

    sharded_mode = world > 1 or force_sharded
    if not sharded_mode:
        state = g.pf_initialize(model, (1,), ys[0], n_local, seed=SEED, device=local_rank)
        t_first = 1

        def step(t):
            g.pf_resample(state, "multinomial")                   # the reference's defaults: priority_fn = nothing, check = :warn
            g.pf_update(state, (t + 1,), (None,), ys[t])
    else:
        from gpf_amd import sharded

        # The library engine (gpf_shard_resample on libgpf's own RCCL communicator) has only ever met one-rank communicators
        # in the build environment.  Its first collective steps run under a guard: if any rank fails to create the
        # communicator or to run them, ALL ranks fall back to the phase-by-phase engine over torch.distributed.
        engine_note = None

        GUARD_STEPS = (("multinomial",), ("multinomial",), ("stratified",), ("multinomial_sorted",))

        def make_state():
            st = sharded.pf_initialize(model, (1,), ys[0], n_global, seed=SEED, device=local_rank)
            for tp, (meth,) in enumerate(GUARD_STEPS, start=1):     # four steps before the counted warm-up: the headline's resampler twice, then the two
                sharded.pf_resample(st, meth)                        # resamplers whose slabs travel through the receive windows (never yet over xGMI)
                sharded.pf_update(st, (tp + 1,), (None,), ys[tp])
            st.synchronize()
            return st

        def attempt():
            """one engine configuration under the guard: the state after the guard steps, or why not.  A transport that delivers WRONG data without
            failing (mailboxes and windows have never crossed xGMI) would show in the global estimate: it must be finite, within reach of the exact
            value, and the same on every rank; the verdict is all-reduced, so all ranks move on together"""
            ok, err, st, lml_g = 1, "", None, float("nan")
            try:
                st = make_state()
                lml_g = sharded.get_lml_est(st)
                exact_g = g.models.kalman_loglik(model, ys[:len(GUARD_STEPS) + 1])
                if not (lml_g == lml_g and abs(lml_g - exact_g) < 1.0):
                    ok, err = 0, f"log-ML after the guarded steps {lml_g!r} against the exact {exact_g!r}"
            except Exception as e:                                   # noqa: BLE001 -- any failure means: the next configuration
                ok, err = 0, repr(e)
            if dist is not None and world > 1:
                dev_ = "cpu" if one_device else "cuda"
                flag = torch.tensor([ok], dtype=torch.int32, device=dev_)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok_all = int(flag.item())
                if ok_all:                                           # every rank computed the GLOBAL estimate: bit-identical or the configuration is out
                    lo = torch.tensor([lml_g], dtype=torch.float64, device=dev_); hi = lo.clone()
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                    if float(lo.item()) != float(hi.item()):
                        ok_all, err = 0, f"ranks disagree on the global log-ML after the guarded steps ({float(lo.item())!r} .. {float(hi.item())!r})"
            else:
                ok_all = ok
            return ok_all, (err or "another rank failed"), st

        if os.environ.get("GPF_BENCH_TRY_LIBRARY") == "1" and "GPF_SHARD_ENGINE" not in os.environ:
            os.environ["GPF_SHARD_ENGINE"] = "library"               # tests: start like a multi-GPU box does, whatever the process group
            tried_library = True
        else:
            tried_library = False
        ok_all, err, state = attempt()
        if not ok_all and os.environ.get("GPF_SHARD_ENGINE") != "python" and os.environ.get("GPF_SHARD_EXCHANGE") != "rccl":
            # the library engine as round 5 left it: slabs through grouped ncclSend / ncclRecv instead of the receive windows, the (max, flags) mailbox
            # round in its own small launch instead of inside its consumer's (the default for ranks with a device each since round 6)
            engine_note = f"receive windows and the fused (max, flags) round off ({err})"
            print(f"[bench rank {rank}] {engine_note}", file=sys.stderr)
            os.environ["GPF_SHARD_EXCHANGE"] = "rccl"; os.environ["GPF_SHARD_FUSE_MF"] = "0"
            ok_all, err, state = attempt()
        if not ok_all:
            if os.environ.get("GPF_SHARD_ENGINE") == "python" and not tried_library:
                raise SystemExit(f"sharded engine failed: {err}")
            engine_note = f"fell back from the library engine ({err})"
            print(f"[bench rank {rank}] {engine_note}", file=sys.stderr)
            os.environ["GPF_SHARD_ENGINE"] = "python"
            state = make_state()
        t_first = len(GUARD_STEPS) + 1

        def step(t):
            sharded.pf_resample(state, "multinomial")                # the reference's defaults as in the one-GPU loop above: priority_fn = nothing, check = :warn
            sharded.pf_update(state, (t + 1,), (None,), ys[t])

    def barrier():
        state.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the host loop is Python: a generation-2 garbage collection (tens of ms, once per ~1000 sharded steps) would be
    # charged to the filter.  Like timeit, keep the collector out of the timed region (objects made so far are frozen).
    # Done BEFORE the warm-up: the collection itself idles the GPU for tens of ms, long enough for its clocks to drop -- with
    # it between warm-up and timed loop a 20-step run paid ~100 us of ramp-up (5 us per step).
    gc.collect(); gc.freeze(); gc.disable()
    t = t_first
    for _ in range(Wm):
        step(t); t += 1
    barrier()
    headline_traffic = None
    if sharded_mode and getattr(state.backend, "lib_comm", False):
        state.backend.traffic(reset=True)
    t0 = time.perf_counter()
    for _ in range(K):
        step(t); t += 1
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if sharded_mode and getattr(state.backend, "lib_comm", False):
        headline_traffic = state.backend.traffic(reset=True)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = n_global * K / elapsed
    lml = sharded.get_lml_est(state) if sharded_mode else g.get_lml_est(state)
    # BASELINE.json's metric also names the log-ML error: the model is linear-Gaussian, so the exact log p(y_1:T) of the
    # observations consumed so far is a Kalman recursion away (host, NumPy)
    lml_exact = g.models.kalman_loglik(model, ys[:t]) if rank == 0 else None

    # ---- roofline of the dominant kernel: HIP events around every launch, on the handle's stream ----
    roofline = None
    local = state.local if sharded_mode else state
    kid_names = g._lib.KERNEL_NAMES
    kids = [g._lib.K_STEP, g._lib.K_MAX, g._lib.K_SCAN, g._lib.K_SEARCH, g._lib.K_GATHER]
    n_ev = min(K, 200)
    for kid in kids:
        local.kernel_timing(kid, True)
    tt_ = 1
    for _ in range(n_ev):
        step(tt_); tt_ += 1
    per = {}
    for kid in kids:
        ms, cnt = local.kernel_time(kid)
        local.kernel_timing(kid, False)
        if cnt:
            per[kid_names[kid]] = (ms / cnt * 1e3, cnt)          # us per launch
    if sharded_mode:
        # the sharded resample times its two kernels under the search / gather ids: k_push_scan (every shard evaluates the
        # targets of ALL output slots and stages its hits: W 16 B per hit) and k_push (R staged hit, R cdf cell, R row, W packed row)
        per = {{"k_search": "k_push_scan", "k_gather": "k_push"}.get(k, k): v for k, v in per.items()}
    if per and rank == 0:
        ab = algorithmic_bytes(model.dim, local.row_width)
        ab["k_push_scan"] = 16
        ab["k_push"] = 16 + 8 + 8 * local.row_width + 8 * local.row_width + 8
        share = {k: v[0] * v[1] for k, v in per.items()}
        dom = max(share, key=share.get)
        us = per[dom][0]
        achieved = ab[dom] * n_local / (us * 1e-6) / 1e9
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json; FETCH_SIZE +
        # WRITE_SIZE collected in separate runs and corrected as MI355X_MICROARCH.md §HBM prescribes); only valid for
        # the workload they were measured on
        traffic, traffic_source = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            sha = kernel_sources_sha16()
            if n_local != N_PER_GPU or dom not in pmc["kernels"]:
                traffic_source = "none: no PMC pass for this kernel / size"
            elif pmc.get("kernel_sources_sha16") != sha:
                # the counters were collected on other kernel sources than the ones being timed: do not quote them
                traffic_source = (f"stale: profiles/pmc_traffic.json (tag {pmc.get('tag')}) was collected on kernel sources "
                                  f"{pmc.get('kernel_sources_sha16')}, this build is {sha}")
            else:
                traffic = pmc["kernels"][dom]["traffic_bytes"]
                traffic_source = {"file": "profiles/pmc_traffic.json", "tag": pmc.get("tag"), "kernel_sources_sha16": sha,
                                  "passes": f"profiles/{pmc.get('tag')}_pmc_FETCH_SIZE.csv, profiles/{pmc.get('tag')}_pmc_WRITE_SIZE.csv"}
        except Exception as e:                                   # noqa: BLE001
            traffic, traffic_source = None, f"none: {e!r}"
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": ab[dom] * n_local, "avg_launch_us": round(us, 2),
                    "all_kernels_us": {k: round(v[0], 2) for k, v in per.items()}}

    # ---- the stand-alone resample gather (BASELINE.json names it): the hot loop above fuses the gather into the next
    #      propagate, so time k_gather itself on the same state by asking for the ESS between resample and update
    gather = None
    if not sharded_mode and rank == 0 and not args.headline_only:
        gather = {}
        W = state.row_width
        gbytes = (4 + 8 * W + 8 * W + 8) * n_local                # R anc, R row (random), W row, W lw = 16d + 12
        for meth, kw in (("multinomial", {}), ("multinomial_sorted", {}), ("stratified", {"sort_particles": False})):
            state.kernel_timing(g._lib.K_GATHER, True)
            for i in range(AUX_STEPS):
                g.pf_resample(state, meth, check=False, **kw)
                g.get_ess(state)                                   # forces materialize() = the stand-alone k_gather
                g.pf_update(state, (i + 2,), (None,), ys[1 + i])
            ms, cnt = state.kernel_time(g._lib.K_GATHER)
            state.kernel_timing(g._lib.K_GATHER, False)
            us = ms / max(cnt, 1) * 1e3
            gather[meth] = {"avg_launch_us": round(us, 2), "achieved": round(gbytes / us / 1e3, 1), "unit": "GB/s",
                            "frac": round(gbytes / us / 1e3 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": gbytes}

    # ---- sharded runs: variants of the same filter reported beside the headline workload, same timing protocol
    class Lockstep:
        """The secondary legs below run the same collective sequence on every rank.  A leg that fails on ONE rank (a transport that has never met real
        hardware) must not leave the others waiting in a barrier it never reaches: steps run through `step`, which remembers the first failure and turns the
        rest of the leg into no-ops; the barriers are still met; `agree` (an all-reduce) makes every rank raise together, so the leg is reported as failed
        on the line and the next one starts in step."""
        def __init__(self):
            self.err = None

        def step(self, fn, *a):
            if self.err is None:
                try:
                    return fn(*a)
                except Exception as e:                               # noqa: BLE001
                    self.err = e
            return None

        def barrier(self):
            try:
                state.synchronize()
            except Exception as e:                                   # noqa: BLE001 -- (a flagged device wait surfaces here)
                self.err = self.err or e
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        def agree(self):
            ok = 1 if self.err is None else 0
            if dist is not None:
                fl = torch.tensor([ok], dtype=torch.int32, device="cpu" if one_device else "cuda")
                dist.all_reduce(fl, op=dist.ReduceOp.MIN)
                ok = int(fl.item())
            if not ok:
                raise RuntimeError(repr(self.err) if self.err is not None else "failed on another rank")

    def variant(step_fn, k):
        """k steps of step_fn after 5 untimed ones, barriers on both sides, MAX over ranks -> seconds"""
        ls = Lockstep()
        for i in range(5):
            ls.step(step_fn, i)
        gc.collect(); gc.disable()
        ls.barrier()
        v0 = time.perf_counter()
        for i in range(k):
            ls.step(step_fn, i)
        ls.barrier()
        ve = time.perf_counter() - v0
        gc.enable()
        if dist is not None:
            tv = torch.tensor([ve], dtype=torch.float64, device="cpu" if one_device else "cuda")
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
            ve = float(tv.item())
        ls.agree()
        return ve

    def variant_line(workload, k, seconds):
        return {"workload": workload, "value": round(n_global * k / seconds, 1), "unit": "particle-steps/sec", "steps": k,
                "ms_per_step": round(seconds / k * 1e3, 5)}

    strat = island = plans = sorted_variant = strat_sorted = links = exchange_modes = calibration = headline_phases = None
    if not sharded_mode and not args.headline_only:
        # the OPT-IN sorted form of the multinomial resampler (gpf.h GPF_RESAMPLE_MULTINOMIAL_SORTED): same offspring-count
        # law, ancestors in non-decreasing order -- NOT the reference's slot order, so a named variant beside the unchanged headline
        kv = min(K, 200)

        def sorted_step(tq):
            g.pf_resample(state, "multinomial_sorted")
            g.pf_update(state, (tq + 1,), (None,), ys[1 + tq % (n_obs - 1)])
        sorted_variant = variant_line("same filter, opt-in multinomial_sorted resample every step (sorted uniforms: monotone ancestors)",
                                      kv, variant(sorted_step, kv))
        for kid in kids:                                           # per-kernel HIP-event times of the variant, in a loop of their own
            local.kernel_timing(kid, True)
        for i in range(min(kv, 50)):
            sorted_step(i)
        sk = {}
        for kid in kids:
            ms, cnt = local.kernel_time(kid)
            local.kernel_timing(kid, False)
            if cnt:
                sk[kid_names[kid]] = round(ms / cnt * 1e3, 2)
        sorted_variant["all_kernels_us"] = sk
        if "k_step" in sk and rank == 0:
            # the fused gather + propagate of this variant against the same algorithmic bytes as the headline's dominant kernel
            abk = algorithmic_bytes(model.dim, local.row_width)["k_step"]
            sorted_variant["k_step_roofline"] = {"achieved": round(abk * n_local / (sk["k_step"] * 1e-6) / 1e9, 1), "unit": "GB/s",
                                                 "frac": round(abk * n_local / (sk["k_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "algorithmic_bytes_per_launch": abk * n_local}
        # :stratified with the reference's DEFAULT sort_particles=true (src/resample.jl:145,156-157): sortperm of the weights in front of
        # every resample (key pass, one equal-count partition pass, one workgroup per bucket in LDS: DESIGN.md 4.5)
        def strat_sorted_step(tq):
            g.pf_resample(state, "stratified", sort_particles=True, check=False)
            g.pf_update(state, (tq + 1,), (None,), ys[1 + tq % (n_obs - 1)])
        strat_sorted = variant_line("same filter, stratified resample every step with the reference's default sort_particles=true",
                                    kv, variant(strat_sorted_step, kv))
    if sharded_mode:
        kv = min(K, 200)

        def step_of(method, **kw):
            def f(tq):
                sharded.pf_resample(state, method, **kw)                 # (check = :warn like the headline loop)
                sharded.pf_update(state, (tq + 1,), (None,), ys[1 + tq % (n_obs - 1)])
            return f
        lib_engine = getattr(state.backend, "lib_comm", False)
        links = {}

        def phases_of(step_fn, k=30):
            """`phases_us`: where a step of this variant spends its time on THIS rank (rank 0 prints its own), microseconds per step from events at
            the phase boundaries on the handle's stream + one host timer around the wait for the exchange's split sizes (gpf.h gpf_phase_times).
            A pass of its own behind the timed loop: the marks cost an event each."""
            if not lib_engine:
                return None
            ls = Lockstep()
            for i in range(3):
                ls.step(step_fn, i)
            ls.barrier()
            ls.step(state.backend.phase_timing, True)
            for i in range(k):
                ls.step(step_fn, i)
            ph = ls.step(state.backend.phase_times)
            ls.step(state.backend.phase_timing, False)
            ls.barrier()
            ls.agree()
            return ph
        try:
            headline_phases = phases_of(step_of("multinomial"))
        except Exception as e:                                       # noqa: BLE001
            headline_phases = {"error": repr(e)}
        if headline_traffic is not None:                             # the timed headline loop itself (i.i.d. multinomial, the plan named in exchange_plans.timed)
            calls, sent, recv, eb = headline_traffic
            peers = max(world - 1, 1)
            links["multinomial_headline"] = {"entry_bytes": eb, "observed_entries_out_per_step": round(sent / max(calls, 1), 1),
                                             "observed_bytes_per_link_per_step": round(sent * eb / max(calls, 1) / peers, 1),
                                             "predicted_entries_out_per_step": round(n_local * (world - 1) / world, 1),
                                             "predicted_bytes_per_link_per_step": round(n_local / world * eb, 1),
                                             "prediction": "i.i.d. ancestors: (G-1)/G of a shard's rows leave it every step, n / G entries per link (DESIGN.md 6.7)",
                                             "rank": rank}

        def link_bytes(name, run, predicted_entries_out, note):
            """bytes this rank put on each of its G - 1 links per step (gpf_comm_traffic: what the library's exchange really sent) beside the
            scaling worksheet's prediction"""
            if not lib_engine:
                return run()
            state.backend.traffic(reset=True)
            out = run()
            calls, sent, recv, eb = state.backend.traffic(reset=True)
            peers = max(world - 1, 1)
            links[name] = {"entry_bytes": eb, "observed_entries_out_per_step": round(sent / max(calls, 1), 1),
                           "observed_bytes_per_link_per_step": round(sent * eb / max(calls, 1) / peers, 1),
                           "predicted_entries_out_per_step": predicted_entries_out,
                           "predicted_bytes_per_link_per_step": (None if predicted_entries_out is None else round(predicted_entries_out * eb / peers, 1)),
                           "prediction": note, "rank": rank}
            return out
        # STRATIFIED resampling (BASELINE.json configs[2]: monotone targets, almost no row leaves its shard)
        try:
            strat = variant_line("same filter, stratified resample every step, sort_particles=false (BASELINE.json configs[2])",
                                 kv, link_bytes("stratified", lambda: variant(step_of("stratified"), kv), None,
                                                "boundary slabs: ~ cv sqrt(h n) slots per shard boundary (DESIGN.md 6.9: 2-4e3 at n = 1e6), not (G-1)/G of the rows"))
            strat["phases_us"] = phases_of(step_of("stratified"))
        except Exception as e:                                       # noqa: BLE001 -- a variant must not take the headline line down
            strat = {"error": repr(e)}
        # how the slabs travel (gpf.h gpf_comm_set_exchange): peer stores into the destination ranks' receive windows ("p2p": no host wait, no
        # ncclGroup) against packed entries through grouped ncclSend / ncclRecv ("rccl").  The line above ran the mode named in "timed"; both are timed here
        if lib_engine:
            timed_mode = state.backend.exchange()
            strat["exchange"] = timed_mode
            exchange_modes = {"timed": timed_mode}
            for md in ("p2p", "rccl"):
                try:
                    state.backend.set_exchange(md)
                except Exception as e:                               # noqa: BLE001 -- no windows on this communicator: the line says so
                    exchange_modes[md] = {"unavailable": str(e)}
                    continue
                exchange_modes[md] = {}
                for meth in ("stratified", "multinomial_sorted"):
                    try:
                        ln = variant_line(f"{meth} resample every step, slabs through {md}", kv, variant(step_of(meth), kv))
                        ln["phases_us"] = phases_of(step_of(meth))
                    except Exception as e:                           # noqa: BLE001
                        ln = {"error": repr(e)}
                    exchange_modes[md][meth] = ln
            try:
                state.backend.set_exchange(timed_mode)
            except Exception:                                        # noqa: BLE001
                pass
        # the opt-in sorted form of the HEADLINE's resampler across shards: the same offspring-count law as :multinomial,
        # ascending targets -> every shard serves one slot range, the exchange is boundary slabs like the stratified one
        try:
            sorted_variant = variant_line("same filter, opt-in multinomial_sorted resample every step across the shards (sorted uniforms: one served slot range per shard)",
                                          kv, link_bytes("multinomial_sorted", lambda: variant(step_of("multinomial_sorted"), kv), None,
                                                         "boundary slabs: the sorted uniforms' spread ~ sqrt(N) slots per shard boundary plus the shards' weight imbalance"))
            sorted_variant["phases_us"] = phases_of(step_of("multinomial_sorted"))
        except Exception as e:                                       # noqa: BLE001 -- a variant must not take the headline line down
            sorted_variant = {"error": repr(e)}
        # the communication-free "island" mode (every shard resamples locally with the reference's sub-state semantics,
        # SURVEY.md 8e): a different estimator, reported for comparison only
        try:
            island = variant_line("same filter, every shard resamples its own particles (multinomial, sub-state semantics), no exchange",
                                  kv, variant(step_of("multinomial", local=True), kv))
        except Exception as e:                                       # noqa: BLE001
            island = {"error": repr(e)}
        # the two exchange plans of the i.i.d. resamplers (gpf.h gpf_comm_set_plan): the headline ran the plan named in "timed"; both
        # are timed over a FIXED 100 steps whatever --steps says (the driver's scaling command runs 20), so that the first multi-GPU run can decide
        if lib_engine:
            timed_plan = state.backend.plan()
            plans = {"timed": timed_plan}
            kp = 100
            for pl in ("push", "pull"):
                try:
                    state.backend.set_plan(pl)
                    plans[pl] = variant_line(f"headline workload, exchange plan {pl}", kp,
                                             link_bytes(f"multinomial_{pl}", lambda: variant(step_of("multinomial"), kp), round(n_local * (world - 1) / world, 1),
                                                        "i.i.d. ancestors: (G-1)/G of a shard's rows leave it every step, n / G entries per link"))
                    plans[pl]["phases_us"] = phases_of(step_of("multinomial"))
                except Exception as e:                               # noqa: BLE001
                    plans[pl] = {"error": repr(e)}
            # ... and the push plan with its rows through the receive windows (gpf.h GPF_SHARD_EXCHANGE_P2P_ALL): the same look-ups, no host wait for the
            # split sizes, no ncclGroup -- scattered 8 (W + 2)-byte peer stores against RCCL's bulk copies on a bandwidth-bound exchange
            try:
                if one_device:
                    # ranks SHARING a GPU (functional checks of this command line on a 1-GPU box): a propagate whose lanes wait on the window holds the
                    # registers the peer's 1024-thread look-up kernel needs on the same device -- with half of all slots waiting the peers starve each
                    # other for seconds per step.  Ranks with a GPU each wait on their own device for a kernel on a peer's.
                    raise RuntimeError("skipped: the ranks of this run share one GPU")
                mode0 = state.backend.exchange()
                state.backend.set_plan("push"); state.backend.set_exchange("p2p_all")
                plans["push_windows"] = variant_line("headline workload, push plan, rows through the receive windows (p2p_all)", kp,
                                                     link_bytes("multinomial_push_windows", lambda: variant(step_of("multinomial"), kp), round(n_local * (world - 1) / world, 1),
                                                                "i.i.d. ancestors: (G-1)/G of a shard's rows leave it every step, as 8 (W + 2)-byte window entries"))
                plans["push_windows"]["phases_us"] = phases_of(step_of("multinomial"))
                state.backend.set_exchange(mode0)
            except Exception as e:                                   # noqa: BLE001
                plans["push_windows"] = {"error": repr(e)}
            try:
                state.backend.set_plan(timed_plan)
            except Exception:                                        # noqa: BLE001
                pass
            # what the transports cost on this machine (gpf.h gpf_comm_calibrate): the headline's exchange shape -- n / G packed entries to and from every
            # peer, grouped ncclSend / ncclRecv -- as a measured per-link rate (the scaling worksheet of DESIGN.md 6.7 ASSUMES 76 GB/s), and one mailbox round
            try:
                ls = Lockstep()
                calibration = ls.step(state.backend.calibrate, max(n_local // world, 1), 20)
                slab = ls.step(state.backend.calibrate, 4096, 20)     # ... and a boundary slab's size: the group's latency floor
                ls.agree()
                calibration["slab_exchange"] = slab
            except Exception as e:                                   # noqa: BLE001
                calibration = {"error": repr(e)}

    # ---- CPU baseline: the oracle (port of the reference algorithm), bounded sample, rank 0 / N=1 only ----
    cpu = cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.headline_only:
        from oracle import oracle as o          # cpu_baseline leg: the only place bench.py touches oracle/
        o.lib()
        n_cpu, k_cpu = n_local, CPU_STEPS
        orc = o.OracleFilter(model.model_id, model.params, n_cpu, SEED).initialize(ys[0])
        c0 = time.perf_counter()
        for s in range(1, k_cpu + 1):
            orc.resample("multinomial", check=False)
            orc.update(ys[s])
        ce = time.perf_counter() - c0
        cpu = {"value": round(n_cpu * k_cpu / ce, 1), "unit": "particle-steps/sec", "cores": 1, "kind": "port",
               "sample": f"same workload, N={n_cpu}, first {k_cpu} steps, single-thread C oracle ({ce:.1f} s); "
                         "reference (Julia) not runnable on this box"}
        # the same sample multi-threaded (OpenMP over particles; the prefix sum stays sequential).  16 threads: with one
        # fork-join per primitive and first-touch NumPy buffers, more threads are slower on the 256-thread GPU-box host
        ncores = min(os.cpu_count() or 1, 16)
        used = o.set_threads(ncores)
        orc = o.OracleFilter(model.model_id, model.params, n_cpu, SEED).initialize(ys[0])
        c0 = time.perf_counter()
        for s in range(1, k_cpu + 1):
            orc.resample("multinomial", check=False)
            orc.update(ys[s])
        ce = time.perf_counter() - c0
        o.set_threads(1)
        cpu_all = {"value": round(n_cpu * k_cpu / ce, 1), "unit": "particle-steps/sec", "cores": used, "kind": "port",
                   "sample": f"same sample, OpenMP over particles on {used} threads ({ce:.1f} s)"}

    # world size of the RCCL communicator the timed resamples actually ran their exchange on (0: no RCCL in the data path --
    # one unsharded GPU, or the torch.distributed engine over gloo)
    if not sharded_mode:
        rccl_ranks = 0
    elif getattr(state.backend, "lib_comm", False):
        rccl_ranks = int(state.backend.comm_world()) if os.environ.get("GPF_RCCL_LIBRARY") is None else 0
    else:
        rccl_ranks = world if (dist is not None and dist.is_initialized() and dist.get_backend() == "nccl") else 0
    if rank == 0:
        out = {
            "metric": "particle-steps/sec", "value": round(value, 1), "unit": "particle-steps/sec",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D linear-Gaussian SSM, bootstrap PF, multinomial resample every step "
                                   "(BASELINE.json configs[1])",
                       "particles_per_gpu": n_local, "particles_total": n_global, "T": K,
                       "state_dim": model.dim, "parallelism": f"particle-shard x{world}"},
            "shard_engine": (None if not sharded_mode else
                             (f"library: gpf_shard_resample on libgpf's own RCCL communicator of {world} rank(s)" + (f"; {engine_note}" if engine_note else "")
                              if getattr(state.backend, "lib_comm", False) else
                              f"python: sharded.py composes the phases over torch.distributed ({dist.get_backend() if dist is not None and dist.is_initialized() else 'no'} backend, {world} rank(s))"
                              + (f"; {engine_note}" if engine_note else ""))),
            "log_ml_estimate": lml, "log_ml_exact_kalman": lml_exact, "log_ml_abs_error": abs(lml - lml_exact),
            "rccl_ranks": rccl_ranks,
            "shard_summaries": (state.backend.summary_mode() if sharded_mode and hasattr(state.backend, "summary_mode") else None),
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_multithread": cpu_all, "resample_gather_kernel": gather,
            "stratified_variant": strat, "local_resample_variant": island, "exchange_plans": plans,
            "multinomial_sorted_variant": sorted_variant, "stratified_sort_particles_variant": strat_sorted,
            "exchange_bytes_per_link": links,
            # N > 1, library engine: where a step's time goes (rank 0's own phases), both slab transports, what the links and mailboxes measure here
            "phases_us": headline_phases, "slab_exchange_modes": exchange_modes, "transport_calibration": calibration,
        }
    else:
        out = None

    def flush_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    # The JSON line must be the LAST thing on the job's stdout.  RCCL prints a banner ("RCCL version ...", "Librccl path ...")
    # through C stdio, buffered until exit when stdout is a pipe: every rank flushes it now, all ranks meet, and only then
    # does rank 0 print.
    if dist is not None:
        flush_c_stdio(); sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        flush_c_stdio()
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
