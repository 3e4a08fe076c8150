"""CPU oracle for the particle-filter hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``genparticlefilters.jl_amd``) never does: it has no CPU fallback.

``OracleFilter`` composes the C primitives of ``gpf_oracle.c`` into the reference's operations;
every statement cites the line of GenParticleFilters.jl v0.2.3 (paths relative to
``/root/reference``) it restates.  PARITY STATUS: random streams are *unpinned* against the
reference (Julia is absent and the reference's tests hold no seeds / golden vectors, SURVEY.md
§8c); deterministic arithmetic is pinned by ``tests/test_oracle_*.py``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

MODEL_LGSSM2, MODEL_BEARINGS4, MODEL_SV1, MODEL_OBJECT_MOTION, MODEL_LINE = 1, 2, 3, 4, 5
FLAG_NAN, FLAG_POSINF, FLAG_ALL_NEGINF = 1, 2, 4
METHODS = ("multinomial", "residual", "stratified", "multinomial_sorted")


def build(force: bool = False) -> str:
    """Compile oracle/_build/liboracle.so with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("gpf_oracle.c", "ref_literal.c", "gpf_oracle_math.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    def sig(name, res, *args):
        f = getattr(L, name); f.restype = res; f.argtypes = list(args)
    i32, i64, u32, u64, f64 = C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    pf64, pu64, pi32 = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int)
    sig("o_log_d", f64, f64); sig("o_exp_d", f64, f64); sig("o_exp_fix_d", u64, f64, i32)
    sig("o_atan2_d", f64, f64, f64); sig("o_sincos2pi_d", None, f64, pf64, pf64)
    sig("o_philox_d", None, u32, u32, u32, u32, u32, u32, _u32p)
    sig("o_normal2_d", None, u64, u32, u32, u32, u32, pf64, pf64)
    sig("o_u52_d", f64, u64, u32, u32, u32, u32)
    sig("o_resample_u52_d", f64, u64, u32, u32)
    sig("o_loglik_rows", None, C.c_int, _f64p, _f64p, C.c_int, i64, _f64p, _f64p)
    sig("o_math_vec", None, i32, _f64p, _f64p, i64, _f64p, _f64p)
    sig("o_fix_K", i32, i64)
    sig("o_lse_from", f64, f64, u64, i32, i32); sig("o_ess_from", f64, u64, u64, u64)
    sig("o_model_dim", i32, i32); sig("o_model_nblk", i32, i32)
    sig("o_init", None, i32, _f64p, u64, u32, i64, i64, i32, _f64p, _f64p, _f64p)
    sig("o_step", None, i32, _f64p, u64, u32, i64, i64, i32, i32, _f64p, _f64p, _f64p, _f64p)
    sig("o_init_proposal", None, i32, _f64p, u64, u32, i64, i64, i32, _f64p, _f64p, _f64p)
    sig("o_step_proposal", None, i32, _f64p, u64, u32, i64, i64, i32, i32, _f64p, _f64p, _f64p, _f64p)
    sig("o_init_strata", None, i32, _f64p, u64, u32, i64, i64, i32, _f64p, _f64p, i32, i32, f64, _f64p, _f64p)
    sig("o_init_strata_proposal", None, i32, _f64p, u64, u32, i64, i64, i32, _f64p, _f64p, i32, i32, f64, _f64p, _f64p)
    sig("o_step_strata", None, i32, _f64p, u64, u32, i64, i64, i32, i32, _f64p, _f64p, i32, i32, f64, _f64p, _f64p, _f64p)
    sig("o_move", u64, i32, _f64p, u64, u32, i64, i64, i32, i32, _f64p, i32, i32, _f64p, _f64p, _f64p)
    sig("o_move_proposal", None, i32, _f64p, _f64p, u64, u32, i64, i64, i32, i32, _f64p, i32, _f64p, _f64p, _f64p)
    sig("o_move_proposal_accept", u64, i32, _f64p, _f64p, u64, u32, i64, i64, i32, i32, _f64p, i32, _f64p, _f64p)
    sig("o_max_flags", None, _f64p, i64, pf64, pi32)
    sig("o_fixq", None, _f64p, i64, f64, i32, i32, _u64p)
    sig("o_scan", u64, _u64p, i64, _u64p, pu64, pu64)
    sig("o_set_gid_map", None, C.POINTER(C.c_int32))
    sig("o_targets_multinomial", None, u64, u32, i64, i64, u64, _u64p)
    sig("o_targets_sorted", None, u64, u32, i64, i64, u64, _u64p)
    sig("o_gamma_E", C.c_int32, i64)
    sig("o_spacing_d", u64, u64, u32, u32)
    sig("o_gamma_tile_d", u64, u64, u32, u32, i64, C.c_int32)
    sig("o_targets_stratified", None, u64, u32, i64, i64, i64, u64, _u64p)
    sig("o_targets_stratified_view", None, u64, u32, i64, i64, u64, _u64p)
    sig("o_upper_bound", None, _u64p, i64, _u64p, i64, _i64p)
    sig("o_residual_shift", i32, u64, i64)
    sig("o_residual_split", None, _u64p, i64, i64, u64, i32, _u64p, _u64p)
    sig("o_gather_rows", None, _f64p, i32, _i64p, i64, _f64p)
    sig("o_argsort_desc", None, _f64p, i64, _i64p)
    sig("o_wsum", f64, _u64p, u64, _f64p, i32, i32, i64, i32, f64)
    sig("o_normals", None, u64, u32, i64, _f64p)
    sig("o_set_threads", i32, i32)
    sig("o_set_gid_stride", None, i64)
    sig("o_dereplicate_sample", None, _f64p, i64, i64, i32, i32, u64, u32, _i64p, _f64p)
    # literal Float64 restatement (ref_literal.c)
    sig("lit_logsumexp", f64, _f64p, i64); sig("lit_lognorm", None, _f64p, i64, _f64p)
    sig("lit_softmax", None, _f64p, i64, _f64p); sig("lit_safe_softmax", i32, _f64p, i64, _f64p)
    sig("lit_ess", f64, _f64p, i64)
    sig("lit_multinomial", None, _f64p, i64, _f64p, _i64p)
    sig("lit_residual", i64, _f64p, i64, _f64p, _i64p)
    sig("lit_stratified", None, _f64p, _i64p, i64, _f64p, _i64p)
    sig("lit_update_weights", None, _f64p, _f64p, _i64p, i64, _f64p)
    sig("lit_inv_w_threshold", f64, _f64p, i64, i64)
    sig("lit_systematic", i64, _f64p, i64, i64, f64, _i64p)
    L.o_set_threads(1)            # the reference is single-threaded; bench.py raises this for its all-cores leg only
    _lib = L
    return L


def set_threads(n: int) -> int:
    """OpenMP threads of the per-particle loops (results are independent of it); returns the count in effect"""
    return lib().o_set_threads(int(n))


# ----------------------------------------------------------------------------- thin primitive wrappers
def olog(x: float) -> float:
    return lib().o_log_d(float(x))


def fix_K(n_global: int) -> int:
    return lib().o_fix_K(int(n_global))


def row_width(model: int, keep_prev: bool) -> int:
    d = lib().o_model_dim(model)
    w = 2 * d if keep_prev else d
    return w + (w & 1)


def max_flags(lp: np.ndarray):
    m, f = C.c_double(), C.c_int()
    lib().o_max_flags(np.ascontiguousarray(lp, np.float64), lp.size, C.byref(m), C.byref(f))
    return m.value, f.value


def fixq(lp: np.ndarray, m: float, K: int, uniform: bool = False) -> np.ndarray:
    q = np.empty(lp.size, np.uint64)
    lib().o_fixq(np.ascontiguousarray(lp, np.float64), lp.size, m, K, int(uniform), q)
    return q


def scan(q: np.ndarray):
    cdf = np.empty(q.size, np.uint64)
    hi, lo = C.c_uint64(), C.c_uint64()
    S = lib().o_scan(np.ascontiguousarray(q), q.size, cdf, C.byref(hi), C.byref(lo))
    return cdf, int(S), int(hi.value), int(lo.value)


def upper_bound(cdf: np.ndarray, T: np.ndarray) -> np.ndarray:
    idx = np.empty(T.size, np.int64)
    lib().o_upper_bound(np.ascontiguousarray(cdf), cdf.size, np.ascontiguousarray(T), T.size, idx)
    return idx


def targets_multinomial(seed, epoch, j0, n, S) -> np.ndarray:
    T = np.empty(n, np.uint64)
    lib().o_targets_multinomial(seed, epoch, j0, n, S, T)
    return T


def targets_sorted(seed, epoch, j0, n, S) -> np.ndarray:
    """the n targets of the opt-in "multinomial_sorted" resampler (gpf_oracle.c o_targets_sorted): non-decreasing"""
    T = np.empty(n, np.uint64)
    lib().o_targets_sorted(seed, epoch, j0, n, S, T)
    return T


def targets_stratified(seed, epoch, j0, n, N, S) -> np.ndarray:
    T = np.empty(n, np.uint64)
    lib().o_targets_stratified(seed, epoch, j0, n, N, S, T)
    return T


def argsort_desc(lp: np.ndarray) -> np.ndarray:
    order = np.empty(lp.size, np.int64)
    lib().o_argsort_desc(np.ascontiguousarray(lp, np.float64), lp.size, order)
    return order


def gather_rows(rows: np.ndarray, idx: np.ndarray) -> np.ndarray:
    out = np.empty((idx.size, rows.shape[1]), np.float64)
    lib().o_gather_rows(np.ascontiguousarray(rows), rows.shape[1], np.ascontiguousarray(idx, np.int64),
                        idx.size, out)
    return out


def normals(seed: int, epoch: int, n: int) -> np.ndarray:
    out = np.empty(n, np.float64)
    lib().o_normals(seed, epoch, n, out)
    return out


class OracleError(RuntimeError):
    """Mirrors Julia's ErrorException raised by error(...) in the reference."""


class WeightSummary:
    """m, flags, fixed-point weights, their exact CDF, S = sum q, Q = sum q^2."""

    def __init__(self, lp: np.ndarray, n_global: int, K: int | None = None):
        self.K = fix_K(n_global) if K is None else K
        self.m, self.flags = max_flags(lp)
        self.uniform = bool(self.flags & FLAG_ALL_NEGINF)
        self.bad = bool(self.flags & (FLAG_NAN | FLAG_POSINF))
        if self.bad:
            self.q = np.zeros(lp.size, np.uint64)
        else:
            self.q = fixq(lp, self.m, self.K, self.uniform)
        self.cdf, self.S, self.Qhi, self.Qlo = scan(self.q)

    @property
    def lse(self) -> float:
        return lib().o_lse_from(self.m, self.S, self.K, self.flags)

    @property
    def ess(self) -> float:
        if self.flags:
            return float("nan")   # lognorm of all -Inf / NaN weights is NaN in the reference too
        return lib().o_ess_from(self.S, self.Qhi, self.Qlo)


class OracleFilter:
    """Single-shard oracle with the state of Gen.ParticleFilterState:
    rows (traces), log_weights, log_ml_est, parents (1-based, into the pre-resample array)."""

    def __init__(self, model: int, params, n_particles: int, seed: int, keep_prev: bool = False, history: bool = False):
        self.history = bool(history)
        self.hist_x, self.hist_map = [], []      # per step: latent columns (final order of the step), composed ancestors
        self.model, self.n, self.seed = int(model), int(n_particles), int(seed)
        self.params = np.ascontiguousarray(params, np.float64)
        self.keep_prev = bool(keep_prev)
        self.d = lib().o_model_dim(self.model)
        self.W = row_width(self.model, self.keep_prev)
        self.rows = np.zeros((self.n, self.W))
        self.lw = np.zeros(self.n)
        self.lml_est = 0.0
        self.parents = np.arange(1, self.n + 1, dtype=np.int64)    # initialize.jl:43  collect(1:N)
        self.epoch = 0
        self.has_prev = False
        self.last_obs = None
        self.n_accepted = 0

    # -- initialize.jl:31-44
    def initialize(self, obs, proposal: bool = False, strata=None, layout: str = "contiguous"):
        obs = np.ascontiguousarray(obs, np.float64)
        self.hist_x, self.hist_map = [None], [None]
        if strata is not None:                                           # initialize.jl:92-109 + stratified_map!, utils.jl:29-55
            v = np.ascontiguousarray(strata, np.float64)
            f = lib().o_init_strata_proposal if proposal else lib().o_init_strata     # initialize.jl:111-129 / :92-109
            f(self.model, self.params, self.seed, self.epoch, 0, self.n, self.W, obs, v, v.size,
              int(layout != "contiguous"), olog(float(v.size)), self.rows, self.lw)
        else:
            f = lib().o_init_proposal if proposal else lib().o_init      # initialize.jl:46-62 / :31-44
            f(self.model, self.params, self.seed, self.epoch, 0, self.n, self.W, obs, self.rows, self.lw)
        self.lml_est = 0.0
        self.parents = np.arange(1, self.n + 1, dtype=np.int64)
        self.epoch += 1
        self.has_prev = False
        self.last_obs = obs
        return self

    # -- update.jl:12-25
    def update(self, obs, proposal: bool = False, strata=None, layout: str = "interleaved"):
        obs = np.ascontiguousarray(obs, np.float64)
        if self.history:                                            # the step that ends now, in its final order
            self.hist_x[-1] = self.rows[:, :self.d].copy()
            self.hist_x.append(None); self.hist_map.append(None)
        new_rows = np.empty_like(self.rows)
        if strata is not None:                                           # update.jl:193-210 + stratified_map!, utils.jl:29-55
            v = np.ascontiguousarray(strata, np.float64)
            lib().o_step_strata(self.model, self.params, self.seed, self.epoch, 0, self.n, self.W, int(self.keep_prev), obs, v, v.size,
                                int(layout != "contiguous"), olog(float(v.size)), self.rows, new_rows, self.lw)
        else:
            f = lib().o_step_proposal if proposal else lib().o_step      # update.jl:79-96 / :12-25
            f(self.model, self.params, self.seed, self.epoch, 0, self.n, self.W, int(self.keep_prev),
              obs, self.rows, new_rows, self.lw)                    # :15-22
        self.rows = new_rows                                        # update_refs!, utils.jl:10-15
        self.epoch += 1
        self.has_prev = True
        self.last_obs = obs
        return self

    # -- utils.jl:148,156,163-164,171; Gen.log_ml_estimate
    def summary(self) -> WeightSummary:
        return WeightSummary(self.lw, self.n)

    def effective_sample_size(self) -> float:
        return self.summary().ess

    def log_ml_estimate(self) -> float:
        return self.lml_est + self.summary().lse - olog(float(self.n))

    def log_norm_weights(self) -> np.ndarray:
        return self.lw - self.summary().lse                         # lognorm, utils.jl:100

    def norm_weights(self) -> np.ndarray:
        s = self.summary()
        if s.uniform:                                               # all -Inf: maximum = -Inf, vs .- maximum = NaN -- the PLAIN softmax
            return np.full(self.n, np.nan)                          # (utils.jl:103-107); the uniform fallback is safe_softmax's (:123-126)
        with np.errstate(invalid="ignore", divide="ignore"):
            return s.q.astype(np.float64) / float(s.S)              # softmax, utils.jl:103-107

    # -- resample.jl:19-30 dispatcher + :48-175
    def resample(self, method: str = "multinomial", priority_alpha=None, log_priorities=None,
                 sort_particles: bool = True, check="warn"):
        if method not in METHODS:
            raise OracleError(f"Resampling method {method} not recognized.")   # :28
        N, lw = self.n, self.lw
        # :51-52 / :88-89 / :147-148
        if log_priorities is not None:
            lp, has_prio = np.ascontiguousarray(log_priorities, np.float64), True
        elif priority_alpha is not None:
            lp, has_prio = float(priority_alpha) * lw, True
        else:
            lp, has_prio = lw, False
        # :54 safe_softmax (utils.jl:117-140)
        sp = WeightSummary(lp, N)
        invalid = sp.flags != 0
        if check is True and invalid:
            raise OracleError("Invalid weights.")                   # :55
        if sp.bad:
            # NaN weights: Distributions.Categorical rejects them in the reference; we fail loudly too
            raise OracleError("Invalid weights (NaN).")
        # :57 update_lml_est! (:178-182) -- from the RAW log weights
        sr = WeightSummary(lw, N) if has_prio else sp
        self.lml_est = self.lml_est + (sr.lse - olog(float(N)))
        epoch = self.epoch
        if method == "multinomial":                                 # :59
            T = targets_multinomial(self.seed, epoch, 0, N, sp.S)
            anc = upper_bound(sp.cdf, T)
        elif method == "multinomial_sorted":                        # :59 with the uniforms drawn in sorted order (opt-in)
            T = targets_sorted(self.seed, epoch, 0, N, sp.S)
            anc = upper_bound(sp.cdf, T)
        elif method == "stratified":                                # :155-170
            if sort_particles:
                order = argsort_desc(lp)                            # :156-157
                cdf, S, _, _ = scan(sp.q[order])
            else:
                order, cdf, S = None, sp.cdf, sp.S
            T = targets_stratified(self.seed, epoch, 0, N, N, S)    # :160-162
            k = upper_bound(cdf, T)                                 # :163-166
            anc = order[k] if order is not None else k              # :168
        else:                                                       # residual :96-115
            sh = lib().o_residual_shift(sp.S, N)
            c = np.empty(N, np.uint64); r = np.empty(N, np.uint64)
            lib().o_residual_split(sp.q, N, N, sp.S, sh, c, r)      # :99, :109
            ccdf = np.cumsum(c, dtype=np.uint64)
            n_res = int(ccdf[-1])                                   # n_resampled
            anc = np.empty(N, np.int64)
            anc[:n_res] = upper_bound(ccdf, np.arange(n_res, dtype=np.uint64))   # :101
            if n_res < N:                                           # :108
                rcdf, Rs, _, _ = scan(r)                            # :110 (normalisation == using Rs)
                T = targets_multinomial(self.seed, epoch, n_res, N - n_res, Rs)
                anc[n_res:] = upper_bound(rcdf, T)                  # :113
        if self.history:                                            # persistent traces: remember who descends from whom
            self.hist_map[-1] = anc.copy() if self.hist_map[-1] is None else self.hist_map[-1][anc]
        new_rows = gather_rows(self.rows, anc)                      # :60 / :103,114 / :169
        # update_weights! :190-202
        if not has_prio:
            new_lw = np.zeros(N)                                    # :195
        else:
            log_ws = lw[anc] - lp[anc]                              # :198
            s2 = WeightSummary(log_ws, N)
            new_lw = log_ws + (olog(float(N)) - s2.lse)             # :200
        self.parents = anc + 1                                      # 1-based like Julia
        self.rows, self.lw = new_rows, new_lw                       # update_refs!
        self.epoch += 1
        return invalid

    # -- rejuvenate.jl:18-27 dispatcher, :40-53 move-accept, :74-90 move-reweight
    def rejuvenate(self, method: str = "move", n_iters: int = 1, proposal=None):
        """proposal: the parameter vector of the model's native move proposal -> move_reweight(trace, proposal, proposal_args),
        rejuvenate.jl:134-148, or (method "move") Gen.mh(trace, proposal, proposal_args) (() for the LG-SSM's locally optimal proposal,
        (q, log q, log(1-q)) for line_model's outlier proposal)"""
        if method not in ("move", "reweight"):
            raise OracleError(f"Method {method} not recognized.")   # :25
        new_rows = np.empty_like(self.rows)
        if proposal is not None and method == "move":                   # Gen.mh(trace, proposal, proposal_args) under pf_move_accept! (:40-53)
            q = np.zeros(4); q[:len(proposal)] = proposal
            self.n_accepted = int(lib().o_move_proposal_accept(self.model, self.params, q, self.seed, self.epoch, 0, self.n, self.W,
                                                               int(self.has_prev), self.last_obs, int(n_iters), self.rows, new_rows))
            self.rows = new_rows
            self.epoch += 1
            return self
        if proposal is not None:
            q = np.zeros(4); q[:len(proposal)] = proposal
            lib().o_move_proposal(self.model, self.params, q, self.seed, self.epoch, 0, self.n, self.W, int(self.has_prev),
                                  self.last_obs, int(n_iters), self.rows, new_rows, self.lw)
            self.n_accepted = self.n * int(n_iters)
            self.rows = new_rows
            self.epoch += 1
            return self
        self.n_accepted = int(lib().o_move(self.model, self.params, self.seed, self.epoch, 0, self.n, self.W,
                                           int(self.has_prev), self.last_obs, int(n_iters),
                                           int(method == "reweight"), self.rows, new_rows, self.lw))
        self.rows = new_rows
        self.epoch += 1
        return self

    # ------------------------------------------------------------------ resize family, src/resize.jl
    def _set_count(self, n_new: int):
        self.n = int(n_new)

    def resize(self, n_particles: int, method: str = "multinomial", priority_alpha=None, check="warn"):
        """pf_resize! dispatcher (resize.jl:16-28) + pf_multinomial_resize! (:46-68) / pf_residual_resize! (:87-124)"""
        if method == "optimal":
            return self.optimal_resize(n_particles, check=check)                        # :22-23
        if method not in ("multinomial", "residual"):
            raise OracleError(f"Resampling method {method} not recognized.")           # :26
        n_old, n_new, lw = self.n, int(n_particles), self.lw
        K = fix_K(max(n_old, n_new))                  # both N_old 2^K and n_new 2^K below 2^62
        lp, has_prio = (lw, False) if priority_alpha is None else (float(priority_alpha) * lw, True)
        sp = WeightSummary(lp, n_old, K)                                                # safe_softmax, :54/:96
        invalid = sp.flags != 0
        if (check is True and invalid) or sp.bad:
            raise OracleError("Invalid weights.")                                       # :56/:97
        sr = WeightSummary(lw, n_old, K) if has_prio else sp
        self.lml_est = self.lml_est + (sr.lse - olog(float(n_old)))                     # update_lml_est!, :58/:99
        epoch = self.epoch
        if method == "multinomial":
            anc = upper_bound(sp.cdf, targets_multinomial(self.seed, epoch, 0, n_new, sp.S))   # :63
        else:
            sh = lib().o_residual_shift(sp.S, n_new)
            c = np.empty(n_old, np.uint64); r = np.empty(n_old, np.uint64)
            lib().o_residual_split(sp.q, n_old, n_new, sp.S, sh, c, r)                  # floor(n_particles * w), :106
            ccdf = np.cumsum(c, dtype=np.uint64)
            n_res = int(ccdf[-1])
            anc = np.empty(n_new, np.int64)
            anc[:n_res] = upper_bound(ccdf, np.arange(n_res, dtype=np.uint64))
            if n_res < n_new:                                                           # :115-122
                rcdf, Rs, _, _ = scan(r)
                anc[n_res:] = upper_bound(rcdf, targets_multinomial(self.seed, epoch, n_res, n_new - n_res, Rs))
        new_rows = gather_rows(self.rows, anc)                                          # :64
        if not has_prio:                                                                # update_weights!(state, n, lp), :424-438
            new_lw = np.zeros(n_new)
        else:
            log_ws = lw[anc] - lp[anc]
            new_lw = log_ws + (olog(float(n_new)) - WeightSummary(log_ws, n_new).lse)
        self.parents, self.rows, self.lw = anc + 1, new_rows, new_lw
        self._set_count(n_new)
        self.epoch += 1
        return invalid

    def optimal_resize(self, n_particles: int, check="warn"):
        """pf_optimal_resize! (resize.jl:149-200) with find_inv_w_threshold (resize.jl:203-219) in exact fixed point
        (DESIGN.md §8b): the inverse weight threshold c is the exact pair (a, B), c w_i >= 1 <=> a q_i >= B."""
        n_old, n = self.n, int(n_particles)
        if n < 1 or n > n_old:
            raise OracleError("optimal resize: need 1 <= n_particles <= current count")      # :185
        s = WeightSummary(self.lw, n_old)                                                   # safe_softmax, :152
        invalid = s.flags != 0
        if (check is True and invalid) or s.bad:
            raise OracleError("Invalid weights.")                                           # :153
        q = [int(v) for v in s.q]
        S = s.S
        qs = sorted(q, reverse=True)                                                        # sort(weights), :204
        pre = [0]
        for v in qs:
            pre.append(pre[-1] + v)
        a, B = n, S                                                                         # float(n_particles), :218
        for d in range(n - 1, -1, -1):       # first kappa ascending = largest position d descending; A = d, B = S - pre[d]
            if qs[d] > 0 and S - pre[d] <= (n - d) * qs[d]:                                 # B / kappa + A <= n, :212-213
                a, B = n - d, S - pre[d]                                                    # (n - A) / B, :215
                break
        keep = np.array([a * v >= B for v in q], bool)                                      # inv_w_thresh .* weights .>= 1, :156
        keep_idx = np.flatnonzero(keep)
        n_keep = keep_idx.size
        n_res = n - n_keep
        anc = np.empty(n, np.int64)
        anc[:n_keep] = keep_idx                                                             # :180
        if n_res > 0:
            sq = np.where(keep, 0, s.q).astype(np.uint64)
            scdf, Bp, _, _ = scan(sq)
            if Bp == 0:                                                                     # safe_softmax of the rest, :166-168
                invalid = True
                if check is True:
                    raise OracleError("Invalid weights.")
                scdf, Bp, _, _ = scan(np.where(keep, 0, 1).astype(np.uint64))
            x = int(targets_multinomial(self.seed, self.epoch, 0, 1, Bp)[0])                # u = rand() * step_size, :171
            beta, rho = divmod(Bp, n_res)
            T = np.array([m * beta + (m * rho + x) // n_res for m in range(n_res)], np.uint64)
            anc[n_keep:] = upper_bound(scdf, T)                                             # :172-178
        ratio = olog(float(n)) - olog(float(n_old))                                         # :189
        rw = lib().o_lse_from(s.m, B, s.K, s.flags) - olog(float(a))                        # log_tot_weight - log(inv_w_thresh), :192
        new_lw = np.empty(n)
        new_lw[:n_keep] = self.lw[keep_idx] + ratio                                         # :194
        new_lw[n_keep:] = rw + ratio                                                        # :195
        self.parents, self.rows, self.lw = anc + 1, gather_rows(self.rows, anc), new_lw     # :197
        self._set_count(n)
        self.epoch += 1
        self.n_keep = n_keep
        return invalid

    def replicate(self, n_replicates: int, layout: str = "contiguous"):
        """pf_replicate!, resize.jl:236-244"""
        k, n_old = int(n_replicates), self.n
        idx = np.repeat(np.arange(n_old), k) if layout == "contiguous" else np.tile(np.arange(n_old), k)
        self.parents, self.rows, self.lw = idx + 1, self.rows[idx].copy(), self.lw[idx].copy()
        self._set_count(n_old * k)
        return self

    def dereplicate(self, n_replicates: int, layout: str = "contiguous", method: str = "keepfirst"):
        """pf_dereplicate!, resize.jl:267-297"""
        k, n_old = int(n_replicates), self.n
        assert n_old % k == 0                                                           # :270
        n_new = n_old // k
        if method == "keepfirst":                                                       # :272-277
            idx = np.arange(0, n_old, k) if layout == "contiguous" else np.arange(n_new)
            new_lw = self.lw[idx].copy()
        else:                                                                           # :278-293
            idx = np.empty(n_new, np.int64); new_lw = np.empty(n_new)
            lib().o_dereplicate_sample(self.lw, n_new, n_old, k, int(layout != "contiguous"), self.seed, self.epoch, idx, new_lw)
            self.epoch += 1
        self.parents, self.rows, self.lw = idx + 1, self.rows[idx].copy(), new_lw
        self._set_count(n_new)
        return self

    # -- Gen.sample_unweighted_traces (utils.jl:189-194): categorical draws, state untouched
    def sample_unweighted(self, n_samples: int):
        s = self.summary()
        if s.bad:
            raise OracleError("Invalid weights (NaN).")
        idx = upper_bound(s.cdf, targets_multinomial(self.seed, self.epoch, 0, int(n_samples), s.S))
        self.epoch += 1
        return self.rows[idx].copy(), idx + 1

    # -- statistics.jl:13-14, 48-50
    def mean(self, col: int) -> float:
        s = self.summary()
        return lib().o_wsum(s.q, s.S, self.rows, self.W, col, self.n, 1, 0.0)

    def var(self, col: int) -> float:
        s = self.summary()
        mu = lib().o_wsum(s.q, s.S, self.rows, self.W, col, self.n, 1, 0.0)
        return lib().o_wsum(s.q, s.S, self.rows, self.W, col, self.n, 2, mu)

    def column(self, col: int) -> np.ndarray:
        return self.rows[:, col].copy()

    # -- trajectory store: statistics.jl:13-14,48-50 with a PAST address (README.md:97-104), step is 1-based
    def history_column(self, step: int, col: int) -> np.ndarray:
        T = len(self.hist_x)
        self.hist_x[-1] = self.rows[:, :self.d].copy()
        idx = np.arange(self.n)
        for q in range(T - 1, step - 1, -1):
            if self.hist_map[q] is not None:
                idx = self.hist_map[q][idx]
        return self.hist_x[step - 1][idx, col]

    def history_mean(self, step: int, col: int) -> float:
        s = self.summary()
        v = np.ascontiguousarray(self.history_column(step, col)).reshape(-1, 1)
        return lib().o_wsum(s.q, s.S, v, 1, 0, self.n, 1, 0.0)

    def history_var(self, step: int, col: int) -> float:
        s = self.summary()
        v = np.ascontiguousarray(self.history_column(step, col)).reshape(-1, 1)
        mu = lib().o_wsum(s.q, s.S, v, 1, 0, self.n, 1, 0.0)
        return lib().o_wsum(s.q, s.S, v, 1, 0, self.n, 2, mu)


class OracleSubState:
    """ParticleFilterSubState (view.jl:16-48) over the range start : step : start + (count-1) step of an OracleFilter
    (view.jl:35-48 takes any index vector; the reference's tests use contiguous and strided ranges, e.g. state[k:5:100],
    test/update.jl:33): numpy views alias the source's arrays; semantics of resample.jl:185-187,205-218 and
    utils.jl:17-20,174-178.  Per-particle RNG counters stay the GLOBAL particle ids start + i*step; the resample stream of a
    view is indexed by the slot ids start, start + 1, ... (consecutive from the view's first particle), whatever the step."""

    def __init__(self, source: OracleFilter, start: int, count: int, step: int = 1, index=None):
        """index: an arbitrary vector of distinct particle indices (state[idxs], view.jl:35-48) -- then start = index[0], the per-particle
        RNG counters are the particles' own ids index[i], the resample stream the slot ids index[0], index[0] + 1, ..."""
        self.source, self.start, self.n, self.step = source, int(start), int(count), int(step)
        if index is None:
            self.sl = slice(self.start, self.start + (self.n - 1) * self.step + 1, self.step)
            self.gid_map = None
        else:
            self.sl = np.ascontiguousarray(index, np.int64)
            assert self.sl.size == np.unique(self.sl).size == self.n and self.start == int(self.sl[0])
            self.gid_map = np.ascontiguousarray(self.sl - self.sl[0], np.int32)
        self.last_obs = source.last_obs
        self.n_accepted = 0

    def _Stride(self, step):
        """per-particle oracle calls inside: local particle i carries the RNG counter gid0 + i*step (or gid0 + gid_map[i])"""
        view = self

        class _Ctx:
            def __enter__(self_):
                lib().o_set_gid_stride(step)
                if view.gid_map is not None:
                    lib().o_set_gid_map(view.gid_map.ctypes.data_as(C.POINTER(C.c_int32)))

            def __exit__(self_, *a):
                lib().o_set_gid_stride(1)
                lib().o_set_gid_map(None)
        return _Ctx()

    # aliases of the source's arrays (the source may swap its row buffer: always re-derive)
    @property
    def rows(self): return self.source.rows[self.sl]
    @property
    def lw(self): return self.source.lw[self.sl]
    @property
    def parents(self): return self.source.parents[self.sl]

    def summary(self) -> WeightSummary:
        return WeightSummary(np.ascontiguousarray(self.lw), self.n)

    def effective_sample_size(self) -> float:
        return self.summary().ess

    def log_ml_estimate(self) -> float:                                 # utils.jl:174-178
        return self.source.lml_est + self.summary().lse - olog(float(self.n))

    def update(self, obs, proposal: bool = False, strata=None, layout: str = "interleaved"):   # update.jl:12-25 / :193-210 on the view + utils.jl:17-20
        s = self.source
        obs = np.ascontiguousarray(obs, np.float64)
        rin = np.ascontiguousarray(self.rows); lw = np.ascontiguousarray(self.lw)
        rout = np.empty_like(rin)
        with self._Stride(self.step):
            if strata is not None:
                v = np.ascontiguousarray(strata, np.float64)
                lib().o_step_strata(s.model, s.params, s.seed, s.epoch, self.start, self.n, s.W, int(s.keep_prev), obs, v, v.size,
                                    int(layout != "contiguous"), olog(float(v.size)), rin, rout, lw)
            else:
                f = lib().o_step_proposal if proposal else lib().o_step
                f(s.model, s.params, s.seed, s.epoch, self.start, self.n, s.W, int(s.keep_prev), obs, rin, rout, lw)
        s.rows[self.sl] = rout; s.lw[self.sl] = lw
        s.epoch += 1; s.has_prev = True; self.last_obs = obs
        return self

    def rejuvenate(self, method: str = "move", n_iters: int = 1):
        s = self.source
        rin = np.ascontiguousarray(self.rows); lw = np.ascontiguousarray(self.lw)
        rout = np.empty_like(rin)
        with self._Stride(self.step):
            self.n_accepted = int(lib().o_move(s.model, s.params, s.seed, s.epoch, self.start, self.n, s.W, int(s.has_prev),
                                               self.last_obs, int(n_iters), int(method == "reweight"), rin, rout, lw))
        s.rows[self.sl] = rout; s.lw[self.sl] = lw
        s.epoch += 1
        return self

    def resample(self, method: str = "multinomial", priority_alpha=None, sort_particles: bool = True, check="warn"):
        if method not in METHODS:
            raise OracleError(f"Resampling method {method} not recognized.")
        s, N = self.source, self.n
        lw = np.ascontiguousarray(self.lw)
        lp, has_prio = (lw, False) if priority_alpha is None else (float(priority_alpha) * lw, True)
        sp = WeightSummary(lp, N)
        invalid = sp.flags != 0
        if (check is True and invalid) or sp.bad:
            raise OracleError("Invalid weights.")
        sr = WeightSummary(lw, N) if has_prio else sp                   # update_lml_est! is a no-op (resample.jl:185-187)
        epoch = s.epoch
        if method == "multinomial":
            anc = upper_bound(sp.cdf, targets_multinomial(s.seed, epoch, self.start, N, sp.S))
        elif method == "multinomial_sorted":                            # (slot ids start, start + 1, ... like the other streams of a view)
            anc = upper_bound(sp.cdf, targets_sorted(s.seed, epoch, self.start, N, sp.S))
        elif method == "stratified":
            if sort_particles:
                order = argsort_desc(lp); cdf, S, _, _ = scan(sp.q[order])
            else:
                order, cdf, S = None, sp.cdf, sp.S
            # strata are local to the view, RNG counters keep the global particle id
            T = _targets_stratified_view(s.seed, epoch, self.start, N, S)
            k = upper_bound(cdf, T)
            anc = order[k] if order is not None else k
        else:
            sh = lib().o_residual_shift(sp.S, N)
            c = np.empty(N, np.uint64); r = np.empty(N, np.uint64)
            lib().o_residual_split(sp.q, N, N, sp.S, sh, c, r)
            ccdf = np.cumsum(c, dtype=np.uint64); n_res = int(ccdf[-1])
            anc = np.empty(N, np.int64)
            anc[:n_res] = upper_bound(ccdf, np.arange(n_res, dtype=np.uint64))
            if n_res < N:
                rcdf, Rs, _, _ = scan(r)
                anc[n_res:] = upper_bound(rcdf, targets_multinomial(s.seed, epoch, self.start + n_res, N - n_res, Rs))
        new_rows = gather_rows(np.ascontiguousarray(self.rows), anc)
        if not has_prio:
            new_lw = np.full(N, sr.lse - olog(float(N)))                # resample.jl:210
        else:
            log_ws = lw[anc] - lp[anc]                                  # :213
            new_lw = log_ws + (sr.lse - WeightSummary(log_ws, N).lse)   # :215-216
        s.rows[self.sl] = new_rows; s.lw[self.sl] = new_lw
        s.parents[self.sl] = anc + 1                                    # local to the view, like the reference
        s.epoch += 1
        return invalid

    def mean(self, col: int) -> float:
        s = self.summary()
        return lib().o_wsum(s.q, s.S, np.ascontiguousarray(self.rows), self.source.W, col, self.n, 1, 0.0)


def _targets_stratified_view(seed, epoch, start, n, S) -> np.ndarray:
    """stratified targets with LOCAL stratum index j and RNG counter keyed by the global id start + j"""
    T = np.empty(n, np.uint64)
    lib().o_targets_stratified_view(seed, epoch, start, n, S, T)
    return T


# ---------------------------------------------------------------------- many small filters in one state: loops over sub-states
# (the batched device forms are gpf_resample_blocks / gpf_update_blocks / gpf_rejuvenate_blocks, gpf_k_block.hpp; every block runs
#  under the call's one epoch, which then advances once -- like one call on the whole state)
def blocks_of(f: OracleFilter, nb: int):
    return [(b0, min(b0 + nb, f.n)) for b0 in range(0, f.n, nb)]


def resample_blocks(f: OracleFilter, nb: int, method: str = "multinomial", ess_frac=None, sort_particles: bool = True, check=False,
                    priority_alpha=None) -> np.ndarray:
    """for b in blocks: if ess_frac is None or ESS(state[b]) < ess_frac * len(b): pf_resample!(state[b], method); returns the mask"""
    e, mask = f.epoch, []
    for a, b in blocks_of(f, nb):
        v = f[a:b]
        f.epoch = e
        go = ess_frac is None or v.effective_sample_size() < ess_frac * v.n      # (NaN ESS of invalid weights: False, like the reference's `<`)
        if go:
            v.resample(method, priority_alpha=priority_alpha, sort_particles=sort_particles, check=check)
        mask.append(bool(go))
    f.epoch = e + 1
    return np.array(mask)


def update_blocks(f: OracleFilter, nb: int, obs_rows, proposals=None, strata=None, layout: str = "interleaved") -> None:
    """for b in blocks: pf_update!(state[b], ..., observations[b][, proposal, proposal_args])   (per-view updates, also with different
    proposals per view, test/update.jl:179-189); proposals: one bool per block"""
    e = f.epoch
    for k, (a, b) in enumerate(blocks_of(f, nb)):
        f.epoch = e
        if strata is not None:                                              # every block stratified by itself (update.jl:193-210 on the sub-state)
            f[a:b].update(np.asarray(obs_rows[k], np.float64), strata=strata, layout=layout)
        else:
            f[a:b].update(np.asarray(obs_rows[k], np.float64), proposal=bool(proposals[k]) if proposals is not None else False)
    f.epoch = e + 1


def rejuvenate_blocks(f: OracleFilter, nb: int, obs_rows, method: str = "move", mask=None, n_iters: int = 1) -> int:
    """for b in blocks (those with mask[b], if given): pf_rejuvenate!(state[b], ...) with the block's observation; returns the accept count"""
    e, acc = f.epoch, 0
    for k, (a, b) in enumerate(blocks_of(f, nb)):
        if mask is not None and not mask[k]:
            continue
        f.epoch = e
        v = f[a:b]; v.last_obs = np.asarray(obs_rows[k], np.float64)
        v.rejuvenate(method, n_iters); acc += v.n_accepted
    f.epoch = e + 1
    return acc


def initialize_blocks(f: OracleFilter, nb: int, obs_rows, strata=None, layout: str = "contiguous") -> OracleFilter:
    """per-block initialisation (initialize.jl:39-41 on every sub-state with its own observation; with strata: initialize.jl:92-109)"""
    for k, (a, b) in enumerate(blocks_of(f, nb)):
        rows = np.zeros((b - a, f.W)); lw = np.zeros(b - a)
        if strata is not None:
            v = np.ascontiguousarray(strata, np.float64)
            lib().o_init_strata(f.model, f.params, f.seed, f.epoch, a, b - a, f.W, np.ascontiguousarray(obs_rows[k], np.float64), v, v.size,
                                int(layout != "contiguous"), olog(float(v.size)), rows, lw)
        else:
            lib().o_init(f.model, f.params, f.seed, f.epoch, a, b - a, f.W, np.ascontiguousarray(obs_rows[k], np.float64), rows, lw)
        f.rows[a:b] = rows; f.lw[a:b] = lw
    f.lml_est = 0.0; f.parents = np.arange(1, f.n + 1, dtype=np.int64)
    f.epoch += 1; f.has_prev = False
    return f


def _oracle_getitem(self, idx):
    if not isinstance(idx, slice):                                       # state[idxs] with any vector of distinct indices (view.jl:35-48)
        ix = np.asarray(idx, np.int64)
        return OracleSubState(self, int(ix[0]), ix.size, 1, index=ix)
    start, stop, step = idx.indices(self.n)
    assert step >= 1 and stop > start
    return OracleSubState(self, start, (stop - start + step - 1) // step, step)


OracleFilter.__getitem__ = _oracle_getitem
