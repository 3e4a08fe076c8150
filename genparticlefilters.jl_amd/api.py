"""Host-side mirror of the reference's operator interface for the hot path.

Same names, argument meaning and error behaviour as GenParticleFilters.jl v0.2.3 (Julia's `f!`
is spelled `f` here; the Julia glue in julia/GenParticleFiltersAMD.jl keeps the bang):

    pf_initialize(model, model_args, observations, n_particles)        src/initialize.jl:31-44
    pf_update(state, new_args, argdiffs, observations)                 src/update.jl:12-25
    pf_resample(state, method; priority_fn, check[, sort_particles])   src/resample.jl:19-175
    pf_rejuvenate(state, kern, kern_args, n_iters; method)             src/rejuvenate.jl:18-90
    effective_sample_size / get_ess / log_ml_estimate / get_lml_est /
    get_log_weights / get_log_norm_weights / get_norm_weights          src/utils.jl:148-186
    mean / var                                                         src/statistics.jl:13-14,48-50

Every call goes through the C ABI of libgpf_hip.so (include/gpf.h); nothing is computed on the CPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings

import numpy as np

from . import _lib
from .models import NativeModel

RESAMPLE_METHODS = {"multinomial": 0, "residual": 1, "stratified": 2,
                    "multinomial_sorted": 4}   # opt-in extension (gpf.h GPF_RESAMPLE_MULTINOMIAL_SORTED): ancestors come out non-decreasing
REJUVENATE_METHODS = {"move": 0, "reweight": 1}


class ErrorException(RuntimeError):
    """Julia's ErrorException, raised where the reference calls error(...)."""


class Tempering:
    """priority_fn = w -> alpha * w (reference test/resample.jl:15 uses alpha = 1/2); evaluated on the GPU."""

    def __init__(self, alpha: float):
        self.alpha = float(alpha)

    def __call__(self, w):
        return self.alpha * w


# native rejuvenation kernels (the `kern` argument of pf_rejuvenate)
class _NativeKernel:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f"<native kernel {self.name}>"


# native proposal for the custom-proposal forms of pf_initialize / pf_update (src/initialize.jl:46-62, src/update.jl:79-96)
locally_optimal = _NativeKernel("locally_optimal")     # exact conditional q(x_t | x_{t-1}, y_t); lgssm2 only
# the proposals of the reference's own tests for line_model as one native proposal: slope ~ uniform_discrete(0, 0) at the
# first step (test/initialize.jl:16-17), outlier ~ bernoulli(0.0) at every step (test/initialize.jl:18-19, test/update.jl:42-43)
line_fixed = _NativeKernel("line_fixed")
_PROPOSAL_IDS = {"locally_optimal": 1, "line_fixed": 2}


def _proposal_id(proposal) -> int:
    if not isinstance(proposal, _NativeKernel) or proposal.name not in _PROPOSAL_IDS:
        raise ErrorException("device filters support native proposals only (`locally_optimal`, `line_fixed`)")
    return _PROPOSAL_IDS[proposal.name]

mh = _NativeKernel("mh")                        # Gen.mh(trace, select(current step latent))
move_reweight = _NativeKernel("move_reweight")  # move_reweight(trace, selection), src/rejuvenate.jl:125-132


def _pd(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class DeviceParticleFilterState:
    """Device-resident counterpart of Gen.ParticleFilterState{U} (fields traces / new_traces /
    log_weights / log_ml_est / parents).  Owns an opaque libgpf handle."""

    def __init__(self, model: NativeModel, n_particles: int, seed: int = 1, keep_prev: bool = False,
                 device: int = 0, n_global: int | None = None, gid0: int = 0, stream: int | None = None,
                 history: int = 0):
        self._L = _lib.load()
        self.model, self.n_particles, self.seed = model, int(n_particles), int(seed)
        self.keep_prev = bool(keep_prev)
        self._params = np.ascontiguousarray(model.params, np.float64)
        cfg = _lib.GpfConfig()
        cfg.abi_version, cfg.model, cfg.n_params = _lib.ABI_VERSION, model.model_id, self._params.size
        cfg.keep_prev, cfg.params = int(self.keep_prev), _pd(self._params)
        cfg.n_particles = self.n_particles
        cfg.n_global = self.n_particles if n_global is None else int(n_global)
        cfg.gid0, cfg.seed, cfg.device, cfg.stream = int(gid0), self.seed, int(device), stream
        self._h = C.c_void_p()
        st = self._L.gpf_create(C.byref(cfg), C.byref(self._h))
        if st != _lib.OK:
            msg = self._L.gpf_last_error(None).decode()
            self._h = None
            raise ErrorException(f"gpf_create failed ({st}): {msg}")
        d, w = C.c_int32(), C.c_int32()
        self._L.gpf_state_dim(self._h, C.byref(d), C.byref(w))
        self.dim, self.row_width = d.value, w.value
        if history:                      # trajectory store for `history` time steps (persistent-trace queries)
            self._check(self._L.gpf_history_enable(self._h, int(history)))

    # -- state[idxs] / view(state, idxs): a sub-state over a range start:step:stop or over ANY vector of distinct indices
    #    (src/view.jl:35-48 takes `idxs::AbstractVector`; the reference's tests use contiguous and strided ranges, state[1:50],
    #    state[k:5:100]).  0-based half-open Python slices / ranges, or a list / NumPy array of 0-based indices.
    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.n_particles)
        elif isinstance(idx, range):
            start, stop, step = idx.start, idx.stop, idx.step
        elif isinstance(idx, (list, tuple, np.ndarray)):
            ix = np.asarray(idx)
            if ix.dtype == bool:                                    # state[mask], like Julia's logical indexing
                if ix.shape != (self.n_particles,):
                    raise ErrorException("a boolean index must have one entry per particle")
                ix = np.flatnonzero(ix)
            if ix.ndim != 1 or ix.size == 0 or not np.issubdtype(ix.dtype, np.integer):
                raise ErrorException("device sub-states over an index vector need a non-empty 1-D integer array")
            return DeviceParticleFilterSubState(self, int(ix[0]), ix.size, 0, index=np.ascontiguousarray(ix, np.int64))
        else:
            raise TypeError("device sub-states: use state[a:b], state[a:b:step] or state[index_vector]")
        if step < 1 or stop <= start:
            raise ErrorException("device sub-states cover non-empty ranges with a positive step")
        return DeviceParticleFilterSubState(self, start, (stop - start + step - 1) // step, step)

    def view(self, idx):
        return self[idx]

    # -- plumbing
    def _check(self, st: int):
        if st != _lib.OK:
            raise ErrorException(self._L.gpf_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.gpf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._L.gpf_synchronize(self._h))

    def checkpoint(self) -> np.ndarray:
        """gpf.h gpf_checkpoint_save: the whole state (population, log-weights, parents, log-ML estimate, RNG epoch, latest observation) as one uint8 array;
        `restore` on a state created with the same arguments continues bit for bit"""
        nb = C.c_int64(0)
        self._check(self._L.gpf_checkpoint_size(self._h, C.byref(nb)))
        out = np.empty(nb.value, np.uint8)
        self._check(self._L.gpf_checkpoint_save(self._h, out.ctypes.data, nb.value))
        return out

    def restore(self, blob):
        """gpf.h gpf_checkpoint_load"""
        blob = np.ascontiguousarray(np.frombuffer(blob, np.uint8) if isinstance(blob, (bytes, bytearray, memoryview)) else blob, np.uint8)
        self._check(self._L.gpf_checkpoint_load(self._h, blob.ctypes.data, blob.size))
        return self

    def set_lazy_search(self, enable: bool = True):
        """gpf.h gpf_set_lazy_search: pf_resample(state, "multinomial") leaves its ancestor search to the pf_update that follows (one fused
        kernel); same results, off by default"""
        self._check(self._L.gpf_set_lazy_search(self._h, int(bool(enable))))
        return self

    # -- fields of ParticleFilterState
    @property
    def log_weights(self) -> np.ndarray:
        out = np.empty(self.n_particles)
        self._check(self._L.gpf_get_log_weights(self._h, _pd(out), out.size))
        return out

    @log_weights.setter
    def log_weights(self, lw):
        lw = np.ascontiguousarray(lw, np.float64)
        self._check(self._L.gpf_set_log_weights(self._h, _pd(lw), lw.size))

    @property
    def parents(self) -> np.ndarray:
        out = np.empty(self.n_particles, np.int64)
        self._check(self._L.gpf_get_parents(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), out.size))
        return out

    @property
    def log_ml_est(self) -> float:
        # the running estimate alone = log_ml_estimate - (logsumexp(lw) - log N); exposed via get_lml_est
        raise AttributeError("use get_lml_est(state); the running log_ml_est lives on the device")

    @property
    def traces(self) -> np.ndarray:
        """(n_particles, row_width) Float64 rows: columns 0..dim-1 = x_t, dim..2dim-1 = x_{t-1} if keep_prev."""
        out = np.empty((self.n_particles, self.row_width))
        self._check(self._L.gpf_get_rows(self._h, _pd(out), out.size))
        return out

    @traces.setter
    def traces(self, rows):
        rows = np.ascontiguousarray(rows, np.float64)
        self._check(self._L.gpf_set_rows(self._h, _pd(rows), rows.size))

    def column(self, col: int) -> np.ndarray:
        out = np.empty(self.n_particles)
        self._check(self._L.gpf_get_column(self._h, int(col), _pd(out), out.size))
        return out

    def history_column(self, step: int, col: int) -> np.ndarray:
        """trace[step => col] of every current particle (step is 1-based like the Julia address t => :name)"""
        out = np.empty(self.n_particles)
        self._check(self._L.gpf_history_column(self._h, int(step), int(col), _pd(out), out.size))
        return out

    # -- measurement hooks
    def kernel_timing(self, kernel_id: int, enable: bool = True):
        self._check(self._L.gpf_kernel_timing(self._h, kernel_id, int(enable)))

    def kernel_time(self, kernel_id: int):
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._L.gpf_kernel_time(self._h, kernel_id, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def debug_math(self, which: int, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        b = a if b is None else np.ascontiguousarray(b, np.float64)
        o1, o2 = np.empty_like(a), np.empty_like(a)
        self._check(self._L.gpf_debug_math(self._h, which, _pd(a), _pd(b), a.size, _pd(o1), _pd(o2)))
        return o1, o2


class DeviceParticleFilterSubState(DeviceParticleFilterState):
    """ParticleFilterSubState (src/view.jl:16-22): aliases particles start, start+step, ... (count of them) of `source`; every pf_* function
    accepts it, with the reference's sub-state semantics (local parents, no log-ML update on resampling, weights reset to
    the block average, log_ml_estimate relative to the source's running estimate)."""

    def __init__(self, source: DeviceParticleFilterState, start: int, count: int, step: int = 1, index=None):
        self.source = source                     # keeps the parent alive
        self.start, self.step = int(start), int(step)
        self.index = index                       # state[idxs] over an arbitrary index vector (step == 0)
        self._L = source._L
        self.model, self.seed, self.keep_prev = source.model, source.seed, source.keep_prev
        self.n_particles, self.dim, self.row_width = int(count), source.dim, source.row_width
        self._h = C.c_void_p()
        if index is not None:
            st = self._L.gpf_view_create_indexed(source._h, index.ctypes.data_as(C.POINTER(C.c_int64)), int(count), C.byref(self._h))
        else:
            st = self._L.gpf_view_create_strided(source._h, int(start), int(step), int(count), C.byref(self._h))
        if st != _lib.OK:
            self._h = None
            raise ErrorException(self._L.gpf_last_error(source._h).decode())


ParticleFilterState = DeviceParticleFilterState
ParticleFilterSubState = DeviceParticleFilterSubState
ParticleFilterView = (DeviceParticleFilterState, DeviceParticleFilterSubState)      # src/view.jl:32-33 (Union; use with isinstance)


def _obs_vector(observations) -> np.ndarray:
    o = observations
    if type(o) is np.ndarray and o.dtype == np.float64 and o.ndim == 1 and o.flags.c_contiguous:
        return o                                   # (the usual call: a row of the observation matrix -- nothing to convert)
    return np.ascontiguousarray(np.atleast_1d(np.asarray(o, np.float64)))


# ----------------------------------------------------------------------------- the four operations
def pf_initialize(model: NativeModel, model_args: tuple, observations, *rest, seed: int = 1,
                  keep_prev: bool = False, device: int = 0, dynamic: bool = False, **kw) -> DeviceParticleFilterState:
    """src/initialize.jl:31-44.  `model_args` is accepted for signature parity; native models take their
    time-varying inputs through the per-step data vector `observations`. `dynamic` has no meaning for
    fixed-shape device rows and is ignored."""
    # pf_initialize(model, args, obs, n_particles) | (model, args, obs, strata, n_particles; layout) |
    # (model, args, obs, proposal, proposal_args, n_particles)
    layout = kw.pop("layout", "contiguous")                        # initialize.jl:66
    strata = None
    if len(rest) == 1:
        proposal, n_particles = None, rest[0]
    elif len(rest) == 2:
        proposal, strata, n_particles = None, _strata_values(model, rest[0], "initialize"), rest[1]
    elif len(rest) == 3:
        proposal, n_particles = rest[0], rest[2]
        _proposal_id(proposal)
    elif len(rest) == 4:                                            # (strata, proposal, proposal_args, n): initialize.jl:111-129
        strata, proposal, n_particles = _strata_values(model, rest[0], "initialize"), rest[1], rest[3]
        _proposal_id(proposal)
    else:
        raise TypeError("pf_initialize(model, model_args, observations, [strata,] [proposal, proposal_args,] n_particles)")
    state = DeviceParticleFilterState(model, n_particles, seed=seed, keep_prev=keep_prev, device=device, **kw)
    obs = _obs_vector(observations)
    if strata is not None and proposal is not None:
        state._check(state._L.gpf_initialize_strata_proposal(state._h, _pd(obs), obs.size, _pd(strata), strata.size, int(_layout_id(layout)),
                                                             _proposal_id(proposal)))
    elif strata is not None:
        state._check(state._L.gpf_initialize_strata(state._h, _pd(obs), obs.size, _pd(strata), strata.size, int(_layout_id(layout))))
    elif proposal is None:
        state._check(state._L.gpf_initialize(state._h, _pd(obs), obs.size))
    else:
        state._check(state._L.gpf_initialize_proposal(state._h, _pd(obs), obs.size, _proposal_id(proposal)))
    return state


def choiceproduct(*choices):
    """src/utils.jl:57-98: the strata as a list of choice maps {address: value}: `choiceproduct(("moving", [False, True]))`.
    A dict {address: values} or several (address, values) tuples give the Cartesian product."""
    import itertools
    if len(choices) == 1 and isinstance(choices[0], dict):
        choices = tuple(choices[0].items())
    return [dict(c) for c in itertools.product(*[[(addr, v) for v in vals] for addr, vals in choices])]


def _layout_id(layout) -> bool:
    if layout not in ("contiguous", "interleaved"):
        raise ValueError("layout must be 'contiguous' or 'interleaved'")
    return layout == "interleaved"


def _strata_values(model, strata, phase: str = "update") -> np.ndarray:
    """strata: an iterable of choice maps over the model's ONE discrete latent address (or of plain values); a model may
    stratify a different address at its first step (`phase` = "initialize") than later ("update")"""
    addr = model.info.get("strata_address")
    if isinstance(addr, dict):
        addr = addr[phase]
    if addr is None:
        raise ErrorException(f"model {model.name} has no discrete latent to stratify over")
    vals = []
    for st in strata:
        if isinstance(st, dict):
            if set(st.keys()) != {addr}:
                raise ErrorException(f"device strata constrain the address {addr!r} only, got {sorted(st.keys())}")
            st = st[addr]
        vals.append(float(st))
    if not 1 <= len(vals) <= 8:
        raise ErrorException("1..8 strata supported")
    return np.ascontiguousarray(vals, np.float64)


def pf_update(state: DeviceParticleFilterState, new_args: tuple, argdiffs: tuple, observations,
              proposal=None, proposal_args: tuple = (), *, layout: str = "interleaved"):
    """src/update.jl:12-25 (default proposal), :79-96 (custom proposal: log weight = model_score_diff -
    fwd_proposal_score, src/translate.jl:86-105) and :193-210 (stratified: the 5th argument is the strata).
    Returns `state`, like the reference (update.jl:24)."""
    obs = _obs_vector(observations)
    if proposal is None:
        # (an ESS-triggered loop has this call on its critical path -- the GPU idles between the ESS read and this launch: the address
        #  as an integer instead of a ctypes pointer object, 0.8 us instead of 2.3)
        st = state._L.gpf_update(state._h, obs.ctypes.data, obs.size)
        if st != _lib.OK:
            state._check(st)
    elif isinstance(proposal, _NativeKernel):
        state._check(state._L.gpf_update_proposal(state._h, _pd(obs), obs.size, _proposal_id(proposal)))
    elif isinstance(proposal, (list, tuple)) or hasattr(proposal, "__iter__"):
        strata = _strata_values(state.model, proposal)
        state._check(state._L.gpf_update_strata(state._h, _pd(obs), obs.size, _pd(strata), strata.size, int(_layout_id(layout))))
    else:
        _proposal_id(proposal)
    return state


def pf_step_ess(state: DeviceParticleFilterState, new_args: tuple, argdiffs: tuple, observations, *, ess_threshold: float = 0.5,
                method: str = "multinomial", rejuvenate=None, n_iters: int = 1, check="warn", sort_particles: bool = True) -> bool:
    """One iteration of the reference's README loop (README.md:66-77) in ONE call (gpf.h gpf_step_ess):

        if effective_sample_size(state) < ess_threshold * n_particles
            pf_resample!(state, method; check, sort_particles)
            pf_rejuvenate!(state, kern, (), n_iters; method = rejuvenate)      # rejuvenate: None | "move" | "reweight"
        end
        pf_update!(state, new_args, argdiffs, observations)

    The results are those of the separate calls, bit for bit; the steps that do not resample do not wait for the host's ESS decision
    (the propagate is enqueued speculatively behind the reduction, which leaves its verdict on the device).  Returns whether it resampled."""
    if method not in RESAMPLE_METHODS:
        raise ErrorException(f"Resampling method {method} not recognized.")
    if rejuvenate is not None and rejuvenate not in REJUVENATE_METHODS:
        raise ErrorException(f"Method {rejuvenate} not recognized.")
    if check not in (True, False, "warn"):
        raise ValueError("check must be True, 'warn' or False")
    check_id = 2 if check is True else (1 if check == "warn" else 0)
    obs = _obs_vector(observations)
    res, inv = C.c_int32(0), C.c_int32(0)
    st = state._L.gpf_step_ess(state._h, obs.ctypes.data, obs.size, float(ess_threshold), RESAMPLE_METHODS[method], int(sort_particles), check_id,
                               -1 if rejuvenate is None else REJUVENATE_METHODS[rejuvenate], int(n_iters), C.byref(res),
                               C.byref(inv) if check_id != 0 else None, None)
    if st != _lib.OK:
        raise ErrorException(state._L.gpf_last_error(state._h).decode())
    if check == "warn" and inv.value:
        warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")   # utils.jl:120-135
    return bool(res.value)


def _resample(state, method_id: int, priority_fn, check, sort_particles: bool):
    if check not in (True, False, "warn"):
        raise ValueError("check must be True, 'warn' or False")
    check_id = 2 if check is True else (1 if check == "warn" else 0)
    inv = C.c_int32(0)
    inv_ptr = C.byref(inv) if check_id != 0 else None          # check=false: fully asynchronous
    err_handle = state._h
    if (priority_fn is None and isinstance(state, DeviceParticleFilterSubState) and state.start == 0 and state.index is None
            and state.n_particles == state.source.n_particles and os.environ.get("GPF_VIEW_RESAMPLE") != "eager"):
        # pf_resample!(state[1:end], ...): the library resamples the whole filter with the sub-state semantics itself
        # (gpf_resample_local: same result, no eager gather, no copies through the view handle)
        err_handle = state.source._h
        st = state._L.gpf_resample_local(err_handle, method_id, int(sort_particles), check_id, inv_ptr)
    elif priority_fn is None or isinstance(priority_fn, Tempering):
        alpha = float("nan") if priority_fn is None else priority_fn.alpha
        st = state._L.gpf_resample(state._h, method_id, alpha, int(sort_particles), check_id, inv_ptr)
    else:
        lw = state.log_weights
        try:
            lp = np.ascontiguousarray(priority_fn(lw), np.float64)
            if lp.shape != lw.shape:
                raise TypeError
        except TypeError:
            lp = np.array([priority_fn(float(w)) for w in lw], np.float64)
        st = state._L.gpf_resample_with_priorities(state._h, method_id, _pd(lp), int(sort_particles), check_id, inv_ptr)
    if st != _lib.OK:
        raise ErrorException(state._L.gpf_last_error(err_handle).decode())     # error("Invalid weights."), ...
    if check == "warn" and inv.value:
        warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")   # utils.jl:120-135
    return state


def pf_multinomial_resample(state, *, priority_fn=None, check="warn"):
    """src/resample.jl:48-65"""
    return _resample(state, 0, priority_fn, check, True)


def pf_residual_resample(state, *, priority_fn=None, check="warn"):
    """src/resample.jl:85-120"""
    return _resample(state, 1, priority_fn, check, True)


def pf_stratified_resample(state, *, priority_fn=None, check="warn", sort_particles: bool = True):
    """src/resample.jl:143-175"""
    return _resample(state, 2, priority_fn, check, sort_particles)


def pf_multinomial_sorted_resample(state, *, priority_fn=None, check="warn"):
    """OPT-IN extension, not a reference method: pf_multinomial_resample! (src/resample.jl:48-65) with the N uniforms drawn already sorted
    (uniform spacings in exact integers): offspring counts ~ Multinomial(N, w) as for "multinomial", `state.parents`
    non-decreasing instead of an i.i.d. sequence (src/resample.jl:59) -- the ancestor search is a streaming merge, the row gather reads
    ascending rows."""
    return _resample(state, RESAMPLE_METHODS["multinomial_sorted"], priority_fn, check, True)


def pf_resample(state, method: str = "multinomial", **kwargs):
    """src/resample.jl:19-30 (+ the opt-in "multinomial_sorted")"""
    if method == "multinomial_sorted":
        return pf_multinomial_sorted_resample(state, **kwargs)
    if method == "multinomial":
        return pf_multinomial_resample(state, **kwargs)
    if method == "residual":
        return pf_residual_resample(state, **kwargs)
    if method == "stratified":
        return pf_stratified_resample(state, **kwargs)
    raise ErrorException(f"Resampling method {method} not recognized.")


def pf_resample_blocks(state, block_size: int, method: str = "multinomial", *, priority_fn=None, ess_frac=None, sort_particles: bool = True, check="warn"):
    """Many small filters in one state: the batched form of

        for b in blocks:                                   # consecutive blocks of block_size particles
            if ess_frac is None or get_ess(state[b]) < ess_frac * len(b):
                pf_resample(state[b], method, sort_particles=..., check=...)

    (sub-states: src/view.jl:16-48, src/resample.jl:185-187,205-218; the README loop README.md:60-79 per block) in ONE kernel
    launch (gpf.h gpf_resample_blocks): one workgroup per block, everything out of LDS, the ESS test on the device.  Every block's
    result is bit-identical to the loop above run through views.  Returns the number of blocks that resampled."""
    if method not in RESAMPLE_METHODS:
        raise ErrorException(f"Resampling method {method} not recognized.")
    if check not in (True, False, "warn"):
        raise ValueError("check must be True, 'warn' or False")
    if isinstance(state, DeviceParticleFilterSubState):
        raise ErrorException("pf_resample_blocks works on the whole filter")
    if priority_fn is not None and not isinstance(priority_fn, Tempering):
        raise ErrorException("block-wise resampling takes priority_fn = None or Tempering(alpha) (w -> alpha w)")
    check_id = 2 if check is True else (1 if check == "warn" else 0)
    inv, cnt = C.c_int32(0), C.c_int64(0)
    st = state._L.gpf_resample_blocks(state._h, RESAMPLE_METHODS[method], int(block_size),
                                      float("nan") if priority_fn is None else float(priority_fn.alpha), int(sort_particles),
                                      float("nan") if ess_frac is None else float(ess_frac), check_id, C.byref(inv), C.byref(cnt))
    state._n_blocks_last = (state.n_particles + int(block_size) - 1) // int(block_size) if int(block_size) > 0 else 0
    if st != _lib.OK:
        raise ErrorException(state._L.gpf_last_error(state._h).decode())
    if check == "warn" and inv.value:
        warnings.warn("Invalid weights (all -Inf or zero) in some block: resampled with uniform weights.")
    return int(cnt.value)


def _block_obs(state, observations, block_size: int) -> np.ndarray:
    nb = (state.n_particles + int(block_size) - 1) // int(block_size)
    obs = np.ascontiguousarray(observations, np.float64)
    if obs.ndim == 1:
        obs = obs.reshape(nb, -1)
    if obs.ndim != 2 or obs.shape[0] != nb:
        raise ErrorException(f"one observation vector per block expected: {nb} rows, got an array of shape {obs.shape}")
    return obs


def pf_initialize_blocks(model: NativeModel, model_args: tuple, observations, n_particles: int, block_size: int, *, seed: int = 1,
                         keep_prev: bool = False, device: int = 0, strata=None, layout: str = "contiguous"):
    """many small filters in one state, each with its own data: block b (block_size consecutive particles) is initialised with
    observations[b] -- the batched form of per-view initialisation (gpf.h gpf_initialize_blocks).  strata: every block is initialised
    stratified by itself (src/initialize.jl:92-109 per sub-state, gpf.h gpf_initialize_blocks_strata), the same strata for all blocks"""
    state = DeviceParticleFilterState(model, n_particles, seed=seed, keep_prev=keep_prev, device=device)
    obs = _block_obs(state, observations, block_size)
    if strata is not None:
        v = _strata_values(model, strata, "initialize")
        state._check(state._L.gpf_initialize_blocks_strata(state._h, _pd(obs), obs.shape[1], int(block_size), _pd(v), v.size, int(_layout_id(layout))))
        return state
    state._check(state._L.gpf_initialize_blocks(state._h, _pd(obs), obs.shape[1], int(block_size)))
    return state


def pf_update_blocks(state, new_args: tuple, argdiffs: tuple, observations, block_size: int, proposals=None, *, strata=None, layout: str = "interleaved"):
    """for b in blocks: pf_update!(state[b], new_args, argdiffs, observations[b]) (per-view updates, test/update.jl:179-189) in one launch.
    proposals: one entry per block, None (default proposal, update.jl:12-25) or the model's native proposal (update.jl:79-96) -- "Update with
    different proposals per view" in one launch (gpf.h gpf_update_blocks_proposal)"""
    obs = _block_obs(state, observations, block_size)
    if strata is not None:                                       # every block stratified by itself (src/update.jl:193-210 per sub-state)
        if proposals is not None:
            raise ErrorException("block-wise updates take strata or per-block proposals, not both")
        v = _strata_values(state.model, strata)
        state._check(state._L.gpf_update_blocks_strata(state._h, _pd(obs), obs.shape[1], int(block_size), _pd(v), v.size, int(_layout_id(layout))))
        return state
    if proposals is None:
        state._check(state._L.gpf_update_blocks(state._h, _pd(obs), obs.shape[1], int(block_size)))
        return state
    if len(proposals) != obs.shape[0]:
        raise ErrorException(f"one proposal (or None) per block expected: {obs.shape[0]}, got {len(proposals)}")
    native = [q for q in proposals if q is not None]
    pid = _proposal_id(native[0]) if native else 1
    if any(_proposal_id(q) != pid for q in native):
        raise ErrorException("the blocks' native proposals must be the same one")
    flags = np.ascontiguousarray([0 if q is None else 1 for q in proposals], np.int32)
    state._check(state._L.gpf_update_blocks_proposal(state._h, _pd(obs), obs.shape[1], int(block_size), flags.ctypes.data_as(C.POINTER(C.c_int32)), pid))
    return state


def pf_rejuvenate_blocks(state, kern=None, kern_args: tuple = (), n_iters: int = 1, *, method: str = "move", only_resampled: bool = False,
                         count: bool = False):
    """for b in blocks: pf_rejuvenate!(state[b], ...) with the block's own latest observation; only_resampled: only the blocks the last
    pf_resample_blocks resampled (the README loop rejuvenates inside its `if`)"""
    if method not in REJUVENATE_METHODS:
        raise ErrorException(f"Method {method} not recognized.")
    acc = C.c_uint64(0)
    state._check(state._L.gpf_rejuvenate_blocks(state._h, REJUVENATE_METHODS[method], int(n_iters), int(only_resampled), C.byref(acc) if count else None))
    return int(acc.value) if count else state


def block_resampled(state) -> np.ndarray:
    """which blocks the last pf_resample_blocks resampled (bool per block)"""
    out = np.zeros(getattr(state, "_n_blocks_last", 0), np.int32)
    state._check(state._L.gpf_block_resampled(state._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out.astype(bool)


def block_stats(state, block_size: int):
    """(effective_sample_size(state[b]), log_ml_estimate(state[b])) of every block, src/utils.jl:163-178, in one launch"""
    nb = (state.n_particles + int(block_size) - 1) // int(block_size)
    ess, lml = np.empty(nb), np.empty(nb)
    state._check(state._L.gpf_block_stats(state._h, int(block_size), _pd(ess), _pd(lml)))
    return ess, lml


def _rejuvenate(state, method_id: int, n_iters: int, want_count: bool):
    acc = C.c_uint64(0)
    st = state._L.gpf_rejuvenate(state._h, method_id, int(n_iters), C.byref(acc) if want_count else None)
    state._check(st)
    state.n_accepted = acc.value if want_count else None
    return state


def pf_move_accept(state, kern=mh, kern_args: tuple = (), n_iters: int = 1, *, count: bool = False):
    """src/rejuvenate.jl:40-53 with the native mh kernel; kern_args = () -> Gen.mh(trace, selection) on the current step's latent,
    kern_args = (proposal[, proposal_args]) with a MoveProposal -> Gen.mh(trace, proposal, proposal_args): accept iff
    log(rand()) < weight - fwd_score + bwd_score"""
    if kern is not mh:
        raise ErrorException("device states support the native `mh` kernel only (arbitrary Julia/Python callables are out of scope)")
    if kern_args and isinstance(kern_args[0], MoveProposal):
        mp = kern_args[0]
        q = np.asarray(mp.params, np.float64)
        acc = C.c_uint64(0)
        st = state._L.gpf_rejuvenate_with_proposal(state._h, 0, mp.proposal_id, _pd(q) if q.size else None, int(q.size), int(n_iters),
                                                   C.byref(acc) if count else None)
        state._check(st)
        state.n_accepted = acc.value if count else None
        return state
    return _rejuvenate(state, 0, n_iters, count)


class MoveProposal:
    """a native proposal for move_reweight(trace, proposal, proposal_args) (src/rejuvenate.jl:134-148):
    `locally_optimal_move` (lgssm2: q = p(x_t | x_{t-1}, y_t)), `outlier_propose(q)` (line_model: the current step's outlier ~ bernoulli(q),
    the proposal of test/rejuvenate.jl:19-27)"""

    def __init__(self, name, proposal_id, params=()):
        self.name, self.proposal_id, self.params = name, proposal_id, tuple(float(p) for p in params)

    def __repr__(self):
        return f"<native move proposal {self.name}{self.params}>"


locally_optimal_move = MoveProposal("locally_optimal", 1)


def outlier_propose(q: float) -> MoveProposal:
    import math
    return MoveProposal("outlier_propose", 2, (q, math.log(q) if q > 0 else -math.inf, math.log1p(-q) if q < 1 else -math.inf))


def pf_move_reweight(state, kern=move_reweight, kern_args: tuple = (), n_iters: int = 1, *, count: bool = False):
    """src/rejuvenate.jl:74-90 with the native move_reweight kernel; kern_args = () -> the selection variant (rejuvenate.jl:125-132),
    kern_args = (proposal[, proposal_args]) with a MoveProposal -> the proposal variant (rejuvenate.jl:134-148)"""
    if kern is not move_reweight:
        raise ErrorException("device states support the native `move_reweight` kernel only")
    if kern_args and isinstance(kern_args[0], MoveProposal):
        mp = kern_args[0]
        q = np.asarray(mp.params, np.float64)
        st = state._L.gpf_rejuvenate_proposal(state._h, mp.proposal_id, _pd(q) if q.size else None, int(q.size), int(n_iters))
        state._check(st)
        state.n_accepted = state.n_particles * int(n_iters)
        return (state, state.n_accepted) if count else state
    return _rejuvenate(state, 1, n_iters, count)


def pf_rejuvenate(state, kern=None, kern_args: tuple = (), n_iters: int = 1, *, method: str = "move", **kwargs):
    """src/rejuvenate.jl:18-27"""
    if method == "move":
        return pf_move_accept(state, mh if kern is None else kern, kern_args, n_iters, **kwargs)
    if method == "reweight":
        return pf_move_reweight(state, move_reweight if kern is None else kern, kern_args, n_iters, **kwargs)
    raise ErrorException(f"Method {method} not recognized.")


# ----------------------------------------------------------------------------- resize family (src/resize.jl)
def _refresh_count(state):
    n = C.c_int64()
    state._check(state._L.gpf_n_particles(state._h, C.byref(n)))
    state.n_particles = n.value


def _resize(state, n_particles: int, method_id: int, priority_fn, check):
    if check not in (True, False, "warn"):
        raise ValueError("check must be True, 'warn' or False")
    if priority_fn is not None and not isinstance(priority_fn, Tempering):
        raise ErrorException("device resize supports priority_fn = nothing or Tempering(alpha)")
    check_id = 2 if check is True else (1 if check == "warn" else 0)
    inv = C.c_int32(0)
    alpha = float("nan") if priority_fn is None else priority_fn.alpha
    st = state._L.gpf_resize(state._h, int(n_particles), method_id, alpha, check_id, C.byref(inv) if check_id else None)
    if st == _lib.ERR_INVALID_WEIGHTS:
        raise ErrorException(state._L.gpf_last_error(state._h).decode())
    state._check(st)
    _refresh_count(state)
    if check == "warn" and inv.value:
        warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")
    return state


def pf_multinomial_resize(state, n_particles: int, *, priority_fn=None, check="warn"):
    """src/resize.jl:46-68"""
    return _resize(state, n_particles, 0, priority_fn, check)


def pf_residual_resize(state, n_particles: int, *, priority_fn=None, check="warn"):
    """src/resize.jl:87-124"""
    return _resize(state, n_particles, 1, priority_fn, check)


def pf_optimal_resize(state, n_particles: int, *, check="warn"):
    """src/resize.jl:149-200 (Fearnhead & Clifford): n_particles must not exceed the current count"""
    return _resize(state, n_particles, 3, None, check)


def pf_resize(state, n_particles: int, method: str = "multinomial", **kwargs):
    """src/resize.jl:16-28"""
    if method == "multinomial":
        return pf_multinomial_resize(state, n_particles, **kwargs)
    if method == "residual":
        return pf_residual_resize(state, n_particles, **kwargs)
    if method == "optimal":
        return pf_optimal_resize(state, n_particles, **kwargs)
    raise ErrorException(f"Resampling method {method} not recognized.")


def pf_replicate(state, n_replicates: int, *, layout: str = "contiguous"):
    """src/resize.jl:236-244"""
    state._check(state._L.gpf_replicate(state._h, int(n_replicates), int(layout != "contiguous")))
    _refresh_count(state)
    return state


def pf_dereplicate(state, n_replicates: int, *, layout: str = "contiguous", method: str = "keepfirst"):
    """src/resize.jl:267-297"""
    if method not in ("keepfirst", "sample"):
        raise ErrorException(f"Method {method} not recognized.")
    state._check(state._L.gpf_dereplicate(state._h, int(n_replicates), int(layout != "contiguous"), int(method == "sample")))
    _refresh_count(state)
    return state


# ----------------------------------------------------------------------------- summaries (src/utils.jl)
def effective_sample_size(state) -> float:
    out = C.c_double()
    state._check(state._L.gpf_effective_sample_size(state._h, C.byref(out)))
    return out.value


get_ess = effective_sample_size


def log_ml_estimate(state) -> float:
    out = C.c_double()
    state._check(state._L.gpf_log_ml_estimate(state._h, C.byref(out)))
    return out.value


get_lml_est = log_ml_estimate


def get_log_weights(state) -> np.ndarray:
    return state.log_weights


def get_log_norm_weights(state) -> np.ndarray:
    out = np.empty(state.n_particles)
    state._check(state._L.gpf_get_log_norm_weights(state._h, _pd(out), out.size))
    return out


def get_norm_weights(state) -> np.ndarray:
    out = np.empty(state.n_particles)
    state._check(state._L.gpf_get_norm_weights(state._h, _pd(out), out.size))
    return out


def get_traces(state) -> np.ndarray:
    return state.traces


def sample_unweighted_traces(state, n_samples: int, return_indices: bool = False):
    """Gen.sample_unweighted_traces(state, n_samples) (src/utils.jl:189-194 extends it to sub-states): n i.i.d. particles drawn
    with probability proportional to their weights; the filter itself is left untouched."""
    rows = np.empty((int(n_samples), state.row_width))
    idx = np.empty(int(n_samples), np.int64)
    state._check(state._L.gpf_sample_unweighted(state._h, int(n_samples), _pd(rows), idx.ctypes.data_as(C.POINTER(C.c_int64))))
    return (rows, idx) if return_indices else rows


# ----------------------------------------------------------------------------- statistics (src/statistics.jl)
def _addr_values(state, addr) -> np.ndarray:
    """the values at one address over all particles: a column of the current step, or (t, column) for a past choice"""
    return state.history_column(int(addr[0]), int(addr[1])) if isinstance(addr, tuple) else state.column(int(addr))


def _functional(f, state, addrs):
    if not addrs:
        raise ErrorException("native models have no return value: give at least one address")
    vals = [_addr_values(state, a) for a in addrs]
    try:
        fv = np.asarray(f(*vals), np.float64)
        if fv.shape != vals[0].shape:
            raise TypeError
    except (TypeError, ValueError):
        fv = np.array([f(*xs) for xs in zip(*vals)], np.float64)              # scalar closure: broadcast(f, ...)
    return fv


def mean(*args) -> float:
    """mean(state, addr), src/statistics.jl:13-14.  addr = column of the current-step latent, or a pair
    (t, column) for a PAST choice `t => column` (reference README.md:97; needs pf_initialize(..., history=T)).
    mean(f, state, addrs...), src/statistics.jl:28-35: weighted mean of a host closure of the values at the addresses
    (the values and the normalised weights are read back; f may be vectorised over NumPy arrays or scalar)."""
    if callable(args[0]):
        f, state, addrs = args[0], args[1], args[2:]
        return float(np.sum(get_norm_weights(state) * _functional(f, state, addrs)))
    state, addr = args
    out = C.c_double()
    if isinstance(addr, tuple):
        state._check(state._L.gpf_history_mean(state._h, int(addr[0]), int(addr[1]), C.byref(out)))
    else:
        state._check(state._L.gpf_mean(state._h, int(addr), C.byref(out)))
    return out.value


def var(*args) -> float:
    """var(state, addr), population form (src/statistics.jl:48-50); addr as in mean().
    var(f, state, addrs...), src/statistics.jl:65-76: var(fvals, Weights(w), corrected=false)."""
    if callable(args[0]):
        f, state, addrs = args[0], args[1], args[2:]
        w, fv = get_norm_weights(state), _functional(f, state, addrs)
        mu = np.sum(w * fv) / np.sum(w)
        return float(np.sum(w * (fv - mu) ** 2) / np.sum(w))
    state, addr = args
    out = C.c_double()
    if isinstance(addr, tuple):
        state._check(state._L.gpf_history_var(state._h, int(addr[0]), int(addr[1]), C.byref(out)))
    else:
        state._check(state._L.gpf_var(state._h, int(addr), C.byref(out)))
    return out.value


def proportionmap(state, addr, max_values: int = 256) -> dict:
    """proportionmap(state, addr), src/statistics.jl:91-101: {value: sum of normalised weights of the particles holding it}
    for a discrete-valued column (addr = column, or (t, column) for a past choice).  The distinct values are read from
    the column; each proportion is a weighted reduction on the device."""
    step, col = (int(addr[0]), int(addr[1])) if isinstance(addr, tuple) else (0, int(addr))
    vals = np.unique(state.history_column(step, col) if step else state.column(col))
    if vals.size > max_values:
        raise ErrorException(f"proportionmap: {vals.size} distinct values; the column does not look discrete")
    out = {}
    for v in vals:
        p = C.c_double()
        state._check(state._L.gpf_proportion(state._h, step, col, float(v), C.byref(p)))
        out[float(v)] = p.value
    return out
