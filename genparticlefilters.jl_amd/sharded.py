"""Multi-GPU particle filter: one process per GPU, particles sharded by contiguous global index,
collectives through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests).  DESIGN.md §6.

The reference has no distributed code; what is restated here is the SAME resampling
spec as the single-GPU path (src/resample.jl:48-120,143-175) over a global weight CDF:

    phase 1   local max / flags                      -> all-gather (2 doubles per rank)
    phase 2   local fixed-point scan under the GLOBAL max -> all-gather of the shard totals
    (2b)      residual: copy-count and residual-weight scans -> all-gather of their totals
    phase 3   RNG counters are keyed by the GLOBAL slot id, so every shard evaluates the target of EVERY output slot
              itself (owner = first shard whose inclusive total exceeds it) and learns, without any message, which
              slots draw from its own particles and what it will receive from whom
              -> one host sync (the all-to-all split sizes); no request traffic, no count exchange
    phase 4   owners look the ancestors up in their local CDF, gather the rows and pack them by destination
              -> ONE all-to-all of [row | slot | ancestor id]
    phase 5   scatter by slot, log-weights = 0, log-ML estimate += logsumexp - log N

Every phase is one C-ABI call (gpf_shard_*) and at most one collective; the host only moves buffers.

Because weights are exact integers and RNG counters are keyed by GLOBAL slot id, the ancestors are
bit-identical to the single-GPU run for any number of shards.

The collectives and their split sizes are plain torch.distributed code, the same on CPU and GPU; the
arithmetic and the routing live behind a small backend interface: `HipShardBackend` (the
product: libgpf_hip.so through the C ABI) -- the tests inject a CPU backend built on the oracle to
exercise the collectives with gloo.  Restrictions: priority_fn = nothing or Tempering(alpha) (the latter in the library engine only, and not together
with sort_particles = true, where every rank sorts the gathered weights: gpf_shard_resample_sorted).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .api import DeviceParticleFilterState, ErrorException, _obs_vector, _pd

RESAMPLE_METHODS = {"multinomial": 0, "residual": 1, "stratified": 2,
                    # the opt-in sorted form of the multinomial resampler (gpf.h GPF_RESAMPLE_MULTINOMIAL_SORTED): ascending
                    # targets like the strata's, so every shard serves ONE slot range and the exchange is boundary slabs
                    "multinomial_sorted": 4}
# tests: issue the real collectives even in a 1-rank process group (exercises the RCCL call path on a 1-GPU box)
_FORCE_COLLECTIVES = os.environ.get("GPF_SHARD_FORCE_COLLECTIVES") == "1"
SPACE_COUNTS = 1 << 62


def shard_range(n_global: int, rank: int, world: int):
    """Contiguous ranges; the first (n_global % world) ranks hold one extra particle."""
    base, extra = divmod(n_global, world)
    n = base + (1 if rank < extra else 0)
    gid0 = rank * base + min(rank, extra)
    return gid0, n


class HipShardBackend:
    """Shard arithmetic on the GPU through the C ABI (include/gpf.h, gpf_shard_*)."""

    def __init__(self, model, n_global, gid0, n_local, seed, keep_prev, device):
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # one explicit HIP stream shared by the kernels (through the C ABI) and torch (routing ops + collectives):
        # torch's default stream is the NULL stream, which the library would replace by a stream of its own.
        self.stream = torch.cuda.Stream(self.device)
        torch.cuda.set_stream(self.stream)
        stream = self.stream.cuda_stream
        self.state = DeviceParticleFilterState(model, n_local, seed=seed, keep_prev=keep_prev, device=device,
                                               n_global=n_global, gid0=gid0, stream=stream)
        self.L, self.h = self.state._L, self.state._h
        self.n, self.W = n_local, self.state.row_width
        self.lib_comm = False

    def _ck(self, st):
        self.state._check(st)

    def initialize(self, obs):
        self._ck(self.L.gpf_initialize(self.h, _pd(obs), obs.size))

    def update(self, obs):
        self._ck(self.L.gpf_update(self.h, _pd(obs), obs.size))

    def rejuvenate(self, method_id, n_iters):
        self._ck(self.L.gpf_rejuvenate(self.h, method_id, n_iters, None))

    def weight_max(self):
        out = torch.empty(2, dtype=torch.float64, device=self.device)
        self._ck(self.L.gpf_shard_weight_max(self.h, out.data_ptr()))
        return out

    def weight_scan(self, mf_all, want_q=True):
        out = torch.empty(5, dtype=torch.int64, device=self.device)      # [1:5] stay undefined without want_q: nobody reads them
        self._ck(self.L.gpf_shard_weight_scan(self.h, mf_all.data_ptr(), mf_all.shape[0], int(want_q), out.data_ptr()))
        return out

    def scan_flags(self) -> int:
        """validity flags of the global weights, as soon as the weight scan has started (no stream synchronisation)"""
        out = C.c_int32()
        self._ck(self.L.gpf_shard_flags(self.h, C.byref(out)))
        return out.value

    def residual_scan(self, tot_all):
        out = torch.empty(2, dtype=torch.int64, device=self.device)
        self._ck(self.L.gpf_shard_residual_scan(self.h, tot_all.data_ptr(), tot_all.shape[0], out.data_ptr()))
        return out

    def push_count(self, method_id, tot_all, cr_all, me, bounds):
        G = tot_all.shape[0]
        self._bounds = (C.c_int64 * (G + 1))(*bounds)
        self._ck(self.L.gpf_shard_push_count(self.h, method_id, tot_all.data_ptr(), cr_all.data_ptr() if cr_all is not None else None, G, me,
                                             self._bounds))

    def counts(self, G):
        """entries sent to each shard | received from each shard; synchronises the stream"""
        out = (C.c_int64 * (2 * G))()
        self._ck(self.L.gpf_shard_counts(self.h, G, out))
        return list(out)

    def push(self, method_id, tot_all, cr_all, me, bounds, capacity):
        """packs at most `capacity` entries (the caller checks the counts afterwards and calls again if they did not fit)"""
        G = tot_all.shape[0]
        if getattr(self, "_sendbuf", None) is None or self._sendbuf.shape[0] < capacity:
            self._sendbuf = torch.empty((capacity, self.W + 1), dtype=torch.float64, device=self.device)
        self._ck(self.L.gpf_shard_push(self.h, method_id, tot_all.data_ptr(), cr_all.data_ptr() if cr_all is not None else None, G, me,
                                       self._bounds, capacity, self._sendbuf.data_ptr() if capacity else None))
        return self._sendbuf

    # ---- stratified with sort_particles = true, phase by phase (gpf_shard_sorted_count / _push: the planner on the gathered log-weights)
    def log_weights_tensor(self):
        return torch.from_numpy(np.ascontiguousarray(self.state.log_weights, np.float64)).to(self.device)

    def sorted_count(self, lw_all, me, bounds):
        G = len(bounds) - 1
        self._lw_all = lw_all.contiguous()                      # alive until the stream has consumed it
        self._ck(self.L.gpf_shard_sorted_count(self.h, self._lw_all.data_ptr(), G, me))

    def sorted_push(self, me, bounds, capacity):
        G = len(bounds) - 1
        if getattr(self, "_sendbuf", None) is None or self._sendbuf.shape[0] < capacity:
            self._sendbuf = torch.empty((capacity, self.W + 1), dtype=torch.float64, device=self.device)
        self._ck(self.L.gpf_shard_sorted_push(self.h, G, me, capacity, self._sendbuf.data_ptr() if capacity else None))
        return self._sendbuf

    def commit(self, packed, mf_all, tot_all):
        packed = packed.contiguous()
        self._ck(self.L.gpf_shard_commit(self.h, packed.data_ptr(), packed.shape[0], mf_all.data_ptr(), tot_all.data_ptr(), tot_all.shape[0]))
        self._keep = (packed, mf_all, tot_all)             # alive until the stream has consumed them

    # ---- the library's own communicator: the whole resample behind one C-ABI call (gpf_shard_resample)
    def comm_create(self, rank: int, world: int, group=None):
        """RCCL communicator owned by libgpf (gpf_comm_create).  The 128-byte id is made by rank 0 and handed round with
        torch.distributed's object broadcast -- any channel would do, the library only needs the bytes."""
        idbuf = None
        if world > 1 or _FORCE_COLLECTIVES:
            blob = [None]
            if rank == 0:
                raw = (C.c_char * 128)()
                try:
                    self._ck(self.L.gpf_comm_unique_id(C.cast(raw, C.c_void_p)))
                    blob = [bytes(raw)]
                except ErrorException as e:                      # every rank must learn it, or the others wait for the id forever
                    blob = [("error", str(e))]
            if world > 1:
                dist.broadcast_object_list(blob, src=0, group=group)
            if isinstance(blob[0], tuple):
                raise ErrorException(f"gpf_comm_unique_id failed on rank 0: {blob[0][1]}")
            idbuf = (C.c_char * 128).from_buffer_copy(blob[0])
        ok, err = 1, ""
        try:
            self._ck(self.L.gpf_comm_create(self.h, C.cast(idbuf, C.c_void_p) if idbuf is not None else None, rank, world))
        except ErrorException as e:
            ok, err = 0, str(e)
        if world > 1:                                            # all ranks use the library engine, or none does
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if not int(flag.item()) and ok:
                err = "gpf_comm_create failed on another rank"
            ok = int(flag.item())
        if not ok:
            raise ErrorException(err)
        self.lib_comm = True
        self._rccl_world = world if idbuf is not None else 0      # no id: a single shard without any RCCL communicator

    def summary_mode(self) -> str:
        """how the small summaries of a sharded resample travel: "mailbox" (peer stores from the producing kernels, gpf.h
        gpf_comm_summary_mode), "rccl" (all-gathers issued by the library), "torch.distributed" (python engine), "none" (one shard)"""
        if not self.lib_comm:
            return "torch.distributed"
        mb = C.c_int32(0)
        self._ck(self.L.gpf_comm_summary_mode(self.h, C.byref(mb)))
        return "mailbox" if mb.value else ("rccl" if self.comm_world() else "none")

    def set_plan(self, plan: str) -> None:
        """exchange plan of the i.i.d. resamplers in the library engine (gpf.h gpf_comm_set_plan): "push" (every shard evaluates all
        n_global targets, one row exchange) or "pull" (own targets only, requests out and rows back); the same bits either way.
        Every rank must choose the same plan."""
        if not self.lib_comm:
            raise ErrorException("the exchange plan is selectable in the library engine only (python engine: push)")
        if plan not in ("push", "pull"):
            raise ErrorException(f"exchange plan {plan!r}: push or pull")
        self._ck(self.L.gpf_comm_set_plan(self.h, 1 if plan == "pull" else 0))

    def plan(self) -> str:
        if not self.lib_comm:
            return "push"
        p = C.c_int32(0)
        self._ck(self.L.gpf_comm_plan(self.h, C.byref(p)))
        return "pull" if p.value else "push"

    def set_exchange(self, mode: str) -> None:
        """how the rows of the resamplers with ascending targets (stratified, multinomial_sorted) cross shards (gpf.h gpf_comm_set_exchange): "p2p" = peer
        stores into the destination ranks' slot-addressed receive windows, no host wait and no ncclGroup; "rccl" = packed entries through grouped
        ncclSend / ncclRecv; "p2p_all" = the i.i.d. resamplers' rows through the windows too (opt-in, bandwidth-bound).  The same bits either way; every rank
        must choose the same mode."""
        if not self.lib_comm:
            raise ErrorException("the exchange mode is selectable in the library engine only")
        if mode not in ("p2p", "rccl", "p2p_all"):
            raise ErrorException(f"exchange mode {mode!r}: p2p, p2p_all or rccl")
        self._ck(self.L.gpf_comm_set_exchange(self.h, {"rccl": 0, "p2p": 1, "p2p_all": 2}[mode]))

    def exchange(self) -> str:
        if not self.lib_comm:
            return "torch.distributed"
        m = C.c_int32(0)
        self._ck(self.L.gpf_comm_exchange(self.h, C.byref(m)))
        return {0: "rccl", 1: "p2p", 2: "p2p_all"}[m.value]

    PHASES = ("summaries", "plan", "pack", "host_wait_counts", "exchange", "commit_propagate")

    def phase_timing(self, on: bool) -> None:
        """gpf_phase_timing: events at the phase boundaries of gpf_shard_resample and of the gpf_update that commits it (library engine)"""
        self._ck(self.L.gpf_phase_timing(self.h, int(bool(on))))

    def phase_times(self) -> dict:
        """microseconds PER RESAMPLE of every phase since phase_timing(True) (gpf.h gpf_phase_times; GPU timeline except host_wait_counts)"""
        us = (C.c_double * 6)()
        n = C.c_int64(0)
        self._ck(self.L.gpf_phase_times(self.h, us, C.byref(n)))
        k = max(int(n.value), 1)
        out = {name: round(float(us[i]) / k, 2) for i, name in enumerate(self.PHASES)}
        out["resamples"] = int(n.value)
        return out

    def calibrate(self, entries: int, reps: int = 20) -> dict:
        """gpf_comm_calibrate (collective): the grouped send / receive of `entries` packed entries with every peer and the mailbox round, timed on this machine"""
        out = (C.c_double * 4)()
        self._ck(self.L.gpf_comm_calibrate(self.h, int(entries), int(reps), out))
        return {"entries_per_peer": int(entries), "reps": int(reps), "exchange_us": round(out[0], 2), "link_GBps_measured": round(out[1], 2),
                "mailbox_round_us": round(out[2], 3), "mailbox_rtt_us": round(2 * out[2], 3), "empty_launch_us": round(out[3], 2)}

    def traffic(self, reset: bool = False):
        """(calls, entries sent to other ranks, entries received from other ranks, bytes per entry) of the library engine's resamples so far"""
        out = (C.c_int64 * 4)()
        self._ck(self.L.gpf_comm_traffic(self.h, out, int(reset)))
        return tuple(int(x) for x in out)

    def comm_world(self) -> int:
        """ranks of the library's RCCL communicator (0: none)"""
        return getattr(self, "_rccl_world", 0)

    def shard_resample(self, method_id: int, check, alpha=None, sort_particles=False) -> bool:
        """pf_resample!(state, method; priority_fn, check) over all shards, collectives issued by the library; returns `invalid`.
        alpha: priority_fn = w -> alpha w (gpf_shard_resample_tempered), None: priority_fn = nothing; sort_particles: the stratified resampler's strata over
        the particles in descending weight order (gpf_shard_resample_sorted: the replicated plan)"""
        chk = 2 if check is True else (1 if check == "warn" else 0)
        inv = C.c_int32(0)
        if sort_particles:
            self._ck(self.L.gpf_shard_resample_sorted(self.h, chk, C.byref(inv) if chk else None))
        elif alpha is None:
            self._ck(self.L.gpf_shard_resample(self.h, method_id, chk, C.byref(inv) if chk else None))
        else:
            self._ck(self.L.gpf_shard_resample_tempered(self.h, method_id, float(alpha), chk, C.byref(inv) if chk else None))
        return bool(inv.value)

    def shard_step_ess(self, obs, ess_frac: float, method_id: int, check, rejuv_id: int, n_iters: int):
        """gpf_shard_step_ess: one README-loop iteration on this rank's shard; returns (resampled, invalid)"""
        chk = 2 if check is True else (1 if check == "warn" else 0)
        res, inv = C.c_int32(0), C.c_int32(0)
        self._ck(self.L.gpf_shard_step_ess(self.h, obs.ctypes.data, obs.size, float(ess_frac), method_id, chk, rejuv_id, int(n_iters),
                                           C.byref(res), C.byref(inv) if chk else None))
        return bool(res.value), bool(inv.value)

    def shard_ess(self) -> float:
        out = C.c_double()
        self._ck(self.L.gpf_shard_effective_sample_size(self.h, C.byref(out)))
        return out.value

    def shard_lml(self) -> float:
        out = C.c_double()
        self._ck(self.L.gpf_shard_log_ml_estimate(self.h, C.byref(out)))
        return out.value

    def local_resample(self, method, priority_fn, check, sort_particles):
        """resample THIS shard's particles among themselves with the reference's sub-state semantics: a view of the whole shard"""
        from . import api
        if priority_fn is None:
            # the library's own whole-shard sub-state resample: same result as the view below, without its eager gather and copies
            if check not in (True, False, "warn"):
                raise ValueError("check must be True, 'warn' or False")
            check_id = 2 if check is True else (1 if check == "warn" else 0)
            inv = C.c_int32(0)
            st = self.L.gpf_resample_local(self.h, RESAMPLE_METHODS[method], int(bool(sort_particles)), check_id,
                                           C.byref(inv) if check_id else None)
            if st == _lib.ERR_INVALID_WEIGHTS:
                raise ErrorException(self.L.gpf_last_error(self.h).decode())
            self._ck(st)
            if check == "warn" and inv.value:
                import warnings
                warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")
            return
        if getattr(self, "_view", None) is None:
            self._view = self.state[0:self.n]
        kw = {"sort_particles": sort_particles} if method == "stratified" else {}      # only the stratified resampler reads it
        api.pf_resample(self._view, method, priority_fn=priority_fn, check=check, **kw)

    def lml_est(self) -> float:
        out = C.c_double()
        self._ck(self.L.gpf_shard_lml_est(self.h, C.byref(out)))
        return out.value

    def synchronize(self):
        self.state.synchronize()

    # scalar spec on the host (same functions the kernels use)
    def host_lse(self, m, S, K, flags):
        return self.L.gpf_host_lse(m, S, K, flags)

    def host_ess(self, S, Qhi, Qlo):
        return self.L.gpf_host_ess(S, Qhi, Qlo)

    def host_log(self, x):
        return self.L.gpf_host_log(x)

    def fix_K(self, n):
        return self.L.gpf_host_fix_K(n)


class ShardedParticleFilterState:
    """One rank's view of the sharded filter (the reference's ParticleFilterState, partitioned)."""

    def __init__(self, backend, model, n_global: int, rank: int, world: int, group=None):
        self.backend, self.model, self.n_global = backend, model, int(n_global)
        self.rank, self.world, self.group = rank, world, group
        self.gid0, self.n_local = shard_range(n_global, rank, world)
        self.bounds = [shard_range(n_global, r, world)[0] for r in range(world)] + [int(n_global)]
        self.device = backend.device
        self.K = backend.fix_K(self.n_global)

    @property
    def local(self):
        return self.backend.state

    def synchronize(self):
        self.backend.synchronize()

    def _stage(self, t: torch.Tensor) -> torch.Tensor:
        """gloo cannot move device tensors: stage through the host (only used by the one-GPU, two-process test;
        on the GPU box the backend is nccl = RCCL and tensors stay in HBM)."""
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            return t.cpu()
        return t

    # ---- collectives (tiny, latency-bound: SURVEY.md §2.3 C1-C4)
    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not _FORCE_COLLECTIVES:
            return t.unsqueeze(0)
        src = self._stage(t.contiguous().view(-1))
        flat = torch.empty(self.world * t.numel(), dtype=t.dtype, device=src.device)
        dist.all_gather_into_tensor(flat, src, group=self.group)
        return flat.to(t.device).view((self.world,) + tuple(t.shape))

    def _all_to_all(self, send: torch.Tensor, send_counts, recv_counts) -> torch.Tensor:
        """variable-size all-to-all along dim 0"""
        if self.world == 1 and not _FORCE_COLLECTIVES:
            return send
        src = self._stage(send.contiguous())
        out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=src.device)
        dist.all_to_all_single(out, src, output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=self.group)
        return out.to(send.device)

    # ---- global weight summary: phases 1 + 2
    def _summary(self, want_q=True):
        b = self.backend
        mf_all = self._all_gather(b.weight_max()).contiguous()           # (G, 2): max, flags & (NaN | +Inf)
        tot_all = self._all_gather(b.weight_scan(mf_all, want_q)).contiguous()    # (G, 5): S_r, Ql0..3 (limbs: ESS only)
        return mf_all, tot_all

    def _summary_scalars(self):
        mf_all, tot_all = self._summary()
        mf = mf_all.cpu().numpy()
        t = tot_all.cpu().numpy().astype(object)
        flags = 0
        for f in mf[:, 1]:
            flags |= int(f)
        S = int(sum(int(x) for x in t[:, 0]))
        Q = sum(int(sum(int(x) for x in t[:, 1 + k])) << (32 * k) for k in range(4))
        return float(mf[:, 0].max()), flags, S, Q


# ----------------------------------------------------------------------------- public API (sharded twins)
def pf_initialize(model, model_args, observations, n_particles: int, *, seed: int = 1, keep_prev: bool = False,
                  device: int = 0, group=None, backend_factory=None) -> ShardedParticleFilterState:
    """src/initialize.jl:31-44 over all ranks: `n_particles` is the GLOBAL particle count."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    gid0, n_local = shard_range(n_particles, rank, world)
    factory = backend_factory or HipShardBackend
    backend = factory(model, n_particles, gid0, n_local, seed, keep_prev, device)
    st = ShardedParticleFilterState(backend, model, n_particles, rank, world, group)
    # engine: "library" = gpf_shard_resample with the library's own RCCL communicator (the default wherever the process group
    # is RCCL, and for a single rank), "python" = the phase-by-phase composition below (gloo tests, oracle backend)
    engine = os.environ.get("GPF_SHARD_ENGINE") or ("library" if hasattr(backend, "comm_create") and (world == 1 or dist.get_backend(group) == "nccl") else "python")
    if engine == "library":
        backend.comm_create(rank, world, group)
    backend.initialize(_obs_vector(observations))
    return st


def pf_update(state: ShardedParticleFilterState, new_args, argdiffs, observations):
    """src/update.jl:12-25: embarrassingly parallel, no communication."""
    state.backend.update(_obs_vector(observations))
    return state


def pf_rejuvenate(state: ShardedParticleFilterState, kern=None, kern_args=(), n_iters: int = 1, *, method: str = "move"):
    """src/rejuvenate.jl:18-90: per particle, no communication."""
    if method not in ("move", "reweight"):
        raise ErrorException(f"Method {method} not recognized.")
    state.backend.rejuvenate(0 if method == "move" else 1, int(n_iters))
    return state


def pf_resample(state: ShardedParticleFilterState, method: str = "multinomial", *, priority_fn=None, check="warn",
                sort_particles: bool = False, local: bool = False):
    """src/resample.jl:19-175 with a global CDF.  Returns `state`.

    local=True: the communication-free alternative of SURVEY.md §8e ("island" filter): every shard resamples its own
    particles with the reference's SUB-STATE semantics (src/resample.jl:185-187,205-218) -- ancestors inside the shard,
    log-weights reset to the shard's average so its total mass is kept, the running log-ML estimate untouched; the global
    estimate stays log_ml_est + logsumexp(all weights) - log N (src/utils.jl:174-178).  A different (higher-variance when
    shard masses diverge) estimator than the global resample; equal to pf_resample!(state[shard range], ...) on an unsharded
    state, which is how it is tested.  priority_fn / sort_particles are supported here."""
    if method not in RESAMPLE_METHODS:
        raise ErrorException(f"Resampling method {method} not recognized.")
    if local:
        state.backend.local_resample(method, priority_fn, check, sort_particles)
        return state
    from .api import Tempering
    if priority_fn is not None and not isinstance(priority_fn, Tempering):
        raise ErrorException("sharded resampling supports priority_fn = nothing or Tempering(alpha) (w -> alpha w); "
                             "any other closure needs local=True")
    b, G, mid = state.backend, state.world, RESAMPLE_METHODS[method]
    sort_particles = bool(sort_particles) and method == "stratified"   # only the stratified resampler reads it (src/resample.jl:143-145)
    if sort_particles and priority_fn is not None:
        raise ErrorException("sharded stratified resampling with sort_particles=True (every rank sorts the gathered weights: the replicated plan) takes no priority_fn")
    if priority_fn is not None and not getattr(b, "lib_comm", False):
        raise ErrorException("a prioritised sharded resample runs in the library engine (gpf_shard_resample_tempered)")
    if getattr(b, "lib_comm", False):                                 # the whole exchange inside the library (RCCL)
        try:
            invalid = b.shard_resample(mid, check, None if priority_fn is None else priority_fn.alpha, sort_particles)
        except ErrorException as e:
            raise ErrorException("Invalid weights.") if "Invalid weights" in str(e) else e
        if invalid and check == "warn":
            import warnings
            warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")
        return state
    mf_all, tot_all = state._summary(want_q=False)                    # phases 1, 2 (no sum q^2: only the ESS needs it)
    if check is not False:                                            # safe_softmax validity (utils.jl:117-140), no stream sync
        flags = b.scan_flags()
        if (flags & 3) or (check is True and flags):
            raise ErrorException("Invalid weights.")                  # resample.jl:55
        if flags:
            import warnings
            warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")
    cap = min(state.n_global, 2 * state.n_local + 65536)
    if os.environ.get("GPF_PUSH_CAPACITY"):                           # tests: force the overflow path
        cap = int(os.environ["GPF_PUSH_CAPACITY"])
    if sort_particles:
        # the replicated plan (gpf.h gpf_shard_resample_sorted): ALL log-weights on every rank (shards of unequal size: padded), the unsharded sort + scan +
        # search on them by every rank itself -- so every rank knows every slot's ancestor and the exchange counts need no message
        lw = b.log_weights_tensor()
        per = max(state.bounds[r + 1] - state.bounds[r] for r in range(G))
        padded = torch.zeros(per, dtype=torch.float64, device=lw.device); padded[:state.n_local] = lw
        gathered = state._all_gather(padded)
        lw_all = torch.cat([gathered[r, :state.bounds[r + 1] - state.bounds[r]] for r in range(G)]).contiguous()
        b.sorted_count(lw_all, state.rank, state.bounds)
        buf = b.sorted_push(state.rank, state.bounds, cap)
        c = b.counts(G)
        sc, rc = c[:G], c[G:]
        if sum(sc) > cap:
            buf = b.sorted_push(state.rank, state.bounds, sum(sc))
        b.commit(state._all_to_all(buf[:sum(sc)], sc, rc), mf_all, tot_all)
        return state
    cr_all = state._all_gather(b.residual_scan(tot_all)).contiguous() if mid == 1 else None     # phase 2b: (G, 2)
    b.push_count(mid, tot_all, cr_all, state.rank, state.bounds)     # phase 3: who owns the target of which slot
    # phase 4 is enqueued BEFORE the host learns the counts, into a buffer sized for a balanced exchange with slack
    # (any size is correct: the kernel stops at the capacity, and the call is repeated if the counts say it overflowed)
    buf = b.push(mid, tot_all, cr_all, state.rank, state.bounds, cap)                  # look up, gather, pack
    c = b.counts(G)                                                   # ONE host sync (the all-to-all split sizes), behind phase 4
    sc, rc = c[:G], c[G:]
    if sum(sc) > cap:                                                 # skewed weights: this shard serves more than 2x its share
        buf = b.push(mid, tot_all, cr_all, state.rank, state.bounds, sum(sc))
    back = state._all_to_all(buf[:sum(sc)], sc, rc)                   # the exchange: [row | slot | ancestor id]
    b.commit(back, mf_all, tot_all)                                   # phase 5: scatter by slot, weights, log-ML
    return state


def pf_step_ess(state: ShardedParticleFilterState, new_args, argdiffs, observations, *, ess_threshold: float = 0.5, method: str = "multinomial",
                rejuvenate=None, n_iters: int = 1, check="warn") -> bool:
    """One iteration of the reference's README loop (README.md:66-77) on the sharded filter, called on every rank:

        if effective_sample_size(state) < ess_threshold * n_global;  pf_resample!(state, method);  pf_rejuvenate!(state, ...; method = rejuvenate);  end
        pf_update!(state, new_args, argdiffs, observations)

    Library engine: ONE C-ABI call (gpf_shard_step_ess: the summary reduction exchanges the shard totals through the mailboxes and leaves the
    verdict on the device, the propagate runs speculatively behind it); python engine: the separate calls.  Returns whether it resampled."""
    if method not in RESAMPLE_METHODS:
        raise ErrorException(f"Resampling method {method} not recognized.")
    if rejuvenate is not None and rejuvenate not in ("move", "reweight"):
        raise ErrorException(f"Method {rejuvenate} not recognized.")
    b = state.backend
    if getattr(b, "lib_comm", False):
        try:
            res, invalid = b.shard_step_ess(_obs_vector(observations), ess_threshold, RESAMPLE_METHODS[method], check,
                                            -1 if rejuvenate is None else (0 if rejuvenate == "move" else 1), n_iters)
        except ErrorException as e:
            raise ErrorException("Invalid weights.") if "Invalid weights" in str(e) else e
        if invalid and check == "warn":
            import warnings
            warnings.warn("Invalid weights (all -Inf or zero): resampled with uniform weights.")
        return res
    go = effective_sample_size(state) < ess_threshold * state.n_global
    if go:
        pf_resample(state, method, check=check)
        if rejuvenate is not None:
            pf_rejuvenate(state, None, (), n_iters, method=rejuvenate)
    pf_update(state, new_args, argdiffs, observations)
    return bool(go)


def effective_sample_size(state: ShardedParticleFilterState) -> float:
    """src/utils.jl:163-164 over the global weights"""
    if getattr(state.backend, "lib_comm", False):
        return state.backend.shard_ess()
    m, flags, S, Q = state._summary_scalars()
    if flags or m == -np.inf:
        return float("nan")
    return state.backend.host_ess(S, Q >> 64, Q & ((1 << 64) - 1))


get_ess = effective_sample_size


def log_ml_estimate(state: ShardedParticleFilterState) -> float:
    """Gen.log_ml_estimate over the global weights"""
    if getattr(state.backend, "lib_comm", False):
        return state.backend.shard_lml()
    m, flags, S, Q = state._summary_scalars()
    b = state.backend
    return b.lml_est() + b.host_lse(m, S, state.K, flags) - b.host_log(float(state.n_global))


get_lml_est = log_ml_estimate
