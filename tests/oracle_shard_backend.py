"""CPU stand-in for HipShardBackend, built on the oracle's C primitives (TEST INFRASTRUCTURE).
Lets the gloo tests drive genparticlefilters.jl_amd/sharded.py -- the real routing and collectives --
without a GPU.  Same method names and tensor shapes as HipShardBackend."""
import numpy as np
import torch

from oracle import oracle as o

SPACE_COUNTS = 1 << 62


class OracleShardBackend:
    def __init__(self, model, n_global, gid0, n_local, seed, keep_prev, device):
        self.device = torch.device("cpu")
        self.model, self.P = model.model_id, np.ascontiguousarray(model.params, np.float64)
        self.N, self.gid0, self.n, self.seed, self.keep_prev = n_global, gid0, n_local, seed, bool(keep_prev)
        self.W = o.row_width(self.model, self.keep_prev)
        self.rows = np.zeros((self.n, self.W)); self.lw = np.zeros(self.n)
        self.parents = np.arange(gid0 + 1, gid0 + self.n + 1, dtype=np.int64)
        self.epoch, self.has_prev, self.lml, self.obs = 0, False, 0.0, None
        self.K = o.fix_K(n_global)
        self.L = o.lib()

    # per-particle operations
    def initialize(self, obs):
        self.L.o_init(self.model, self.P, self.seed, self.epoch, self.gid0, self.n, self.W, obs, self.rows, self.lw)
        self.epoch += 1; self.has_prev = False; self.obs = obs; self.lml = 0.0

    def update(self, obs):
        new = np.empty_like(self.rows)
        self.L.o_step(self.model, self.P, self.seed, self.epoch, self.gid0, self.n, self.W, int(self.keep_prev), obs, self.rows, new, self.lw)
        self.rows = new; self.epoch += 1; self.has_prev = True; self.obs = obs

    def rejuvenate(self, method_id, n_iters):
        new = np.empty_like(self.rows)
        self.L.o_move(self.model, self.P, self.seed, self.epoch, self.gid0, self.n, self.W, int(self.has_prev), self.obs, n_iters,
                      method_id, self.rows, new, self.lw)
        self.rows = new; self.epoch += 1

    # shard phases (same signatures and tensor shapes as HipShardBackend)
    def weight_max(self):
        m, f = o.max_flags(self.lw)
        return torch.tensor([m, float(f & 3)], dtype=torch.float64)

    @staticmethod
    def _combine(mf_all):
        mf = mf_all.numpy()
        flags = 0
        for f in mf[:, 1]:
            flags |= int(f)
        return float(mf[:, 0].max()), flags

    def weight_scan(self, mf_all, want_q=True):
        m, f = self._combine(mf_all)
        uniform = (m == -np.inf) and not (f & 1)
        self._flags = f | (4 if uniform else 0)
        self.q = np.zeros(self.n, np.uint64) if f & 3 else o.fixq(self.lw, m, self.K, uniform)
        self.cdf, S, hi, lo = o.scan(self.q)
        Q = (hi << 64) | lo
        return torch.tensor([S] + [(Q >> (32 * k)) & 0xFFFFFFFF for k in range(4)], dtype=torch.int64)

    def scan_flags(self):
        return self._flags

    def residual_scan(self, tot_all):
        S = int(tot_all[:, 0].sum())
        sh = self.L.o_residual_shift(S, self.N)
        c = np.empty(self.n, np.uint64); r = np.empty(self.n, np.uint64)
        self.L.o_residual_split(self.q, self.n, self.N, S, sh, c, r)
        self.ccdf = np.cumsum(c, dtype=np.uint64)
        self.rcdf, Rs, _, _ = o.scan(r)
        self.serve_residual = True
        return torch.tensor([int(self.ccdf[-1]) if self.n else 0, Rs], dtype=torch.int64)

    def _all_targets(self, method_id, tot_all, cr_all):
        """target, space and owner of EVERY global output slot (every shard can evaluate them: counters use global ids)"""
        t = tot_all.numpy(); G = t.shape[0]
        S = int(t[:, 0].sum())
        inc = np.zeros(self.N, bool)
        if method_id == 0:
            T = o.targets_multinomial(self.seed, self.epoch, 0, self.N, S).astype(np.int64)
        elif method_id == 2:
            T = o.targets_stratified(self.seed, self.epoch, 0, self.N, self.N, S).astype(np.int64)
        elif method_id == 4:                                   # sorted uniforms (opt-in multinomial_sorted): the unsharded spec over the GLOBAL slots
            T = o.targets_sorted(self.seed, self.epoch, 0, self.N, S).astype(np.int64)
        else:
            cr = cr_all.numpy()
            Ctot, Rs = int(cr[:, 0].sum()), int(cr[:, 1].sum())
            T = o.targets_multinomial(self.seed, self.epoch, 0, self.N, Rs).astype(np.int64)
            jg = np.arange(self.N, dtype=np.int64)
            inc = jg < Ctot
            T = np.where(inc, jg, T)
        w = np.cumsum(cr_all.numpy()[:, 1] if method_id == 1 else t[:, 0])
        owner = np.minimum(np.searchsorted(w, T, side="right"), G - 1)
        base = np.concatenate([[0], w[:-1]])[owner]
        if method_id == 1:
            c = np.cumsum(cr_all.numpy()[:, 0])
            oc = np.minimum(np.searchsorted(c, T, side="right"), G - 1)
            owner = np.where(inc, oc, owner)
            base = np.where(inc, np.concatenate([[0], c[:-1]])[oc], base)
        return T - base, inc, owner

    def push_count(self, method_id, tot_all, cr_all, me, bounds):
        G = tot_all.shape[0]
        self._tl, self._inc, self._owner = self._all_targets(method_id, tot_all, cr_all)
        b = np.asarray(bounds)
        self._dest = np.searchsorted(b[1:], np.arange(self.N), side="right")          # shard that holds each slot
        mine = self._owner == me
        send = np.bincount(self._dest[mine], minlength=G)
        recv = np.bincount(self._owner[self._dest == me], minlength=G)
        self._bounds = b
        self._counts = [int(x) for x in np.concatenate([send, recv])]

    def counts(self, G):
        return list(self._counts)

    def push(self, method_id, tot_all, cr_all, me, bounds, capacity):
        hits = np.flatnonzero(self._owner == me)               # slot order = grouped by destination, slot order inside
        tl, inc = self._tl[hits].astype(np.uint64), self._inc[hits]
        a = np.zeros(hits.size, np.int64)
        wcdf = self.rcdf if getattr(self, "serve_residual", False) else self.cdf
        if (~inc).any():
            a[~inc] = o.upper_bound(wcdf, np.ascontiguousarray(tl[~inc]))
        if inc.any():
            a[inc] = o.upper_bound(self.ccdf, np.ascontiguousarray(tl[inc]))
        rows = o.gather_rows(self.rows, a) if hits.size else np.zeros((0, self.W))
        slot_local = hits - self._bounds[self._dest[hits]]
        meta = ((slot_local.astype(np.uint64) << np.uint64(32)) | (a + self.gid0).astype(np.uint64)).view(np.float64)
        out = np.full((capacity, self.W + 1), np.nan)          # like the device buffer: entries beyond the capacity are dropped
        k = min(capacity, hits.size)
        out[:k] = np.concatenate([rows, meta.reshape(-1, 1)], axis=1)[:k]
        return torch.from_numpy(out)

    # stratified with sort_particles = true: the replicated plan (every rank holds all log-weights and works the unsharded spec out itself)
    def log_weights_tensor(self):
        return torch.from_numpy(self.lw.copy())

    def sorted_count(self, lw_all, me, bounds):
        lw = np.ascontiguousarray(lw_all.numpy(), np.float64)
        assert lw.size == self.N
        sp = o.WeightSummary(lw, self.N)                        # safe_softmax of ALL weights (resample.jl:147-151)
        order = o.argsort_desc(lw)                              # :156-157
        cdf, S, _, _ = o.scan(sp.q[order])
        k = o.upper_bound(cdf, o.targets_stratified(self.seed, self.epoch, 0, self.N, self.N, S))   # :160-166
        self._anc = np.asarray(order[k], np.int64)              # ancestor (global id) of every global slot
        b = np.asarray(bounds); G = b.size - 1
        self._bounds = b
        self._dest = np.searchsorted(b[1:], np.arange(self.N), side="right")
        self._owner = np.searchsorted(b[1:], self._anc, side="right")
        send = np.bincount(self._dest[self._owner == me], minlength=G)
        recv = np.bincount(self._owner[self._dest == me], minlength=G)
        self._counts = [int(x) for x in np.concatenate([send, recv])]
        self._me = me

    def sorted_push(self, me, bounds, capacity):
        hits = np.flatnonzero(self._owner == me)                # slot order = grouped by destination
        a = self._anc[hits] - self.gid0
        rows = o.gather_rows(self.rows, a) if hits.size else np.zeros((0, self.W))
        slot_local = hits - self._bounds[self._dest[hits]]
        meta = ((slot_local.astype(np.uint64) << np.uint64(32)) | self._anc[hits].astype(np.uint64)).view(np.float64)
        out = np.full((capacity, self.W + 1), np.nan)
        kk = min(capacity, hits.size)
        out[:kk] = np.concatenate([rows, meta.reshape(-1, 1)], axis=1)[:kk]
        return torch.from_numpy(out)

    def commit(self, packed, mf_all, tot_all):
        pk = np.ascontiguousarray(packed.numpy())
        assert pk.shape[0] == self.n
        meta = np.ascontiguousarray(pk[:, self.W]).view(np.uint64)
        slot = (meta >> np.uint64(32)).astype(np.int64)
        assert np.array_equal(np.sort(slot), np.arange(self.n))                       # every slot exactly once
        rows = np.empty((self.n, self.W)); anc = np.empty(self.n, np.int64)
        rows[slot] = pk[:, :self.W]
        anc[slot] = (meta & np.uint64(0xFFFFFFFF)).astype(np.int64)
        self.rows, self.parents, self.lw = rows, anc + 1, np.zeros(self.n)
        m, f = self._combine(mf_all)
        if m == -np.inf and not (f & 1):
            f |= 4
        self.lml = self.lml + (self.L.o_lse_from(m, int(tot_all[:, 0].sum()), self.K, f) - o.olog(float(self.N)))
        self.epoch += 1
        self.serve_residual = False

    def local_resample(self, method, priority_fn, check, sort_particles):
        """sub-state resample of this shard (resample.jl:185-187,205-218): the oracle's OracleSubState over local arrays, with
        the RNG offset of the shard's first global particle"""
        v = o.OracleSubState.__new__(o.OracleSubState)
        v.source, v.start, v.n, v.sl, v.last_obs, v.n_accepted = self, self.gid0, self.n, slice(0, self.n), self.obs, 0
        v.resample(method, priority_alpha=None if priority_fn is None else priority_fn.alpha, sort_particles=sort_particles, check=check)

    @property
    def lml_est_value(self):
        return self.lml

    def lml_est(self):
        return self.lml

    def synchronize(self):
        pass

    def host_lse(self, m, S, K, flags):
        if m == -np.inf and not (flags & 1):
            flags |= 4
        return self.L.o_lse_from(m, S, K, flags)

    def host_ess(self, S, Qhi, Qlo):
        return self.L.o_ess_from(S, Qhi, Qlo)

    def host_log(self, x):
        return o.olog(x)

    def fix_K(self, n):
        return o.fix_K(n)

    @property
    def state(self):
        return self
