// libgpf_core.hip -- C ABI (include/gpf.h) over the gfx950 kernels of gpf_kernels.hpp: handle lifetime, particle buffers, the
// per-particle kernels (pf_initialize / pf_update! / pf_rejuvenate!), the deferred gather, views' enter / exit, getters.
//
// Host orchestration only: which kernels run for each pf_* operation, on one HIP stream, with all scalars (max, sums, log-ML estimate,
// residual counts) kept in device memory so that the common path (check = false / :warn without reading the flag) never waits for the GPU.
// Build: __graft_entry__.build_hip() -- four translation units (this one, libgpf_resample / _aux / _shard .hip) linked into libgpf_hip.so.
#include "gpf_host.hpp"

using namespace gpf;
using namespace gpfh;

namespace gpfh {

Bufs take_particle_buffers(gpf_filter* h)
{
    Bufs b;
    b.n = h->n; b.ntiles = h->ntiles; b.cur = h->cur;
    b.rows[0] = h->rows[0]; b.rows[1] = h->rows[1]; h->rows[0] = h->rows[1] = nullptr;
    b.lw = h->lw; b.lws = h->lws; b.lp = h->lp; b.dtmp = h->dtmp; h->lw = h->lws = h->lp = h->dtmp = nullptr;
    for (int i = 0; i < 3; ++i) {
        b.cdf[i] = h->cdf[i]; b.t16[i] = h->t16[i]; b.t256[i] = h->t256[i]; h->cdf[i] = h->t16[i] = h->t256[i] = nullptr;
        for (int j = 0; j < 2; ++j) { b.desc[i][j] = h->desc[i][j]; h->desc[i][j] = nullptr; }
        h->table[i] = nullptr; h->dcur[i] = 0;
    }
    b.anc = h->anc; b.order = h->order; b.idx_in = h->idx_in; h->anc = h->order = h->idx_in = nullptr;
    b.keys = h->keys; b.keys_out = h->keys_out; b.sort_tmp = h->sort_tmp;
    h->keys = h->keys_out = nullptr; h->sort_tmp = nullptr; h->sort_tmp_bytes = 0;
    return b;
}

void free_bufs(Bufs& b)
{
    void* p[] = {b.rows[0], b.rows[1], b.lw, b.lws, b.lp, b.dtmp, b.cdf[0], b.cdf[1], b.cdf[2], b.t16[0], b.t16[1], b.t16[2],
                 b.t256[0], b.t256[1], b.t256[2], b.desc[0][0], b.desc[0][1], b.desc[1][0], b.desc[1][1], b.desc[2][0], b.desc[2][1],
                 b.anc, b.order, b.idx_in, b.keys, b.keys_out, b.sort_tmp};
    for (void* q : p) if (q) (void)hipFree(q);
    b = Bufs();
}

// allocate the per-N buffers for h->n particles (fields must be null); sets ntiles, K, logN
gpf_status alloc_particle_buffers(gpf_filter* h)
{
    h->ntiles = (h->n + TILE - 1) / TILE;
    h->K = fix_K(h->cfg.n_global);
    h->logN = log_((double)h->cfg.n_global);
    h->cur = 0;
    const size_t n = (size_t)h->n, rb = n * (size_t)h->W * sizeof(double);
    HIP_TRY(h, hipMalloc(&h->rows[0], rb));
    HIP_TRY(h, hipMalloc(&h->rows[1], rb));
    HIP_TRY(h, hipMalloc(&h->lw, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->lws, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->lp, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->dtmp, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->cdf[0], (size_t)h->ntiles * TILE * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->t16[0], (size_t)h->ntiles * (TILE / 16) * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->t256[0], t256_bytes(h->ntiles)));
    const size_t db = (((size_t)2 * h->ntiles * sizeof(uint64_t)) + 15) & ~(size_t)15;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) {
            HIP_TRY(h, hipMalloc(&h->desc[i][j], db));
            HIP_TRY(h, hipMemsetAsync(h->desc[i][j], 0, db, h->stream));     // descriptors start invalid; kernels keep them so
        }
    HIP_TRY(h, hipMalloc(&h->anc, n * sizeof(int32_t)));
    return GPF_OK;
}


// PROP 0: the model's own sampler; 1: native custom proposal; 2: stratified
template <int M, bool KEEP, int PROP = 0>
void launch_step_t(gpf_filter* h, int grid, const GateIn* gate = nullptr)
{
    constexpr int Wc = row_width(Model<M>::D, KEEP);
    if constexpr ((PROP == 1 && !Model<M>::HAS_PROPOSAL) || (PROP == 2 && !Model<M>::HAS_STRATA)) { (void)h; (void)grid; return; }
    else if (h->pending_packed && h->pend_own) {
        // own-direct commit: the shard's own hits through the ancestor array FIRST (it reads anc[j] >= 0 / -1; the packed entries'
        // launch behind it overwrites the -1 with the received ancestors), then the received entries; one weight vector, one slot array
        const MaxSlots ms = next_slots(h);
        PackedCommit pg{nullptr, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, nullptr, (int)h->pend_mailbox, h->pend_own_range ? 2 : 1, h->cfg.gid0,
                        h->pend_own_range ? h->shard_plan->own_range : nullptr};
        // (a window exchange: the slots outside the own range are read from this rank's receive window in the same launch -- no packed entries)
        if (h->pend_ring) pg.ring = RingIn{h->ring + (int64_t)(h->pend_ring_seq & (RING_PARITIES - 1)) * h->ring_parity_words, h->pend_ring_seq, h->h_timeout};
        GPF_LAUNCH((k_step<M, Wc, KEEP, true, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pg);
        if (h->pend_m > 0) {
            const PackedCommit pc{h->pend_packed, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, nullptr, nullptr, (int)h->pend_mailbox, 0, 0, nullptr};
            g_ev_start = g_ev_stop = nullptr;                    // (timed(): the event pair belongs to the first launch)
            GPF_LAUNCH((k_step<M, Wc, KEEP, false, PROP, true>), dim3(grid_for(h, h->pend_m, STEP_BLOCKS_PER_CU)), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                               h->cfg.gid0, h->pend_m, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
        }
    }
    else if (h->pending_packed) {
        const MaxSlots ms = next_slots(h);
        const PackedCommit pc{h->pend_packed, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, nullptr, (int)h->pend_mailbox, 0, 0, nullptr};
        GPF_LAUNCH((k_step<M, Wc, KEEP, false, PROP, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
    } else if (h->pending_gather && h->pending_search && PROP == 0) {
        // the pending multinomial search rides in the propagate (gpf_k_fused.hpp): one 1024-thread workgroup per CU like k_search_multi
        const MaxSlots ms = next_slots(h);
        const SearchArgs& sa = h->pend_sa;
        const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((sa.n + FCH - 1) / FCH, (int64_t)h->n_cu));
        const size_t tbytes = (multi_lds_bytes(sa.ntiles, sa.w.logg) + 15) & ~(size_t)15;
        if (sa.w.logg == 0)
            GPF_LAUNCH((k_step_search<M, Wc, KEEP, 0>), dim3(gsr), dim3(SBLOCK), tbytes + FUSED_LDS_EXTRA, h->stream, h->args, h->cfg.seed, h->epoch, sa,
                       h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, (uint32_t)(tbytes / 4));
        else
            GPF_LAUNCH((k_step_search<M, Wc, KEEP, 1>), dim3(gsr), dim3(SBLOCK), tbytes + FUSED_LDS_EXTRA, h->stream, h->args, h->cfg.seed, h->epoch, sa,
                       h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, (uint32_t)(tbytes / 4));
        h->pending_search = false;
    } else if (h->pending_gather) {
        const MaxSlots ms = next_slots(h);
        PackedCommit pc{};
        pc.lw_fill = h->pending_fill ? &h->sc->lw_fill : nullptr;
        GPF_LAUNCH((k_step<M, Wc, KEEP, true, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
    }
    else {
        PackedCommit pc{};
        if (gate) pc.gate = *gate;                                   // (gpf_step_ess: a speculative propagate behind the ESS reduction)
        GPF_LAUNCH((k_step<M, Wc, KEEP, false, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, next_slots(h), pc);
    }
}
template <int M, int PROP = 0>
void launch_init_t(gpf_filter* h, int grid)
{
    if constexpr ((PROP == 1 && !Model<M>::HAS_PROPOSAL) || (PROP == 2 && !Model<M>::HAS_STRATA) || (PROP == 3 && !Model<M>::HAS_STRATA_PROPOSAL)) { (void)h; (void)grid; return; }
    else
        GPF_LAUNCH((k_init<M, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->W, h->rows[h->cur], h->lw, next_slots(h));
}
bool model_has_proposal(int model)
{
    switch (model) {
        case MODEL_LGSSM2: return Model<MODEL_LGSSM2>::HAS_PROPOSAL;
        case MODEL_BEARINGS4: return Model<MODEL_BEARINGS4>::HAS_PROPOSAL;
        case MODEL_SV1: return Model<MODEL_SV1>::HAS_PROPOSAL;
        case MODEL_OBJECT_MOTION: return Model<MODEL_OBJECT_MOTION>::HAS_PROPOSAL;
        case MODEL_LINE: return Model<MODEL_LINE>::HAS_PROPOSAL;
    }
    return false;
}
bool model_has_strata(int model)
{
    switch (model) {
        case MODEL_LGSSM2: return Model<MODEL_LGSSM2>::HAS_STRATA;
        case MODEL_BEARINGS4: return Model<MODEL_BEARINGS4>::HAS_STRATA;
        case MODEL_SV1: return Model<MODEL_SV1>::HAS_STRATA;
        case MODEL_OBJECT_MOTION: return Model<MODEL_OBJECT_MOTION>::HAS_STRATA;
        case MODEL_LINE: return Model<MODEL_LINE>::HAS_STRATA;
    }
    return false;
}
template <int M, bool RW>
void launch_move_prop_t(gpf_filter* h, int grid, int n_iters)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    if constexpr (!Model<M>::HAS_MOVE_PROPOSAL) { (void)h; (void)grid; (void)n_iters; return; }
    else if (h->pending_gather)
        GPF_LAUNCH((k_move<M, Wc, RW, true, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
    else
        GPF_LAUNCH((k_move<M, Wc, RW, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
}
bool model_has_move_proposal(int model)
{
    switch (model) {
        case MODEL_LGSSM2: return Model<MODEL_LGSSM2>::HAS_MOVE_PROPOSAL;
        case MODEL_BEARINGS4: return Model<MODEL_BEARINGS4>::HAS_MOVE_PROPOSAL;
        case MODEL_SV1: return Model<MODEL_SV1>::HAS_MOVE_PROPOSAL;
        case MODEL_OBJECT_MOTION: return Model<MODEL_OBJECT_MOTION>::HAS_MOVE_PROPOSAL;
        case MODEL_LINE: return Model<MODEL_LINE>::HAS_MOVE_PROPOSAL;
    }
    return false;
}
template <int M, bool RW>
void launch_move_t(gpf_filter* h, int grid, int n_iters, const ModelArgs& args, uint32_t epoch)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    if (h->pending_gather)           // the resample gather rides on the move (rows read through anc, incoming weights 0)
        GPF_LAUNCH((k_move<M, Wc, RW, true>), dim3(grid), dim3(BLOCK), 0, h->stream, args, h->cfg.seed, epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
    else
        GPF_LAUNCH((k_move<M, Wc, RW, false>), dim3(grid), dim3(BLOCK), 0, h->stream, args, h->cfg.seed, epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
}
// the pending move inside the propagate (k_move_step): old observation + the move's epoch, new observation (h->args) + the update's epoch
template <int M, bool RW>
void launch_move_step_t(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    ObsVec om;
    for (int i = 0; i < MAX_OBS; ++i) om.v[i] = h->pm_args.obs[i];
    const MaxSlots ms = next_slots(h);
    if (h->pending_packed && h->pend_own) {
        // a sharded commit is pending (gpf_shard_commit, own-direct): the shard's own hits through the ancestor array -- and, after a window exchange, the
        // other slots out of the receive window -- in the first launch, the received packed entries in a second one (as launch_step_t)
        PackedCommit pg{nullptr, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, nullptr, (int)h->pend_mailbox, h->pend_own_range ? 2 : 1, h->cfg.gid0,
                        h->pend_own_range ? h->shard_plan->own_range : nullptr};
        if (h->pend_ring) pg.ring = RingIn{h->ring + (int64_t)(h->pend_ring_seq & (RING_PARITIES - 1)) * h->ring_parity_words, h->pend_ring_seq, h->h_timeout};
        GPF_LAUNCH((k_move_step<M, Wc, RW, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                   h->cfg.gid0, h->n, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pg);
        if (h->pend_m > 0) {
            const PackedCommit pc{h->pend_packed, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, nullptr, nullptr, (int)h->pend_mailbox, 0, 0, nullptr};
            g_ev_start = g_ev_stop = nullptr;                    // (timed(): the event pair belongs to the first launch)
            GPF_LAUNCH((k_move_step<M, Wc, RW, false, true>), dim3(grid_for(h, h->pend_m, MOVE_BLOCKS_PER_CU)), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                       h->cfg.gid0, h->pend_m, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
        }
    } else if (h->pending_packed) {
        const PackedCommit pc{h->pend_packed, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, nullptr, (int)h->pend_mailbox, 0, 0, nullptr};
        GPF_LAUNCH((k_move_step<M, Wc, RW, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                   h->cfg.gid0, h->n, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
    } else if (h->pending_gather)
        GPF_LAUNCH((k_move_step<M, Wc, RW, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                   h->cfg.gid0, h->n, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, PackedCommit{});
    else
        GPF_LAUNCH((k_move_step<M, Wc, RW, false>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                   h->cfg.gid0, h->n, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, PackedCommit{});
}

void launch_gather_ex(gpf_filter* h, const int32_t* anc, const double* in, double* out, const PrioView& pv, double* lw_out, int64_t n)
{
    const int grid = grid_for(h, n * (h->W / 2), 8);
    switch (h->W) {
        case 2: GPF_LAUNCH((k_gather<2>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, in, out, pv, lw_out, n); break;
        case 4: GPF_LAUNCH((k_gather<4>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, in, out, pv, lw_out, n); break;
        case 8: GPF_LAUNCH((k_gather<8>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, in, out, pv, lw_out, n); break;
    }
}
void launch_gather(gpf_filter* h, const PrioView& pv, double* lw_out)
{
    launch_gather_ex(h, h->anc, h->rows[h->cur], h->rows[1 - h->cur], pv, lw_out, h->n);
}
void launch_gather_rows_lw(gpf_filter* h, const int32_t* anc, const double* rows_in, const double* lw_in, double* rows_out,
                           double* lw_out, int64_t n)
{
    const int grid = grid_for(h, n * (h->W / 2), 8);
    switch (h->W) {
        case 2: GPF_LAUNCH((k_gather_rows_lw<2>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, rows_in, lw_in, rows_out, lw_out, n); break;
        case 4: GPF_LAUNCH((k_gather_rows_lw<4>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, rows_in, lw_in, rows_out, lw_out, n); break;
        case 8: GPF_LAUNCH((k_gather_rows_lw<8>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, rows_in, lw_in, rows_out, lw_out, n); break;
    }
}

// a pending resample gather is executed now: rows[1-cur][j] = rows[cur][anc[j]], lw = 0
gpf_status materialize(gpf_filter* h)
{
    if (h->pending_packed) {                                     // scatter the received exchange buffer by slot (+ log-ML update)
        double* out = h->rows[1 - h->cur];
        const int64_t m = h->pend_own ? h->pend_m : h->n;
        if (h->pend_own) {                                       // the shard's own hits first (anc[j] >= 0; the scatter below overwrites the -1 of the others)
            const int go = grid_for(h, h->n * (h->W / 2), 8);
            const int64_t* own_rng = h->pend_own_range ? h->shard_plan->own_range : nullptr;
            switch (h->W) {
                case 2: GPF_LAUNCH((k_gather_own<2>), dim3(go), dim3(BLOCK), 0, h->stream, h->anc, h->cfg.gid0, h->rows[h->cur], out, h->lw, h->n, own_rng); break;
                case 4: GPF_LAUNCH((k_gather_own<4>), dim3(go), dim3(BLOCK), 0, h->stream, h->anc, h->cfg.gid0, h->rows[h->cur], out, h->lw, h->n, own_rng); break;
                case 8: GPF_LAUNCH((k_gather_own<8>), dim3(go), dim3(BLOCK), 0, h->stream, h->anc, h->cfg.gid0, h->rows[h->cur], out, h->lw, h->n, own_rng); break;
            }
        }
        const int grid = grid_for(h, std::max<int64_t>(m, 1), 8);
        if (h->pend_ring) {                                      // a window exchange: the other slots' entries out of the receive window (+ log-ML update)
            const RingIn rin{h->ring + (int64_t)(h->pend_ring_seq & (RING_PARITIES - 1)) * h->ring_parity_words, h->pend_ring_seq, h->h_timeout};
            // (ascending targets: few slots outside the own range -- a small grid; i.i.d. targets: (G-1)/G of all slots sit in the window)
            const int gr = h->pend_own_range ? grid_for(h, std::min<int64_t>(h->n, (int64_t)1 << 16), 8) : grid_for(h, h->n, 8);
            const int64_t* own_rng_ring = h->pend_own_range ? h->shard_plan->own_range : nullptr;
            switch (h->W) {
                case 2: GPF_LAUNCH((k_commit_ring<2>), dim3(gr), dim3(BLOCK), 0, h->stream, rin, h->n, own_rng_ring, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
                case 4: GPF_LAUNCH((k_commit_ring<4>), dim3(gr), dim3(BLOCK), 0, h->stream, rin, h->n, own_rng_ring, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
                case 8: GPF_LAUNCH((k_commit_ring<8>), dim3(gr), dim3(BLOCK), 0, h->stream, rin, h->n, own_rng_ring, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
            }
        } else
        switch (h->W) {
            case 2: GPF_LAUNCH((k_commit_packed<2>), dim3(grid), dim3(BLOCK), 0, h->stream, h->pend_packed, m, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
            case 4: GPF_LAUNCH((k_commit_packed<4>), dim3(grid), dim3(BLOCK), 0, h->stream, h->pend_packed, m, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
            case 8: GPF_LAUNCH((k_commit_packed<8>), dim3(grid), dim3(BLOCK), 0, h->stream, h->pend_packed, m, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
        }
        HIP_TRY(h, hipGetLastError());
        h->cur ^= 1;
        h->pending_packed = false; h->pend_own = false; h->pend_ring = false;
        h->max_valid = false;
        return GPF_OK;
    }
    if (!h->pending_gather) return GPF_OK;
    { gpf_status fs = finish_search(h); if (fs) return fs; }         // (a lazy multinomial resample: its ancestors first)
    gpf_status s = timed(h, GPF_K_GATHER, [&] { launch_gather(h, raw_view(h), h->lw); });
    if (s) return s;
    if (h->pending_fill) GPF_LAUNCH(k_fill_from, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->lw, h->n, &h->sc->lw_fill);
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;                    // update_refs! (utils.jl:10-15)
    h->pending_gather = false; h->pending_fill = false;
    h->max_valid = false;           // log-weights are all 0 now
    return GPF_OK;
}

// ------------------------------------------------------------------ trajectory store
void hist_clear(gpf_filter* h)
{
    for (double* p : h->hist_x) if (p) (void)hipFree(p);
    for (int32_t* p : h->hist_map) if (p) (void)hipFree(p);
    h->hist_x.clear(); h->hist_map.clear(); h->hist_step = -1;
}
// snapshot the latent columns of the CURRENT step (final order: called when the step is over, or at query time)
gpf_status hist_snapshot(gpf_filter* h)
{
    if (!h->hist_on || h->hist_step < 0) return GPF_OK;
    gpf_status s = materialize(h);
    if (s) return s;
    double*& dst = h->hist_x[h->hist_step];
    if (!dst) HIP_TRY(h, hipMalloc(&dst, (size_t)h->n * h->d * sizeof(double)));
    GPF_LAUNCH(k_hist_snapshot, dim3(grid_for(h, h->n * h->d, 8)), dim3(BLOCK), 0, h->stream, h->rows[h->cur], h->W, h->d, h->n, dst);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// a resample happened during the current step: compose its ancestors into the step's map
gpf_status hist_on_resample(gpf_filter* h)
{
    if (!h->hist_on || h->hist_step < 0) return GPF_OK;
    int32_t* old = h->hist_map[h->hist_step];
    int32_t* neu = nullptr;
    HIP_TRY(h, hipMalloc(&neu, (size_t)h->n * sizeof(int32_t)));
    GPF_LAUNCH(k_hist_compose, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->anc, old, h->n, neu);
    HIP_TRY(h, hipGetLastError());
    if (old) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(old); }
    h->hist_map[h->hist_step] = neu;
    return GPF_OK;
}
gpf_status hist_begin_step(gpf_filter* h, bool first)
{
    if (!h->hist_on) return GPF_OK;
    if (first) hist_clear(h);
    else { gpf_status s = hist_snapshot(h); if (s) return s; }     // the step that ends now, in its final order
    if ((int)h->hist_x.size() >= h->hist_cap) return fail(h, GPF_ERR_STATE, "trajectory store full: raise max_steps of gpf_history_enable");
    h->hist_x.push_back(nullptr); h->hist_map.push_back(nullptr);
    h->hist_step = (int)h->hist_x.size() - 1;
    return GPF_OK;
}

// Every change of a filter's rows / log-weights -- through the filter itself or through any view of it -- bumps the ROOT's
// mutation counter.  A view's cached summaries (raw CDF, sum q^2, producer maxima) describe the weights at the value it last
// saw; view_enter drops them when the counter has moved (the reference's SubArray views are live, src/view.jl:35-48).
void mutated(gpf_filter* h)
{
    gpf_filter* root = h->parent ? h->parent : h;
    root->mutations += 1;
    if (h->parent) h->seen_mutations = root->mutations;          // its own change: this view's bookkeeping is already current
}

// A view re-derives its aliased pointers from the parent on every call (the parent may have swapped its row buffers),
// shares the parent's epoch counter, and forces a pending gather of the parent first.
gpf_status view_enter(gpf_filter* v)
{
    if (v->orphaned) return fail(v, GPF_ERR_STATE, "stale view: its filter was destroyed");
    gpf_filter* p = v->parent;
    if (p->generation != v->parent_generation) return fail(v, GPF_ERR_STATE, "stale view: the parent filter was resized or re-created");
    if (!p->initialized) return fail(v, GPF_ERR_STATE, "parent filter not initialised");
    if (p->pending_move) { gpf_status ms = finish_move(p); if (ms) { v->err = p->err; return ms; } }
    gpf_status s = materialize(p);
    if (s) { v->err = p->err; return s; }
    const int64_t o = v->view_start;
    if (v->view_step == 1) {
        v->rows[0] = p->rows[p->cur] + o * v->W;
        v->rows[1] = p->rows[1 - p->cur] + o * v->W;
        v->lw = p->lw + o;
        v->anc = p->anc + o;
    } else {                                                     // strided: a compact copy of particles o + i * step
        v->rows[0] = v->vrows[0]; v->rows[1] = v->vrows[1]; v->lw = v->vlw; v->anc = v->vanc;
        if (v->view_step == 0)                                   // state[idxs]: an arbitrary index vector
            GPF_LAUNCH(k_view_index_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur], p->lw, p->anc,
                       v->vrows[0], v->vlw, v->vanc, v->W, v->vidx, v->n, 1);
        else
        GPF_LAUNCH(k_view_strided_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur] + o * v->W, p->lw + o, p->anc + o,
                   v->vrows[0], v->vlw, v->vanc, v->W, v->view_step, v->n, 1);
        HIP_TRY(v, hipGetLastError());
    }
    v->cur = 0;
    v->epoch = p->epoch;
    v->has_prev = p->has_prev;
    v->initialized = true;
    if (v->seen_mutations != p->mutations) {                     // the aliased weights changed behind this view's back
        v->raw_valid = false; v->raw_sum_valid = false; v->raw_has_q = false; v->raw_q_folded = false; v->max_valid = false;
        v->seen_mutations = p->mutations;
    }
    return GPF_OK;
}
// after a mutating call on a view: update_refs! for sub-states copies back (utils.jl:17-20); parent caches are stale
gpf_status view_exit(gpf_filter* v)
{
    if (!v->parent) return GPF_OK;
    gpf_filter* p = v->parent;
    if (v->view_step != 1) {                                     // strided: scatter the compact copy back into the source
        const int64_t o = v->view_start;
        if (v->view_step == 0)
            GPF_LAUNCH(k_view_index_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur], p->lw, p->anc,
                       v->rows[v->cur], v->vlw, v->vanc, v->W, v->vidx, v->n, 0);
        else
        GPF_LAUNCH(k_view_strided_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur] + o * v->W, p->lw + o, p->anc + o,
                   v->rows[v->cur], v->vlw, v->vanc, v->W, v->view_step, v->n, 0);
        HIP_TRY(v, hipGetLastError());
        v->cur = 0;
    } else if (v->cur == 1) {
        HIP_TRY(v, hipMemcpyAsync(v->rows[0], v->rows[1], (size_t)v->n * v->W * sizeof(double), hipMemcpyDeviceToDevice, v->stream));
        v->cur = 0;
    }
    p->epoch = v->epoch;
    p->has_prev = p->has_prev || v->has_prev;
    p->raw_valid = false; p->raw_sum_valid = false; p->max_valid = false; p->raw_has_q = false; p->raw_q_folded = false;
    return GPF_OK;
}

gpf_status check_ready(gpf_handle h, bool keep_pending_move)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(h, hipSetDevice(h->cfg.device));       // launches go to the calling thread's current device: before anything below enqueues
    if (h->pending_move && !keep_pending_move) { gpf_status ms = finish_move(h); if (ms) return ms; }   // (a lazy move: only the plain pf_update! carries it)
    if (h->parent) { gpf_status vs = view_enter(h); if (vs) return vs; }
    if (!h->initialized) return fail(h, GPF_ERR_STATE, "filter not initialised: call gpf_initialize (pf_initialize) first");
    return GPF_OK;
}

gpf_status set_obs(gpf_filter* h, const double* obs, int n_obs)
{
    if (n_obs < 0 || n_obs > MAX_OBS || (n_obs > 0 && !obs)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad observation vector");
    // a native model's step is defined by its full data vector: an empty choicemap() (no constraint, weight 0) has no
    // device meaning and must not silently become "observed zeros"
    if (n_obs != model_obs_dim(h->cfg.model))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model takes " + std::to_string(model_obs_dim(h->cfg.model)) + " observation values per step");
    for (int i = 0; i < MAX_OBS; ++i) h->args.obs[i] = i < n_obs ? obs[i] : 0.0;
    return GPF_OK;
}

// a pending lazy move is wanted as a state after all: the stand-alone k_move with the arguments and the epoch of its pf_rejuvenate! call
gpf_status finish_move(gpf_filter* h)
{
    if (!h->pending_move) return GPF_OK;
    h->pending_move = false;
    if (h->pending_packed) { gpf_status ms = materialize(h); if (ms) return ms; }   // (a sharded commit waited with the move: scatter it, then the stand-alone move)
    const bool fused_gather = h->pending_gather;
    const int grid = move_grid(h);
    const int n_iters = h->pm_iters;
    gpf_status s = timed(h, GPF_K_MOVE, [&] {
        if (h->pm_method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_t<MM, true>(h, grid, n_iters, h->pm_args, h->pm_epoch))); }
        else                                         { DISPATCH_MODEL(h, (launch_move_t<MM, false>(h, grid, n_iters, h->pm_args, h->pm_epoch))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;                                                 // (the epoch was consumed at the call)
    if (fused_gather) { h->pending_gather = false; h->pending_fill = false; h->max_valid = false; }
    if (h->pm_method == GPF_REJUVENATE_REWEIGHT) { h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = true; }
    mutated(h);
    return GPF_OK;
}
// which models / sizes k_step_search covers: the key-table regime of the search (up to 2.5 M particles)
gpf_status copy_out(gpf_handle h, const void* dsrc, void* out, size_t bytes)
{
    HIP_TRY(h, hipMemcpyAsync(out, dsrc, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GPF_OK;
}


} // namespace gpfh

extern "C" {

int gpf_abi_version(void) { return GPF_ABI_VERSION; }

const char* gpf_last_error(gpf_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

gpf_status gpf_create(const gpf_config* cfg, gpf_handle* out)
{
    if (!cfg || !out) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null config/out");
    *out = nullptr;
    if (cfg->abi_version != GPF_ABI_VERSION) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "ABI version mismatch");
    const int d = model_dim(cfg->model);
    if (d == 0) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "unknown model id");
    if (cfg->n_params < 0 || cfg->n_params > MAX_PARAMS || (cfg->n_params > 0 && !cfg->params))
        return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "bad parameter vector");
    if (cfg->n_particles < 1 || cfg->n_global < cfg->n_particles || cfg->gid0 < 0 ||
        cfg->gid0 + cfg->n_particles > cfg->n_global || cfg->n_global >= ((int64_t)1 << 31))
        return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "bad particle counts (need 1 <= n <= n_global < 2^31)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, GPF_ERR_NO_DEVICE, "no HIP device: libgpf_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "bad device ordinal");

    gpf_filter* h = new gpf_filter();
    h->lazy_search = getenv("GPF_LAZY_SEARCH") && !strcmp(getenv("GPF_LAZY_SEARCH"), "1");
    h->lazy_move = !(getenv("GPF_LAZY_MOVE") && !strcmp(getenv("GPF_LAZY_MOVE"), "0"));
    h->cfg = *cfg;
    h->cfg.params = nullptr;
    for (int i = 0; i < cfg->n_params; ++i) h->args.P[i] = cfg->params[i];
    h->args.gstride = 1;
    h->d = d;
    h->W = row_width(d, cfg->keep_prev != 0);
    h->n = cfg->n_particles;
    gpf_status st = GPF_OK;
    auto body = [&]() -> gpf_status {
        HIP_TRY(h, hipSetDevice(cfg->device));
        hipDeviceProp_t prop;
        HIP_TRY(h, hipGetDeviceProperties(&prop, cfg->device));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        {   // the key tables of the searches, the bucket sort and the fused kernels are sized for gfx950's 160 KB of LDS per workgroup
            int lds_max = 0;
            HIP_TRY(h, hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device));
            if (lds_max < 160 * 1024)
                return fail(h, GPF_ERR_NO_DEVICE, "libgpf_hip is built for gfx950 (MI355X: 160 KB of LDS per workgroup); this device offers " + std::to_string(lds_max) + " bytes");
        }
        if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
        else { HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
        if (cfg->device < 16) {                                  // one more filter on this device (gpf_host.hpp ChainGate)
            std::lock_guard<std::mutex> lk(g_chain[cfg->device].mu);
            g_chain[cfg->device].live += 1; h->chain_counted = true;
        }
        { gpf_status a_ = alloc_particle_buffers(h); if (a_) return a_; }
        const size_t n = (size_t)h->n, rb = n * (size_t)h->W * sizeof(double);
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(h, hipMalloc(&h->mslots[b], (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemsetAsync(h->mslots[b], 0, (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long), h->stream));
        }
        HIP_TRY(h, hipMalloc(&h->blockQ, (size_t)4 * 8 * h->n_cu * sizeof(uint64_t) + 64));   // (4 limbs per scan workgroup, <= 8 workgroups per CU)
        // (zeroed: the ESS scan recognises this launch's partials by a tag in their top bits -- recycled memory may hold another filter's)
        HIP_TRY(h, hipMemsetAsync(h->blockQ, 0, (size_t)4 * 8 * h->n_cu * sizeof(uint64_t) + 64, h->stream));
        HIP_TRY(h, hipMalloc(&h->partial, MAX_PARTIALS * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->acc_part, MAX_PARTIALS * sizeof(unsigned long long)));
        HIP_TRY(h, hipMalloc(&h->dscal, 4 * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->sc, sizeof(Scalars)));
        HIP_TRY(h, hipHostMalloc(&h->h_sc, sizeof(Scalars)));
        HIP_TRY(h, hipHostMalloc(&h->h_timeout, sizeof(int32_t)));
        *h->h_timeout = 0;
        { gpf_status su = resample_device_setup(h); if (su) return su; }
        { gpf_status su = shard_device_setup(h); if (su) return su; }
        HIP_TRY(h, hipMemsetAsync(h->sc, 0, sizeof(Scalars), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->lw, 0, n * sizeof(double), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->rows[0], 0, rb, h->stream));
        GPF_LAUNCH(k_iota, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->anc, h->n);   // parents = 1:N
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return GPF_OK;
    };
    st = body();
    if (st != GPF_OK) { g_err = h->err; gpf_destroy(h); return st; }
    *out = h;
    return GPF_OK;
}

gpf_status gpf_destroy(gpf_handle h)
{
    if (!h) return GPF_OK;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->parent && !h->orphaned) {                              // a view leaves its filter's list
        auto& vs = h->parent->views;
        vs.erase(std::remove(vs.begin(), vs.end(), h), vs.end());
    }
    for (auto& t : h->timers) for (auto& e : t.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& m : h->phases.marks) (void)hipEventDestroy(m.second);
    h->phases.marks.clear();
    if (h->chain_counted || h->chain_ev) {                       // (gpf_host.hpp ChainGate; the stream is drained: nothing waits for this filter's event any more)
        ChainGate& cg = g_chain[h->cfg.device < 16 ? h->cfg.device : 0];
        std::lock_guard<std::mutex> lk(cg.mu);
        if (h->chain_counted) { cg.live -= 1; h->chain_counted = false; }
        if (h->chain_ev) {
            if (cg.last == h->chain_ev) { cg.last = nullptr; cg.last_stream = nullptr; }
            (void)hipEventDestroy(h->chain_ev); h->chain_ev = nullptr;
        }
    }
    h->pending_packed = false;                                   // the filter goes away: nothing to scatter a deferred commit into
    if (h->planner) { gpf_destroy(h->planner); h->planner = nullptr; }
    for (void* q : {(void*)h->sorted_src, (void*)h->sorted_gath, (void*)h->anc_cursors}) if (q) (void)hipFree(q);
    { const std::vector<gpf_filter*> own = h->blk_views; h->blk_views.clear(); for (gpf_filter* v : own) gpf_destroy(v); }
    // view handles the host still holds outlive this filter as orphans: every later call on them fails ("stale view"), their own gpf_destroy frees
    // what they own -- the aliased buffers and the stream (drained above) are never touched through them again
    for (gpf_filter* v : h->views) { v->orphaned = true; v->stream = nullptr; v->rows[0] = v->rows[1] = nullptr; v->lw = nullptr; v->anc = nullptr; }
    h->views.clear();
    gpf_comm_destroy(h);
    hist_clear(h);
    if (h->hist_dev_maps) (void)hipFree(h->hist_dev_maps);
    if (h->parent) { h->rows[0] = h->rows[1] = nullptr; h->lw = nullptr; h->anc = nullptr; }   // aliases of the parent's buffers (or of the compact copies below)
    for (void* q : {(void*)h->vrows[0], (void*)h->vrows[1], (void*)h->vlw, (void*)h->vanc, (void*)h->vidx, (void*)h->vgid}) if (q) (void)hipFree(q);
    { Bufs b = take_particle_buffers(h); free_bufs(b); }
    void* bufs[] = {h->mslots[0], h->mslots[1], h->blockQ, h->partial, h->dscal, h->sc, h->push_stage, h->shard_counts, h->shard_plan, h->tree_buf, h->acc_part,
                    h->pull_req, h->pull_counts, h->pull_pc, h->pull_pc_all, h->blk_words, h->blk_mask, h->blk_stats, h->blk_obs};
    for (void* b : bufs) if (b) hipFree(b);
    if (h->h_sc) hipHostFree(h->h_sc);
    if (h->h_sc_ticket) hipHostFree(h->h_sc_ticket);
    if (h->h_shard_counts) hipHostFree(h->h_shard_counts);
    if (h->h_pull_pc_all) hipHostFree(h->h_pull_pc_all);
    if (h->h_qpub) hipHostFree(h->h_qpub);
    if (h->h_spart) hipHostFree(h->h_spart);
    if (h->sp_g) { (void)hipFree(h->sp_g); (void)hipFree(h->sp_vlo); }
    if (h->splan_F) { (void)hipFree(h->splan_F); (void)hipFree(h->splan_arrive); }
    if (h->gate_part) { (void)hipFree(h->gate_part); hipHostFree(h->h_gate); }
    if (h->sum_part) (void)hipFree(h->sum_part);
    for (int k = 0; k < gpf_filter::BLK_STAGE; ++k) if (h->h_blk_obs[k]) hipHostFree(h->h_blk_obs[k]);
    if (h->h_blk_done) hipHostFree(h->h_blk_done);
    if (h->blk_stage_counter) (void)hipFree(h->blk_stage_counter);
    if (h->h_flags) hipHostFree(h->h_flags);
    if (h->h_sort_flag) hipHostFree(h->h_sort_flag);
    if (h->h_timeout) hipHostFree(h->h_timeout);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return GPF_OK;
}

gpf_status gpf_set_lazy_search(gpf_handle h, int32_t enable)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!enable) { gpf_status s = finish_search(h); if (s) return s; }
    h->lazy_search = enable != 0;
    return GPF_OK;
}

gpf_status gpf_synchronize(gpf_handle h)
{
    // (work that was left for a later call to pick up is enqueued now: a timed loop that ends in pf_rejuvenate! pays for its move)
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->pending_move) { gpf_status ms = finish_move(h); if (ms) return ms; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_scan_timeout(h);
}

static gpf_status initialize_impl(gpf_handle h, const double* obs, int32_t n_obs, int prop)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (prop == 1 && !model_has_proposal(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no native proposal");
    if (prop == 2 && !model_has_strata(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no discrete latent to stratify over");
    if (prop == 3 && h->cfg.model != MODEL_LINE) return fail(h, GPF_ERR_INVALID_ARGUMENT, "stratified initialisation with a native proposal: line_model only");
    if (h->parent) return fail(h, GPF_ERR_STATE, "gpf_initialize on a sub-state view");
    h->generation += 1;
    gpf_status s = set_obs(h, obs, n_obs);
    if (s) return s;
    h->blk_obs_size = 0;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if ((s = hist_begin_step(h, true))) return s;
    const int grid = step_grid(h);
    s = timed(h, GPF_K_STEP, [&] {
        if (prop == 1)      { DISPATCH_MODEL(h, (launch_init_t<MM, 1>(h, grid))); }
        else if (prop == 2) { DISPATCH_MODEL(h, (launch_init_t<MM, 2>(h, grid))); }
        else if (prop == 3) { DISPATCH_MODEL(h, (launch_init_t<MM, 3>(h, grid))); }
        else                { DISPATCH_MODEL(h, (launch_init_t<MM, 0>(h, grid))); }
    });
    if (s) return s;
    h->pending_gather = false; h->pending_fill = false; h->pending_search = false; h->pending_move = false;
    h->pending_packed = false;
    h->max_valid = true;
    GPF_LAUNCH(k_iota, dim3(grid), dim3(BLOCK), 0, h->stream, h->anc, h->n);            // parents = 1:N (initialize.jl:43)
    HIP_TRY(h, hipMemsetAsync(&h->sc->lml_est, 0, sizeof(double), h->stream));                   // log_ml_est = 0.
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->initialized = true;
    h->has_prev = false;
    h->raw_valid = false; h->raw_sum_valid = false;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_initialize(gpf_handle h, const double* obs, int32_t n_obs) { return initialize_impl(h, obs, n_obs, 0); }
// the native proposal a model has: the locally optimal one (lgssm2), the reference tests' fixed proposals (line_model)
static bool proposal_matches(gpf_handle h, int32_t proposal)
{
    if (!h) return true;                                          // reported by the callee
    if (proposal == GPF_PROPOSAL_LOCALLY_OPTIMAL) return h->cfg.model != MODEL_LINE;
    if (proposal == GPF_PROPOSAL_LINE_FIXED) return h->cfg.model == MODEL_LINE;
    return false;
}
gpf_status gpf_initialize_proposal(gpf_handle h, const double* obs, int32_t n_obs, int32_t proposal)
{
    if (!proposal_matches(h, proposal)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown proposal id for this model");
    return initialize_impl(h, obs, n_obs, 1);
}

static gpf_status update_impl(gpf_handle h, const double* obs, int32_t n_obs, int prop)
{
    gpf_status s = check_ready(h, prop == 0);                    // (the plain propagate carries a pending lazy move)
    if (s) return s;
    if (prop == 1 && !model_has_proposal(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no native proposal");
    if (prop == 2 && !model_has_strata(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no discrete latent to stratify over");
    if ((s = set_obs(h, obs, n_obs))) return s;
    h->blk_obs_size = 0;                                         // one observation for all particles again
    if ((s = hist_begin_step(h, false))) return s;
    if (prop != 0 && (s = finish_search(h))) return s;           // (only the plain propagate carries a pending search)
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    if (h->pending_move) {
        // gather (if pending) -> move -> propagate in one launch (k_move_step): the old observation and epoch travel with the move
        const bool rw = h->pm_method == GPF_REJUVENATE_REWEIGHT;
        s = timed(h, GPF_K_STEP, [&] {
            if (rw) { DISPATCH_MODEL(h, (launch_move_step_t<MM, true>(h, grid))); }
            else    { DISPATCH_MODEL(h, (launch_move_step_t<MM, false>(h, grid))); }
        });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        phase_mark(h, GPF_PHASE_COMMIT);
        h->pending_move = false;
        h->pending_gather = false; h->pending_fill = false;
        h->pending_packed = false; h->pend_own = false; h->pend_ring = false;   // (a sharded commit rode in the launch)
        h->max_valid = true;
        h->cur ^= 1;                // (read rows[cur], wrote the other buffer once: the move's and the update's swaps cancel to one)
        h->epoch += 1;
        h->has_prev = true;
        h->raw_valid = false; h->raw_sum_valid = false;
        mutated(h);
        return GPF_OK;
    }
    s = timed(h, GPF_K_STEP, [&] {
        if (prop == 1) {
            if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true, 1>(h, grid))); }
            else      { DISPATCH_MODEL(h, (launch_step_t<MM, false, 1>(h, grid))); }
        } else if (prop == 2) {
            if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true, 2>(h, grid))); }
            else      { DISPATCH_MODEL(h, (launch_step_t<MM, false, 2>(h, grid))); }
        } else {
            if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true>(h, grid))); }
            else      { DISPATCH_MODEL(h, (launch_step_t<MM, false>(h, grid))); }
        }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    phase_mark(h, GPF_PHASE_COMMIT);                         // (gpf_phase_timing: the propagate that committed a sharded resample)
    h->pending_gather = false; h->pending_fill = false;      // a pending resample gather was fused into this step
    h->pending_packed = false; h->pend_own = false; h->pend_ring = false;     // ... or a pending sharded commit
    h->max_valid = true;
    h->cur ^= 1;                    // update_refs! (utils.jl:10-15)
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    mutated(h);
    return view_exit(h);            // sub-state: copy back (utils.jl:17-20)
}

gpf_status gpf_update(gpf_handle h, const double* obs, int32_t n_obs) { return update_impl(h, obs, n_obs, 0); }
gpf_status gpf_update_proposal(gpf_handle h, const double* obs, int32_t n_obs, int32_t proposal)
{
    if (!proposal_matches(h, proposal)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown proposal id for this model");
    return update_impl(h, obs, n_obs, 1);
}

} // extern "C"
namespace gpfh {
// the plain propagate of the current observation (h->args), enqueued behind a launch that leaves an ESS verdict on the device: it returns
// before its first store when the verdict says "resample first" (gpf_step_ess, gpf_shard_step_ess)
gpf_status speculative_step(gpf_filter* h, const GateIn& gate)
{
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    gpf_status s = timed(h, GPF_K_STEP, [&] {
        if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true>(h, grid, &gate))); }
        else      { DISPATCH_MODEL(h, (launch_step_t<MM, false>(h, grid, &gate))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// ran = true: it was the step's pf_update! (update_impl's bookkeeping); false: it returned without touching anything -- its maximum slots go back
void speculative_step_done(gpf_filter* h, bool ran)
{
    if (!ran) { h->mcur ^= 1; return; }
    h->max_valid = true;
    h->cur ^= 1;                        // update_refs! (utils.jl:10-15)
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    mutated(h);
}
} // namespace gpfh
extern "C" {

// One iteration of the reference's README loop (README.md:66-77) in one call:
//     if effective_sample_size(state) < ess_frac * N;  pf_resample!(state, method);  pf_rejuvenate!(state, kern, ...);  end
//     pf_update!(state, new_args, argdiffs, observations)
// The results are those of the four calls in that order, bit for bit (the fall-back below IS that sequence).  What the single call buys is
// the host's round trip: the ESS reduction leaves its verdict on the device as well (k_sum_host<GATE>), the propagate is enqueued
// SPECULATIVELY right behind it and returns at once if the verdict says "resample first" -- so on the steps that do not resample (the
// majority of an ESS-triggered filter's) the GPU never waits for the host's decision; on the others the host, which folds the same sums,
// enqueues resample -> move -> propagate as before.
gpf_status gpf_step_ess(gpf_handle h, const double* obs, int32_t n_obs, double ess_frac, int32_t resample_method, int32_t sort_particles,
                        int32_t check, int32_t rejuvenate_method, int32_t n_iters, int32_t* resampled, int32_t* invalid, double* ess_out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!(ess_frac == ess_frac) || ess_frac < 0.0) return fail(h, GPF_ERR_INVALID_ARGUMENT, "ess_frac must be >= 0");
    if (rejuvenate_method >= 0 && rejuvenate_method != GPF_REJUVENATE_MOVE && rejuvenate_method != GPF_REJUVENATE_REWEIGHT)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Method not recognized.");                        // rejuvenate.jl:25
    if (resample_method != GPF_RESAMPLE_MULTINOMIAL && resample_method != GPF_RESAMPLE_RESIDUAL && resample_method != GPF_RESAMPLE_STRATIFIED &&
        resample_method != GPF_RESAMPLE_MULTINOMIAL_SORTED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");             // resample.jl:28
    if (rejuvenate_method >= 0 && !h->cfg.keep_prev) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate needs keep_prev = 1 (x_{t-1} must travel with the particle)");
    if (resampled) *resampled = 0;
    if (invalid) *invalid = 0;
    const double thr = ess_frac * (double)h->n;
    static const bool spec_off = getenv("GPF_STEP_SPECULATE") && !strcmp(getenv("GPF_STEP_SPECULATE"), "0");   // (A/B measurements, tests: the plain sequence)
    const bool fast = !spec_off && !h->parent && !h->hist_on && h->cfg.n_global == h->n && !h->pending_packed && !h->pending_gather && !h->pending_fill &&
                      !h->pending_search && !h->raw_valid && !h->raw_sum_valid && h->blk_obs_size == 0 && sum_host_ok(h) &&
                      n_obs == model_obs_dim(h->cfg.model) && obs != nullptr;
    if (!fast) {
        // the host-decided sequence (sub-states, shards, trajectory stores, a resample still pending, summaries already at hand, ...)
        double ess = 0.0;
        if ((s = gpf_effective_sample_size(h, &ess))) return s;
        if (ess_out) *ess_out = ess;
        if (ess < thr) {
            if ((s = gpf_resample(h, resample_method, std::nan(""), sort_particles, check, invalid))) return s;
            if (resampled) *resampled = 1;
            if (rejuvenate_method >= 0 && (s = gpf_rejuvenate(h, rejuvenate_method, n_iters, nullptr))) return s;
        }
        return gpf_update(h, obs, n_obs);
    }
    // ---- ESS reduction with the verdict on the device, the propagate speculatively behind it
    const ModelArgs old_args = h->args;                          // (a rejuvenation moves under the CURRENT step's observation)
    const int mcur0 = h->mcur;
    // (an error behind this point leaves the step's observation and the maximum slots as they were: a later call on the handle -- a rejuvenation, say --
    //  must not work under the NEXT step's observation.  The particle buffers of a failed call are undefined either way.)
    auto undo = [&](gpf_status st) { h->args = old_args; h->mcur = mcur0; return st; };
    if ((s = sum_host_launch(h, &thr))) return s;
    const int64_t gate_ticket = h->q_ticket;                     // (this launch's: the live counter moves on with the calls below)
    if ((s = set_obs(h, obs, n_obs))) return undo(s);
    const GateIn gate{h->gate_part + h->gate_cur * GATE_WORDS, thr, &h->sc->gate_go, h->h_gate, gate_ticket, nullptr};
    if ((s = speculative_step(h, gate))) return undo(s);
    int go = 0;
    if ((s = sum_host_fold(h, &thr, &go))) return undo(s);
    if (ess_out) {
        uint64_t hi, lo;
        normalise_Q(h->sum_cache, hi, lo);
        *ess_out = h->sum_cache.flags ? std::nan("") : ess_from(h->sum_cache.S, hi, lo);
    }
    if (!go) {
        speculative_step_done(h, true);                          // the speculative propagate WAS the step's pf_update!
        return sum_gate_check(h, go, gate_ticket);
    }
    // the propagate returned without touching anything: take its maximum slots back, restore the step's observation, and run the sequence
    // from the resample on -- the summary is with the host as after effective_sample_size(state) (a :residual resample skips its weight scan)
    speculative_step_done(h, false);
    h->args = old_args;
    if ((s = gpf_resample(h, resample_method, std::nan(""), sort_particles, check, invalid))) return s;
    if (resampled) *resampled = 1;
    if (rejuvenate_method >= 0 && (s = gpf_rejuvenate(h, rejuvenate_method, n_iters, nullptr))) return s;
    if ((s = gpf_update(h, obs, n_obs))) return s;
    return sum_gate_check(h, go, gate_ticket);                   // (after the sequence is enqueued: the check is a safety net, not a dependency)
}

// stratified initialisation / update: the strata are values of the model's discrete latent
} // extern "C"
namespace gpfh {
gpf_status set_strata(gpf_handle h, const double* values, int32_t n_strata, int32_t interleaved)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!values || n_strata < 1 || n_strata > MAX_STRATA) return fail(h, GPF_ERR_INVALID_ARGUMENT, "need 1..8 strata");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "stratified initialisation / update of a sharded filter is not supported");
    for (int k = 0; k < MAX_STRATA; ++k) h->args.strata[k] = k < n_strata ? values[k] : 0.0;
    h->args.n_strata = n_strata; h->args.interleaved = interleaved != 0;
    h->args.logK = log_((double)n_strata);
    return GPF_OK;
}
} // namespace gpfh
extern "C" {
gpf_status gpf_initialize_strata(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved)
{
    gpf_status s = set_strata(h, values, n_strata, interleaved);
    return s ? s : initialize_impl(h, obs, n_obs, 2);
}
gpf_status gpf_initialize_strata_proposal(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved,
                                          int32_t proposal)
{
    if (!proposal_matches(h, proposal) || proposal != GPF_PROPOSAL_LINE_FIXED) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown proposal id for this model");
    gpf_status s = set_strata(h, values, n_strata, interleaved);
    return s ? s : initialize_impl(h, obs, n_obs, 3);
}
gpf_status gpf_update_strata(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved)
{
    gpf_status s = set_strata(h, values, n_strata, interleaved);
    return s ? s : update_impl(h, obs, n_obs, 2);
}

static gpf_status rejuvenate_impl(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted, bool with_proposal);
gpf_status gpf_rejuvenate(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted)
{
    return rejuvenate_impl(h, method, n_iters, n_accepted, false);
}
gpf_status gpf_rejuvenate_proposal(gpf_handle h, int32_t proposal, const double* params, int32_t n_params, int32_t n_iters)
{
    return gpf_rejuvenate_with_proposal(h, GPF_REJUVENATE_REWEIGHT, proposal, params, n_params, n_iters, nullptr);
}
gpf_status gpf_rejuvenate_with_proposal(gpf_handle h, int32_t method, int32_t proposal, const double* params, int32_t n_params, int32_t n_iters, uint64_t* n_accepted)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (n_params < 0 || n_params > 4 || (n_params > 0 && !params)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad proposal parameters");
    const bool ok = (proposal == GPF_MOVE_PROPOSAL_LOCALLY_OPTIMAL && h->cfg.model == MODEL_LGSSM2 && n_params == 0) ||
                    (proposal == GPF_MOVE_PROPOSAL_LINE_OUTLIER && h->cfg.model == MODEL_LINE && n_params == 3);
    if (!ok || !model_has_move_proposal(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown move proposal for this model (or wrong parameter count)");
    for (int i = 0; i < 4; ++i) h->args.q[i] = i < n_params ? params[i] : 0.0;
    return rejuvenate_impl(h, method, n_iters, n_accepted, true);
}
static gpf_status rejuvenate_impl(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted, bool with_proposal)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->parent && h->parent->blk_obs_size != 0)
        return fail(h, GPF_ERR_STATE, "rejuvenation of a view after a block-wise update of its filter: use gpf_rejuvenate_blocks on the filter");
    if (h->blk_obs_size < 0)
        return fail(h, GPF_ERR_STATE, "the filter was resized after a block-wise update: no current observation until the next update");
    if (h->blk_obs_size > 0 && !h->parent) {                     // the latest observations are per block (gpf_update_blocks)
        if (with_proposal) return fail(h, GPF_ERR_STATE, "proposal moves are not available after a block-wise update");
        return gpf_rejuvenate_blocks(h, method, n_iters, 0, n_accepted);
    }
    if (method != GPF_REJUVENATE_MOVE && method != GPF_REJUVENATE_REWEIGHT)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Method not recognized.");                        // rejuvenate.jl:25
    if (!h->cfg.keep_prev) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate needs keep_prev = 1 (x_{t-1} must travel with the particle)");
    if (n_iters < 0) return fail(h, GPF_ERR_INVALID_ARGUMENT, "n_iters < 0");
    // lazy move: a plain selection move of a whole filter (or of a whole shard) whose acceptance count nobody asked for waits for the pf_update! that
    // follows (k_move_step); its epoch is consumed now.  A pending sharded commit waits with it: k_move_step reads the own hits through the
    // ancestor array, the receive window and the packed entries itself (no k_gather_own + k_commit_packed in front of the move)
    const bool lazy = h->lazy_move && !with_proposal && !n_accepted && !h->parent && !h->hist_on;
    if (h->pending_packed && !lazy && (s = materialize(h))) return s;     // sharded deferred commit: scatter first
    if (h->pending_fill && (s = materialize(h))) return s;       // (the move kernel's fused gather assumes incoming weights 0)
    if ((s = finish_search(h))) return s;                        // (a lazy multinomial resample: the move kernel reads the ancestor array)
    if (lazy) {
        h->pending_move = true; h->pm_method = method; h->pm_iters = n_iters; h->pm_epoch = h->epoch; h->pm_args = h->args;
        h->epoch += 1;
        return GPF_OK;
    }
    const bool fused_gather = h->pending_gather;                 // a pending resample gather rides on the move kernel
    const int grid = move_grid(h);
    s = timed(h, GPF_K_MOVE, [&] {
        if (with_proposal && method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_prop_t<MM, true>(h, grid, n_iters))); }
        else if (with_proposal)                     { DISPATCH_MODEL(h, (launch_move_prop_t<MM, false>(h, grid, n_iters))); }
        else if (method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_t<MM, true>(h, grid, n_iters, h->args, h->epoch))); }
        else                                        { DISPATCH_MODEL(h, (launch_move_t<MM, false>(h, grid, n_iters, h->args, h->epoch))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    h->epoch += 1;
    if (fused_gather) { h->pending_gather = false; h->pending_fill = false; h->max_valid = false; }   // log-weights are all 0 now (resample.jl:195)
    if (method == GPF_REJUVENATE_REWEIGHT) { h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = true; }
    mutated(h);
    if ((s = view_exit(h))) return s;
    if (n_accepted) {
        if (method == GPF_REJUVENATE_REWEIGHT) *n_accepted = (uint64_t)h->n * (uint64_t)n_iters;     // every particle moves (rejuvenate.jl:81-86)
        else {
            GPF_LAUNCH(k_sum_accepts, dim3(1), dim3(BLOCK), 0, h->stream, h->acc_part, grid, reinterpret_cast<unsigned long long*>(&h->sc->n_accept));
            HIP_TRY(h, hipGetLastError());
            if ((s = fetch_scalars(h))) return s;
            *n_accepted = h->h_sc->n_accept;
        }
    }
    return GPF_OK;
}

gpf_status gpf_get_log_weights(gpf_handle h, double* out, int64_t n)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if ((s = materialize(h))) return s;
    return copy_out(h, h->lw, out, (size_t)n * sizeof(double));
}

gpf_status gpf_get_parents(gpf_handle h, int64_t* out, int64_t n)
{
    gpf_status s0 = check_ready(h);                             // views: generation check + pointers; device; initialised
    if (s0) return s0;
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if (h->pending_packed) { gpf_status s = materialize(h); if (s) return s; }   // a deferred sharded commit also carries the parents
    { gpf_status s = finish_search(h); if (s) return s; }                         // a lazy multinomial resample: its ancestors now
    GPF_LAUNCH(k_parents, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->anc, h->n, reinterpret_cast<int64_t*>(h->dtmp));
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(int64_t));
}

gpf_status gpf_state_dim(gpf_handle h, int32_t* dim, int32_t* row_width_out)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (dim) *dim = h->d;
    if (row_width_out) *row_width_out = h->W;
    return GPF_OK;
}

gpf_status gpf_get_column(gpf_handle h, int32_t column, double* out, int64_t n)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n != h->n || column < 0 || column >= h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad column/output");
    if ((s = materialize(h))) return s;
    GPF_LAUNCH(k_extract_column, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->rows[h->cur], h->W, column, h->n, h->dtmp);
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(double));
}

gpf_status gpf_get_rows(gpf_handle h, double* out, int64_t n_doubles)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n_doubles != h->n * h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if ((s = materialize(h))) return s;
    return copy_out(h, h->rows[h->cur], out, (size_t)n_doubles * sizeof(double));
}

gpf_status gpf_set_rows(gpf_handle h, const double* rows, int64_t n_doubles)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!rows || n_doubles != h->n * h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad input array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->pending_move) { gpf_status s = finish_move(h); if (s) return s; }
    if (h->parent) { gpf_status s = view_enter(h); if (s) return s; }
    { gpf_status s = materialize(h); if (s) return s; }
    HIP_TRY(h, hipMemcpyAsync(h->rows[h->cur], rows, (size_t)n_doubles * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->parent && h->view_step != 1) { gpf_status s = view_exit(h); if (s) return s; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->initialized = true;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_set_log_weights(gpf_handle h, const double* lw, int64_t n)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!lw || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad input array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->pending_move) { gpf_status s = finish_move(h); if (s) return s; }
    if (h->parent) { gpf_status s = view_enter(h); if (s) return s; h->parent->raw_valid = false; h->parent->raw_sum_valid = false; h->parent->max_valid = false; }
    { gpf_status s = materialize(h); if (s) return s; }
    h->max_valid = false;
    HIP_TRY(h, hipMemcpyAsync(h->lw, lw, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->parent && h->view_step != 1) { gpf_status s = view_exit(h); if (s) return s; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->raw_valid = false; h->raw_sum_valid = false;
    h->initialized = true;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_kernel_timing(gpf_handle h, int32_t id, int32_t enable)
{
    if (!h || id < 0 || id >= GPF_K_COUNT) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad kernel id");
    Timer& t = h->timers[id];
    for (auto& e : t.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    t.ev.clear();
    t.on = enable != 0;
    return GPF_OK;
}

gpf_status gpf_kernel_time(gpf_handle h, int32_t id, double* total_ms, int64_t* launches)
{
    if (!h || id < 0 || id >= GPF_K_COUNT) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad kernel id");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (auto& e : h->timers[id].ev) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, e.first, e.second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)h->timers[id].ev.size();
    return GPF_OK;
}

gpf_status gpf_debug_math(gpf_handle h, int32_t which, const double* a, const double* b, int64_t n, double* out, double* out2)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!a || !out || n < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arrays");
    double *da = nullptr, *db = nullptr, *d1 = nullptr, *d2 = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_TRY(h, hipMalloc(&da, bytes)); HIP_TRY(h, hipMalloc(&db, bytes));
    HIP_TRY(h, hipMalloc(&d1, bytes)); HIP_TRY(h, hipMalloc(&d2, bytes));
    HIP_TRY(h, hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(db, b ? b : a, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(d2, 0, bytes, h->stream));
    GPF_LAUNCH(k_debug_math, dim3(grid_for(h, n, 8)), dim3(BLOCK), 0, h->stream, which, da, db, n, h->cfg.seed, h->epoch,
                       (uint32_t)TAG_UPDATE, d1, d2);
    HIP_TRY(h, hipMemcpyAsync(out, d1, bytes, hipMemcpyDeviceToHost, h->stream));
    if (out2) HIP_TRY(h, hipMemcpyAsync(out2, d2, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(da); hipFree(db); hipFree(d1); hipFree(d2);
    return GPF_OK;
}

gpf_status gpf_debug_levels(gpf_handle h, int32_t which, void* out, int64_t* n_bytes)
{
    if (!h || !out || !n_bytes) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    const int64_t nt = h->ntiles;
    const int logg = multi_logg(nt);
    const void* src = nullptr; int64_t bytes = 0;
    switch (which) {
        case 0: src = h->cdf[0]; bytes = nt * TILE * 8; break;
        case 1: src = h->t16[0]; bytes = nt * (TILE / 16) * 8; break;
        case 2: src = h->t256[0]; bytes = nt * (TILE / 256) * 8; break;
        case 3: src = k32_of(h->t256[0], nt); bytes = nt * (TILE / 32) * 4; break;
        case 4: src = off16_of(h->t256[0], nt); bytes = logg >= 0 ? nt * TILE * 2 : 0; break;
        case 5: src = coarse_of(h->t256[0], nt); bytes = logg >= 0 ? (nt * TILE / (4 << logg)) * 2 : 0; break;
        default: return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad level");
    }
    if (bytes > *n_bytes) return fail(h, GPF_ERR_INVALID_ARGUMENT, "output too small");
    *n_bytes = bytes;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return bytes ? copy_out(h, src, out, (size_t)bytes) : GPF_OK;
}


} // extern "C"

// =================================================================================== host scalar spec
extern "C" {
int32_t gpf_host_fix_K(int64_t n_global) { return fix_K(n_global); }
int32_t gpf_host_gamma_E(int64_t n_tiles) { return gamma_E(n_tiles); }
uint64_t gpf_host_div128(uint64_t P, uint64_t den) { return div128(P, div128_setup(den)); }
uint64_t gpf_host_muldiv128(uint64_t p, uint64_t W, uint64_t den) { return muldiv128(p, W, div128_setup(den)); }
uint64_t gpf_host_gamma_tile(uint64_t seed, uint32_t gid, uint32_t epoch, int64_t shape, int32_t Eg) { return gamma_tile(seed, gid, epoch, shape, Eg); }
double gpf_host_log(double x) { return log_(x); }
double gpf_host_lse(double m, uint64_t S, int32_t K, int32_t flags)
{
    int f = flags;
    if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    return lse_from(m, S, K, f);
}
double gpf_host_ess(uint64_t S, uint64_t Q_hi, uint64_t Q_lo) { return ess_from(S, Q_hi, Q_lo); }
void gpf_host_math(int32_t which, const double* a, const double* b, int64_t n, double* out, double* out2)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = exp_(a[i]); break;
            case 1: out[i] = log_(a[i]); break;
            case 2: sincos2pi(a[i], out[i], out2[i]); break;
            case 3: out[i] = atan2_(a[i], b[i]); break;
            case 4: out[i] = sqrt_(a[i]); break;
            case 5: out[i] = a[i] / b[i]; break;
            case 7: out[i] = neglog_u52(d2u(a[i])); break;
            default: out[i] = 0.0;
        }
    }
}

} // extern "C"

#ifdef GPF_DBG_STRAT
extern "C" int gpf_debug_strat(unsigned long long* out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpf::g_dbg_strat), (size_t)n_words * sizeof(unsigned long long));
}
#endif

#ifdef GPF_DBG_SORT
extern "C" int gpf_debug_sort_buckets(unsigned long long* out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpf::g_dbg_bk), (size_t)n_words * sizeof(unsigned long long));
}
extern "C" int gpf_debug_sort(unsigned long long* out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpf::g_dbg_sort), (size_t)n_words * sizeof(unsigned long long));
}
#endif
