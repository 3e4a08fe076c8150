// libgpf_shard.hip -- multi-GPU: the shard-level phases of a sharded pf_resample!, the library's own RCCL communicator,
// the shard mailboxes, gpf_shard_resample in one call.
#include "gpf_host.hpp"
#include <chrono>

using namespace gpf;
using namespace gpfh;

template <int METHOD>
static void launch_push(gpf_filter* h, const PushArgs& a, int grid, size_t lds, const CdfLevels& lw_, const CdfLevels& lc_, int64_t capacity, double* out)
{
    switch (h->W) {
        case 2: GPF_LAUNCH((k_push<METHOD, 2>), dim3(grid), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out); break;
        case 4: GPF_LAUNCH((k_push<METHOD, 4>), dim3(grid), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out); break;
        case 8: GPF_LAUNCH((k_push<METHOD, 8>), dim3(grid), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out); break;
    }
}

static void launch_push_multi(gpf_filter* h, const PushArgs& a, int grid, const CdfLevels& lw_, int64_t capacity, double* out)
{
#define GPF_PN(LG, WW) GPF_LAUNCH((k_push_multi<LG, WW>), dim3(grid), dim3(SBLOCK), multi_lds_bytes(h->ntiles, LG), h->stream, a, lw_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out)
    if (lw_.logg == 0) { switch (h->W) { case 2: GPF_PN(0, 2); break; case 4: GPF_PN(0, 4); break; case 8: GPF_PN(0, 8); break; } }
    else               { switch (h->W) { case 2: GPF_PN(1, 2); break; case 4: GPF_PN(1, 4); break; case 8: GPF_PN(1, 8); break; } }
#undef GPF_PN
}


namespace gpfh {
// what gpf_create asks of this unit's kernels: the dynamic-LDS ceilings of the push kernels
gpf_status shard_device_setup(gpf_filter* h)
{
    const int max_dyn = (int)((lds_pad(LDS_TILE_TABLE) + 4) * sizeof(uint64_t));
#define GPF_PUSH_ATTR(M, W) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_push<M, W>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn))
        GPF_PUSH_ATTR(0, 2); GPF_PUSH_ATTR(0, 4); GPF_PUSH_ATTR(0, 8);
        GPF_PUSH_ATTR(1, 2); GPF_PUSH_ATTR(1, 4); GPF_PUSH_ATTR(1, 8);
#define GPF_PN_ATTR(LG, W) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_push_multi<LG, W>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET))
        GPF_PN_ATTR(0, 2); GPF_PN_ATTR(0, 4); GPF_PN_ATTR(0, 8); GPF_PN_ATTR(1, 2); GPF_PN_ATTR(1, 4); GPF_PN_ATTR(1, 8);
#undef GPF_PUSH_ATTR
#undef GPF_PN_ATTR
    return GPF_OK;
}
} // namespace gpfh

extern "C" {

// =================================================================================== shard-level ABI
static gpf_status shard_ready(gpf_handle h)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->comm_poisoned)
        return fail(h, GPF_ERR_STATE, "an earlier sharded call failed on this rank after its exchange rounds had begun: the communicator is out of step with its peers (tear the job down)");
    return GPF_OK;
}

static gpf_status shard_max_slots(gpf_handle h);
static gpf_status ensure_shard_counts(gpf_handle h)
{
    if (h->shard_counts) return GPF_OK;
    HIP_TRY(h, hipMalloc(&h->shard_counts, (size_t)2 * MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t)));
    HIP_TRY(h, hipMemsetAsync(h->shard_counts, 0, (size_t)2 * MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t), h->stream));
    HIP_TRY(h, hipHostMalloc(&h->h_shard_counts, (size_t)(2 * MAX_SHARDS + 1) * sizeof(int64_t)));
    h->h_shard_counts[2 * MAX_SHARDS] = 0;
    return GPF_OK;
}
gpf_status gpf_shard_weight_max(gpf_handle h, double* out2)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out2) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if ((s = shard_max_slots(h))) return s;
    GPF_LAUNCH(k_pack_mflags, dim3(1), dim3(BLOCK), 0, h->stream, h->mslots[h->mcur], out2, mb_begin(h, MB_MF));
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// the maximum slots describe the weights to summarise (the producer's slots, or one k_max_partial pass)
static gpf_status shard_max_slots(gpf_handle h)
{
    gpf_status s;
    if ((s = materialize(h))) return s;
    const PrioView pv = h->sum_pv_set ? h->sum_pv : raw_view(h);  // (the engine's prioritised resample summarises alpha lw and log_ws too)
    if (!h->max_valid || h->sum_pv_set) {
        const int gp = (int)std::min<int64_t>(MAX_PARTIALS, (h->n + BLOCK - 1) / BLOCK);
        s = timed(h, GPF_K_MAX, [&] {
            GPF_LAUNCH(k_max_partial, dim3(gp), dim3(BLOCK), 0, h->stream, pv, h->n, next_slots(h));
        });
        if (s) return s;
        h->max_valid = !h->sum_pv_set;
    }
    return GPF_OK;
}

gpf_status gpf_shard_weight_scan(gpf_handle h, const double* mf_all, int32_t G, int32_t want_q, int64_t* out5)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!mf_all || !out5 || G < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if ((s = ensure_shard_counts(h))) return s;
    InFixQ in{h->sum_pv_set ? h->sum_pv : raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
    WSum* const slot = h->sum_slot ? h->sum_slot : &h->sc->raw;
    const bool want_cdf = !h->sum_no_cdf;
    const int gs = wscan_grid(h);
    // the scan folds the gathered (max, flags) pairs itself and writes the shard total straight into out5[0]; it also
    // publishes the global validity flags to pinned host memory (gpf_shard_flags)
    if (!h->h_flags) { HIP_TRY(h, hipHostMalloc(&h->h_flags, 2 * sizeof(int64_t))); h->h_flags[0] = h->h_flags[1] = 0; }
    h->flag_ticket += 1;
    // shard mailboxes: the scan waits for the ranks' (max, flags) entries itself and the kernel that ends up with the shard's
    // {S, limbs} stores them into every peer's mailbox (the scan's last workgroup, or k_export_q when the limbs are wanted)
    ScanExtras ex{h->shard_counts, h->h_flags, h->flag_ticket, 0};
    ex.zero_stride = COUNT_STRIDE;
    if (h->fuse_mf_out) {                                        // (library engine: the scan produces and pushes the first summary itself)
        ex.fuse_mf = 1; ex.mf_me = h->comm_rank; ex.mf_out = h->fuse_mf_out; ex.mf_push = mb_begin(h, MB_MF);
        if (h->mb_active && h->mb_engine) mf_all = static_cast<const double*>(mb_gathered(h, MB_MF));   // the round that has just begun
    }
    ex.wait = mb_wait(h, MB_MF);
    const MboxPush tot_push = mb_begin(h, MB_TOT);
    h->splan_done = false;
    static const bool splan_off = getenv("GPF_SHARD_PLAN_IN_SCAN") && !strcmp(getenv("GPF_SHARD_PLAN_IN_SCAN"), "0");   // (A/B measurements, tests of k_strat_plan)
    if (h->splan_ride && !splan_off && !want_q && h->mb_active && h->mb_engine && (int)G == h->comm_world) {
        // a stratified resample: the plan needs nothing but the G shard totals of THIS round, and the scan's workgroup that ends up with this shard's total
        // pushes the last one of them -- it derives the plan right there (no k_strat_plan launch, no gap in front of the merge kernel)
        if (!h->shard_plan) HIP_TRY(h, hipMalloc(&h->shard_plan, sizeof(ShardPlan)));
        h->push_ticket += 1;
        int64_t* const host_counts = ((h->own_direct && G == 1) || h->ring_now) ? nullptr : h->h_shard_counts;   // (as gpf_shard_push_count)
        ex.splan = StratPlanJob{h->shard_plan, h->cfg.seed, h->epoch, (int)G, h->comm_rank, h->cfg.n_global, static_cast<const int64_t*>(mb_gathered(h, MB_TOT)), mb_wait(h, MB_TOT),
                                h->shard_counts, host_counts, h->push_ticket, h->ring_now ? h->tr_dev : nullptr, (int)(2 * MAX_SHARDS * COUNT_STRIDE)};
        ex.zero128 = nullptr;                                     // (the plan's workgroup clears the counters itself: two workgroups must not store to them)
        h->splan_done = true;
    }
    if (want_q) {
        if ((s = scan_launch_shard(h, 4, in, (int)G, slot, want_cdf, reinterpret_cast<uint64_t*>(out5), mf_all, ex))) return s;
        GPF_LAUNCH(k_export_q, dim3(1), dim3(BLOCK), 0, h->stream, h->blockQ, gs, out5, tot_push);
    } else {
        ex.push = tot_push;
        if ((s = scan_launch_shard(h, 3, in, (int)G, slot, want_cdf, reinterpret_cast<uint64_t*>(out5), mf_all, ex))) return s;
    }
    HIP_TRY(h, hipGetLastError());
    h->raw_valid = false; h->raw_sum_valid = false;            // sc->raw holds the GLOBAL max but no sum: not the unsharded summary
    return GPF_OK;
}

gpf_status gpf_shard_flags(gpf_handle h, int32_t* flags_out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!flags_out || !h->h_flags || h->flag_ticket == 0) return fail(h, GPF_ERR_STATE, "gpf_shard_flags needs gpf_shard_weight_scan first");
    if ((s = wait_ticket(h, h->h_flags + 1, h->flag_ticket, "weight scan flags"))) return s;
    *flags_out = (int32_t)h->h_flags[0];
    return GPF_OK;
}

gpf_status gpf_shard_residual_scan(gpf_handle h, const int64_t* tot_all, int32_t G, int64_t* out2)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!tot_all || !out2 || G < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    // global S into sc->prio (the local CDF in cdf[0] stays local)
    GPF_LAUNCH(k_set_global, dim3(1), dim3(64), 0, h->stream, tot_all, (int)G, &h->sc->prio, mb_wait(h, MB_TOT));
    if ((s = residual_scans(h, &h->sc->prio, h->cfg.n_global))) return s;
    GPF_LAUNCH(k_export_residual, dim3(1), dim3(64), 0, h->stream, h->sc, out2, mb_begin(h, MB_CR), nullptr, 0);
    HIP_TRY(h, hipGetLastError());
    h->residual_scanned = true;
    return GPF_OK;
}
// ... the same WITHOUT a weight scan in front (shard_resample_impl: the one-launch summary reduction of the ESS read has left the global S with the
// host and the gathered (max, flags) in the mailbox -- the unsharded filter's direct form, k_scan_residual2<DIRECT>): the weights are converted in the
// residual scan itself, the exchange counters the weight scan would have cleared are cleared by k_export_residual
static gpf_status shard_residual_scan_direct(gpf_handle h, const double* mf_all, uint64_t S_global, int32_t G, int64_t* out2)
{
    gpf_status s = ensure_shard_counts(h);
    if (s) return s;
    if ((s = materialize(h))) return s;
    const ResidDirect rd{h->lw, 0.0, 0, h->K, S_global, &h->sc->prio, mf_all, G, (int32_t)(h->mb_active && h->mb_engine)};
    if ((s = residual_scans(h, &h->sc->prio, h->cfg.n_global, nullptr, &rd))) return s;
    GPF_LAUNCH(k_export_residual, dim3(1), dim3(64), 0, h->stream, h->sc, out2, mb_begin(h, MB_CR), h->shard_counts, (int)(2 * MAX_SHARDS * COUNT_STRIDE));
    HIP_TRY(h, hipGetLastError());
    h->residual_scanned = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    return GPF_OK;
}

// fill the argument block of the push kernels; bounds: HOST int64[G+1], first global slot of every shard
static gpf_status push_args(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                            const int64_t* bounds, PushArgs& a)
{
    if ((method < 0 || method > 2) && method != GPF_RESAMPLE_MULTINOMIAL_SORTED) return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");
    if (!tot_all || !bounds || G < 1 || G > MAX_SHARDS || me < 0 || me >= G || (method == GPF_RESAMPLE_RESIDUAL && !cr_all))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (bounds[0] != 0 || bounds[G] != h->cfg.n_global || bounds[me] != h->cfg.gid0 || bounds[me + 1] != h->cfg.gid0 + h->n)
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "shard bounds do not match this filter's global range");
    a.seed = h->cfg.seed; a.epoch = h->epoch; a.n_global = h->cfg.n_global; a.G = G; a.me = me;
    int64_t c = 0;
    for (int g = 0; g < G; ++g) {
        if (bounds[g + 1] < bounds[g]) return fail(h, GPF_ERR_INVALID_ARGUMENT, "shard bounds must be non-decreasing");
        a.bounds[g] = bounds[g]; a.chunk0[g] = c;
        c += (bounds[g + 1] - bounds[g] + PUSH_CHUNK - 1) / PUSH_CHUNK;
    }
    a.bounds[G] = bounds[G]; a.chunk0[G] = c; a.nchunks = c;
    if (h->cfg.n_global > h->push_cap) {                       // staging list: one 16-byte entry per GLOBAL output slot at most
        if (h->push_stage) (void)hipFree(h->push_stage);
        h->push_stage = nullptr; h->push_cap = 0;
        HIP_TRY(h, hipMalloc(&h->push_stage, (size_t)h->cfg.n_global * sizeof(ulonglong2)));
        h->push_cap = h->cfg.n_global;
    }
    a.extra = h->push_extra; a.pv = h->push_pv; a.skip_own = 0;
    a.wait_tot = mb_wait(h, MB_TOT);
    a.wait_cr = method == GPF_RESAMPLE_RESIDUAL ? mb_wait(h, MB_CR) : MboxWait{};
    a.tot_all = tot_all; a.cr_all = method == GPF_RESAMPLE_RESIDUAL ? cr_all : nullptr; a.stage = h->push_stage; a.counts = h->shard_counts; a.host_counts = h->h_shard_counts; a.ticket = h->push_ticket;
    a.traffic = h->ring_now ? h->tr_dev : nullptr;                // (window exchange: nobody on the host reads the counts; the plan kernel keeps the traffic statistics)
    a.ring = RingOut{nullptr, 0, 0};
    return GPF_OK;
}

gpf_status gpf_shard_push_count(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                                const int64_t* bounds)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!h->shard_counts) return fail(h, GPF_ERR_STATE, "gpf_shard_push_count needs gpf_shard_weight_scan of the same resample first");
    PushArgs a;
    if ((s = push_args(h, method, tot_all, cr_all, G, me, bounds, a))) return s;    // the counters were cleared by the weight scan
    h->counts_published = false;
    if (method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        // sorted uniforms: ascending targets, so the plan is again ONE served slot range per shard (k_sorted_plan); every shard draws ALL tile
        // totals itself -- with its weight scan when the library engine prepared the job, else now
        if (!h->shard_plan) HIP_TRY(h, hipMalloc(&h->shard_plan, sizeof(ShardPlan)));
        if (!h->splan_F) {
            HIP_TRY(h, hipMalloc(&h->splan_F, (MAX_SHARDS + 1) * sizeof(int64_t)));
            HIP_TRY(h, hipMalloc(&h->splan_arrive, sizeof(unsigned int)));
            HIP_TRY(h, hipMemsetAsync(h->splan_arrive, 0, sizeof(unsigned int), h->stream));
        }
        const int64_t ntl = (h->cfg.n_global + SP_TILE - 1) / SP_TILE;
        if (!(h->sp_cap >= ntl + 1 && h->sp_job.g == h->sp_g && h->sp_job.epoch == h->epoch && h->sp_job.n == h->cfg.n_global && h->sp_job.gid0 == 0 && h->sp_job.ntl == ntl)
            && (s = sorted_job_prepare(h, 0, h->cfg.n_global))) return s;
        if ((s = sorted_gammas_finish(h, ntl > SP_DIRECT_TILES))) return s;
        h->push_ticket += 1;
        a.ticket = h->push_ticket;
        if ((h->own_direct && G == 1) || h->ring_now) a.host_counts = nullptr;      // (one shard, own-direct -- or the window exchange --: nobody waits for the counts, no system-scope publish)
        const SortedPlanJob job{h->sp_g, h->sp_vlo, ntl, ntl > SP_DIRECT_TILES ? 1 : 0, h->splan_F, h->splan_arrive};
        s = timed(h, GPF_K_SEARCH, [&] { GPF_LAUNCH(k_sorted_plan, dim3((unsigned)(G > 1 ? G - 1 : 1)), dim3(MBLOCK), 0, h->stream, a, h->shard_plan, job); });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        h->counts_published = a.host_counts != nullptr;         // (else gpf_shard_counts copies them from the device)
        h->push_counted = true;
        return GPF_OK;
    }
    if (method == GPF_RESAMPLE_STRATIFIED && h->splan_done) {
        // the plan rode in the weight scan's launch (gpf_shard_weight_scan, ScanExtras::splan): nothing to launch
        h->splan_done = false;
        h->counts_published = !((h->own_direct && G == 1) || h->ring_now);
        h->push_counted = true;
        return GPF_OK;
    }
    if (method == GPF_RESAMPLE_STRATIFIED) {
        // contiguous strata x contiguous shard ranges: the plan (served slot range, exchange counts) is closed-form; the
        // counts go to the host right away
        if (!h->shard_plan) HIP_TRY(h, hipMalloc(&h->shard_plan, sizeof(ShardPlan)));
        h->push_ticket += 1;
        a.ticket = h->push_ticket;
        if ((h->own_direct && G == 1) || h->ring_now) a.host_counts = nullptr;      // (as above)
        s = timed(h, GPF_K_SEARCH, [&] { GPF_LAUNCH(k_strat_plan, dim3(1), dim3(128), 0, h->stream, a, h->shard_plan); });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        h->counts_published = a.host_counts != nullptr;         // (else gpf_shard_counts copies them from the device)
        h->push_counted = true;
        return GPF_OK;
    }
#ifndef PUSH_SCAN_BLOCKS_PER_CU
#define PUSH_SCAN_BLOCKS_PER_CU 8
#endif
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(a.nchunks, (int64_t)h->n_cu * PUSH_SCAN_BLOCKS_PER_CU));
    a.skip_own = h->own_direct ? 1 : 0;
    if (h->own_direct) {
        // the shard's own slots: ancestors in place (k_search_own), nothing staged or packed for them; pass 1 walks the other shards' slots
        // only -- with one shard there are none
        const int gso = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + GPF_MULTI_NS * SBLOCK - 1) / (GPF_MULTI_NS * SBLOCK), (int64_t)h->n_cu));
        if (method == GPF_RESAMPLE_RESIDUAL) {
            if (!h->residual_scanned) return fail(h, GPF_ERR_STATE, "residual push needs gpf_shard_residual_scan first");
            const CdfLevels lw_ = levels(h, 2), lc_ = levels(h, 1);
            const size_t lds = search_lds_bytes(h->ntiles, 2);
            s = timed(h, GPF_K_SEARCH, [&] {
                GPF_LAUNCH(k_search_own_res, dim3(gso), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->anc);
            });
            if (s) return s;
            if (G > 1) GPF_LAUNCH((k_push_scan<1>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
            HIP_TRY(h, hipGetLastError());
            h->push_counted = true;
            return GPF_OK;
        }
        const CdfLevels lw_ = levels(h, 0);
        const size_t lds = multi_lds_bytes(h->ntiles, lw_.logg);
        s = timed(h, GPF_K_SEARCH, [&] {
            if (lw_.logg == 0) GPF_LAUNCH((k_search_own<0>), dim3(gso), dim3(SBLOCK), lds, h->stream, a, lw_, h->n, h->ntiles, h->cfg.gid0, h->anc);
            else               GPF_LAUNCH((k_search_own<1>), dim3(gso), dim3(SBLOCK), lds, h->stream, a, lw_, h->n, h->ntiles, h->cfg.gid0, h->anc);
        });
        if (s) return s;
        if (G > 1) GPF_LAUNCH((k_push_scan<0>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
        HIP_TRY(h, hipGetLastError());
        h->push_counted = true;
        return GPF_OK;
    }
    s = timed(h, GPF_K_SEARCH, [&] {
        if (method == GPF_RESAMPLE_MULTINOMIAL) GPF_LAUNCH((k_push_scan<0>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
        else                                    GPF_LAUNCH((k_push_scan<1>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->push_counted = true;
    return GPF_OK;
}

gpf_status gpf_shard_counts(gpf_handle h, int32_t G, int64_t* host_counts)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!host_counts || G < 1 || G > MAX_SHARDS || !h->shard_counts || !h->push_counted) return fail(h, GPF_ERR_STATE, "no counted resample");
    if (h->counts_published) {
        // k_push publishes the counts to pinned host memory when it STARTS: poll the ticket (the kernel keeps running)
        if ((s = wait_ticket(h, h->h_shard_counts + 2 * MAX_SHARDS, h->push_ticket, "push counts"))) return s;
    } else {
        // (the counters sit COUNT_STRIDE words apart on the device, densely in the mirror; the receive counters' stripes behind them)
        std::vector<int64_t> all((size_t)2 * MAX_SHARDS * COUNT_STRIDE);
        HIP_TRY(h, hipMemcpyAsync(all.data(), h->shard_counts, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (int g = 0; g < G; ++g) {
            h->h_shard_counts[g] = all[(size_t)g * COUNT_STRIDE];
            int64_t v = 0;
            for (int st = 0; st < recv_stripes(G); ++st) v += all[(size_t)recv_counter_index(G, g, st) * COUNT_STRIDE];
            h->h_shard_counts[G + g] = v;
        }
    }
    for (int g = 0; g < G; ++g) { host_counts[g] = h->h_shard_counts[g]; host_counts[G + g] = h->h_shard_counts[G + g]; }
    return GPF_OK;
}

gpf_status gpf_shard_push(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                          const int64_t* bounds, int64_t capacity, double* packed_out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!h->push_counted) return fail(h, GPF_ERR_STATE, "gpf_shard_push needs gpf_shard_push_count of the same resample first");
    if (capacity < 0 || (capacity > 0 && !packed_out && !h->ring_now)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    PushArgs a;
    if (method == GPF_RESAMPLE_STRATIFIED || method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        const bool su = method == GPF_RESAMPLE_MULTINOMIAL_SORTED;
        if (!h->shard_plan) return fail(h, GPF_ERR_STATE, "gpf_shard_push needs gpf_shard_push_count of the same resample first");
        if ((s = push_args(h, method, tot_all, cr_all, G, me, bounds, a))) return s;
        if (capacity == 0) return GPF_OK;
        // ancestors of the served slots (a streaming merge over the shard's own CDF) and, in the same kernel, their rows packed in
        // slot order; the grid is sized for the send buffer and stops at the served count, which only the device knows
        const int64_t cap = std::min<int64_t>(capacity, h->cfg.n_global);
        SearchArgs sa{};
        sa.w = levels(h, 0); sa.c = sa.w; sa.ntiles = h->ntiles;
        sa.order = nullptr; sa.sc = h->sc; sa.ws = &h->shard_plan->ws; sa.raw = &h->sc->raw; sa.plan = h->shard_plan;
        sa.n = cap; sa.n_cells = h->n; sa.n_global = h->cfg.n_global; sa.gid0 = h->cfg.gid0; sa.seed = h->cfg.seed; sa.epoch = h->epoch;
        sa.K = h->K; sa.logN = h->logN; sa.anc = nullptr; sa.invN = 1.0 / (double)h->cfg.n_global;
        sa.update_lml = 0;                                            // the commit carries the log-ML update
        sa.pack = PackOut{h->rows[h->cur], packed_out, capacity, h->cfg.gid0, h->W, h->push_extra, h->push_pv, h->own_direct ? h->anc : nullptr, (int)me, RingOut{nullptr, 0, 0}};
        if (h->ring_now) {
            // the window exchange: own slots in place (own-direct), every other served slot straight into the window of the rank that holds it
            if (!h->own_direct || !h->ring_active) return fail(h, GPF_ERR_STATE, "window exchange without own-direct resolution / without windows");
            sa.pack.ring = RingOut{h->ring_peers, (int64_t)(h->ring_seq & (RING_PARITIES - 1)) * h->ring_parity_words, h->ring_seq};
        }
        if (su) {                                                     // the GLOBAL tiles: their totals, or (many tiles) their starting points from k_sorted_tiles
            const bool many = (h->cfg.n_global + SP_TILE - 1) / SP_TILE > SP_DIRECT_TILES;
            sa.sp_g = many ? nullptr : h->sp_g; sa.sp_vlo = many ? h->sp_vlo : nullptr;
        }
        s = timed(h, GPF_K_GATHER, [&] {
            // (sorted uniforms: the launch starts at the tile boundary below the first served slot -- up to one tile of slots more)
            launch_search_strat(h, sa, su ? cap + SP_TILE : cap, su);
        });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        return GPF_OK;
    }
    h->push_ticket += 1;
    if ((s = push_args(h, method, tot_all, cr_all, G, me, bounds, a))) return s;
    if (capacity == 0) return GPF_OK;
    h->counts_published = true;
    if (h->ring_now) {
        // the window exchange of the i.i.d. resamplers (GPF_SHARD_EXCHANGE_P2P_ALL): every looked-up row goes straight into the window slot of the rank
        // that holds it; the traffic statistic is kept by the kernel, nobody publishes or waits for counts
        if (!h->own_direct || !h->ring_active) return fail(h, GPF_ERR_STATE, "window exchange without own-direct resolution / without windows");
        a.ring = RingOut{h->ring_peers, (int64_t)(h->ring_seq & (RING_PARITIES - 1)) * h->ring_parity_words, h->ring_seq};
        a.host_counts = nullptr; h->counts_published = false;
    }
    const bool two = method == GPF_RESAMPLE_RESIDUAL;
    if (two && !h->residual_scanned) return fail(h, GPF_ERR_STATE, "residual push needs gpf_shard_residual_scan first");
    const int64_t nt = two ? 2 : 1;
    const size_t lds = search_lds_bytes(h->ntiles, (int)nt);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((capacity + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu));
    const CdfLevels lw_ = levels(h, two ? 2 : 0);
    const CdfLevels lc_ = levels(h, two ? 1 : 0);
    const bool narrow = method == GPF_RESAMPLE_MULTINOMIAL && lw_.off16 != nullptr && lw_.sample == 0;
    const int gridn = (int)std::max<int64_t>(1, std::min<int64_t>((capacity + 4 * SBLOCK - 1) / (4 * SBLOCK), (int64_t)h->n_cu));
    s = timed(h, GPF_K_GATHER, [&] {
        if (narrow)                                  launch_push_multi(h, a, gridn, lw_, capacity, packed_out);
        else if (method == GPF_RESAMPLE_MULTINOMIAL) launch_push<0>(h, a, grid, lds, lw_, lc_, capacity, packed_out);
        else                                         launch_push<1>(h, a, grid, lds, lw_, lc_, capacity, packed_out);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

gpf_status gpf_shard_commit(gpf_handle h, const double* packed, int64_t m, const double* mf_all, const int64_t* tot_all, int32_t G)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!packed || !mf_all || !tot_all || G < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (m != h->n && !(h->own_direct && m >= 0 && m <= h->n)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "a shard must receive exactly one entry per output slot");
    h->pend_ring = false;
    h->pend_own = h->own_direct; h->pend_m = m;                  // (own-direct engine: m entries from the other shards, the rest through h->anc)
    h->pend_own_range = h->own_direct && h->own_direct_range;
    // Deferred like the single-GPU gather: the next gpf_update propagates the entries straight out of
    // the exchange buffer into their slots (k_step<PACKED>); any other consumer scatters first (materialize()).
    // packed / mf_all / tot_all must stay alive and unchanged until then (the caller keeps them until the next commit).
    h->pending_packed = true;
    h->pend_packed = packed; h->pend_mf = mf_all; h->pend_tot = tot_all; h->pend_G = G;
    h->pend_mailbox = h->mb_active && h->mb_engine;
    h->epoch += 1;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->max_valid = false;
    h->residual_scanned = false;
    h->push_counted = false;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_shard_lml_est(gpf_handle h, double* out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if ((s = materialize(h))) return s;                          // a deferred commit also carries the log-ML update
    if ((s = fetch_scalars(h))) return s;
    *out = h->h_sc->lml_est;
    return GPF_OK;
}

} // extern "C"

// =================================================================================== the sharded resample in one call
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
// librccl, once per process: the copy that is already mapped (a host such as PyTorch brings its own) or the ROCm one
bool rccl_load(std::string& err)
{
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* L = nullptr;
    // tests: GPF_RCCL_LIBRARY names a library with the same nine entry points (tests/loopback_rccl: several ranks on one GPU)
    if (const char* over = getenv("GPF_RCCL_LIBRARY")) {
        if (!(L = dlopen(over, RTLD_NOW | RTLD_LOCAL))) { err = std::string("GPF_RCCL_LIBRARY: ") + dlerror(); return false; }
    }
    if (!L) for (const char* n : names) if ((L = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!L) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((L = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!L) { err = std::string("librccl not found: ") + dlerror(); return false; }
#define GPF_RCCL_SYM(field, name) *reinterpret_cast<void**>(&g_rccl.field) = dlsym(L, name); if (!g_rccl.field) { err = std::string("librccl lacks ") + name; return false; }
    GPF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") GPF_RCCL_SYM(CommInitRank, "ncclCommInitRank") GPF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    GPF_RCCL_SYM(AllGather, "ncclAllGather") GPF_RCCL_SYM(Send, "ncclSend") GPF_RCCL_SYM(Recv, "ncclRecv")
    GPF_RCCL_SYM(GroupStart, "ncclGroupStart") GPF_RCCL_SYM(GroupEnd, "ncclGroupEnd") GPF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GPF_RCCL_SYM
    g_rccl.lib = L;
    return true;
}
#define NCCL_TRY(h, expr)                                                                       \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return fail(h, GPF_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// all-gather of `count` elements per rank on the handle's stream; a 1-rank communicator still goes through RCCL when the
// environment asks for it (GPF_SHARD_FORCE_COLLECTIVES=1: exercises the call path on a 1-GPU box), else it is a copy
gpf_status shard_all_gather(gpf_filter* h, const void* src, void* dst, size_t count, ncclDataType_t dt, size_t elem)
{
    static const bool force = getenv("GPF_SHARD_FORCE_COLLECTIVES") && !strcmp(getenv("GPF_SHARD_FORCE_COLLECTIVES"), "1");
    if (h->comm_world == 1 && !(force && h->comm)) {
        if (dst != src) HIP_TRY(h, hipMemcpyAsync(dst, src, count * elem, hipMemcpyDeviceToDevice, h->stream));
        return GPF_OK;
    }
    NCCL_TRY(h, g_rccl.AllGather(src, dst, count, dt, h->comm, h->stream));
    return GPF_OK;
}
// The local summaries and (for the RCCL all-gathers) the gathered arrays are rings of SH_RING rounds: a prioritised resample runs
// three summary rounds (raw weights, priorities, log_ws) and its commit still reads the first.
// Does the (max, flags) mailbox round ride in its consumer's launch (workgroup 0 of the weight scan / of k_sum_shard pushes, every workgroup waits) instead
// of k_pack_mflags' own launch?  1 - 2 us per step faster, but every workgroup of a GPU-filling launch then waits for pushes that only the peers' launches of
// the same kind make: ranks that SHARE a GPU starve each other.  So: yes exactly when every rank of the communicator sits on a device of its own (decided at
// mailbox_setup from the all-gathered PCI bus ids: the same answer on every rank); GPF_SHARD_FUSE_MF=0 / 1 overrides either way.
bool mailbox_fuse_mf(const gpf_filter* h)
{
    const char* e = getenv("GPF_SHARD_FUSE_MF");                  // (read at every call: a host may switch it between two communicators -- bench.py's guarded fall-back does)
    if (e && (!strcmp(e, "0") || !strcmp(e, "1"))) return e[0] == '1';
    return h->mb_fuse_default;
}
constexpr int SH_RING = 4;
gpf_status shard_scratch(gpf_filter* h)
{
    if (h->sh_mf) return GPF_OK;
    const size_t G = (size_t)h->comm_world;
    HIP_TRY(h, hipMalloc(&h->sh_mf, SH_RING * 2 * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->sh_tot, SH_RING * 5 * sizeof(int64_t)));
    HIP_TRY(h, hipMalloc(&h->sh_cr, 2 * sizeof(int64_t)));
    if (h->comm_world == 1 && !h->comm) {                        // one shard, no communicator: the "gathered" arrays ARE the local ones
        h->sh_mf_all = h->sh_mf; h->sh_tot_all = h->sh_tot; h->sh_cr_all = h->sh_cr;
    } else {
        HIP_TRY(h, hipMalloc(&h->sh_mf_all, SH_RING * 2 * G * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->sh_tot_all, SH_RING * 5 * G * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->sh_cr_all, 2 * G * sizeof(int64_t)));
    }
    HIP_TRY(h, hipMemsetAsync(h->sh_tot, 0, SH_RING * 5 * sizeof(int64_t), h->stream));
    return GPF_OK;
}
// phases 1 + 2 of DESIGN.md §6: (max, flags) and {S, sum q^2 limbs} of every shard, gathered on every rank -- through the shard
// mailboxes (peer stores from the producing kernels, waits in the consuming ones: no collective) or two RCCL all-gathers.
// Leaves h->cur_mf_all / cur_tot_all naming the gathered arrays of this round.
struct EngineScope { gpf_filter* h; explicit EngineScope(gpf_filter* f) : h(f) { h->mb_engine = true; } ~EngineScope() { h->mb_engine = false; } };
// is the global summary of the latest one-launch reduction (k_sum_shard) still a description of the current weights and of the mailbox's current rounds?
bool gsum_valid(const gpf_filter* h)
{
    return h->gsum_ok && h->mb_active && h->gsum_mut == h->mutations && h->gsum_mf_seq == h->mb_cur[MB_MF] && h->gsum_tot_seq == h->mb_cur[MB_TOT] &&
           !h->pending_packed && !h->pending_gather && !h->pending_fill && !h->pending_move && !h->sum_pv_set;
}
gpf_status shard_summary(gpf_filter* h, int want_q, bool reuse_mf = false)
{
    gpf_status s = shard_scratch(h);
    if (s) return s;
    const bool mb = h->mb_active;
    if (reuse_mf) {
        // the ESS read in front of this resample has exchanged (max, flags) already (its MB_MF round is the mailbox's current one): only the scan + its {S} round
        h->sh_round++;
        int64_t* tot = h->sh_tot + 5 * (int)((h->sh_round - 1) % SH_RING);
        if ((s = gpf_shard_weight_scan(h, h->cur_mf_all, h->comm_world, want_q, tot))) return s;
        h->cur_tot_all = static_cast<const int64_t*>(mb_gathered(h, MB_TOT));
        return GPF_OK;
    }
    const size_t G = (size_t)h->comm_world;
    const int r = (int)(h->sh_round++ % SH_RING);
    const bool alias = h->sh_mf_all == h->sh_mf;                 // one shard without communicator
    double* mf = h->sh_mf + 2 * r; int64_t* tot = h->sh_tot + 5 * r;
    double* mf_all = alias ? mf : h->sh_mf_all + 2 * G * r; int64_t* tot_all = alias ? tot : h->sh_tot_all + 5 * G * r;
    // One shard without a communicator: the first summary -- (max, flags) -- is folded inside the scan's launch (no k_pack_mflags launch).
    // With the mailboxes the same fusion is possible (GPF_SHARD_FUSE_MF=1: the scan's workgroup 0 pushes, every workgroup waits).  Round 4 measured
    // it SLOWER on one rank (+6 us per step) -- that was the push's system-scope RELEASE store, an L2 write-back on the critical path of every scan
    // workgroup; with sealed entries (gpf_k_common.hpp mbox_seal) it is 1 - 2 us FASTER (58.6 / 47.5 / 52.2 -> 57.8 / 45.9 / 50.5 us per step,
    // multinomial / stratified / sorted).  Still opt-in: every workgroup of the launch waits for pushes that only the peers' launches of the same
    // kind make, so ranks that SHARE a GPU can starve each other (shard_global_summary_launch); for ranks with a GPU each.  As an RCCL all-gather
    // the summary needs its own launch ahead of the collective.
    const bool fuse_mb = mailbox_fuse_mf(h);
    const bool fuse = alias || (mb && fuse_mb);
    if (fuse) { if ((s = shard_ready(h)) || (s = shard_max_slots(h))) return s; }
    else if ((s = gpf_shard_weight_max(h, mf))) return s;
    if (!mb && !alias && (s = shard_all_gather(h, mf, mf_all, 2, ncclDouble, sizeof(double)))) return s;
    struct FuseScope { gpf_filter* h; ~FuseScope() { h->fuse_mf_out = nullptr; } } fuse_scope{h};
    h->fuse_mf_out = fuse ? mf : nullptr;
    // (fused: MB_MF's round begins inside gpf_shard_weight_scan -- name the gathered array after it)
    if (!fuse) h->cur_mf_all = mb ? static_cast<const double*>(mb_gathered(h, MB_MF)) : mf_all;
    if ((s = gpf_shard_weight_scan(h, fuse ? mf_all : h->cur_mf_all, h->comm_world, want_q, tot))) return s;      // (fused + mailboxes: the callee names the gathered array)
    if (fuse) h->cur_mf_all = mb ? static_cast<const double*>(mb_gathered(h, MB_MF)) : mf_all;
    if (!mb && !alias && (s = shard_all_gather(h, tot, tot_all, 5, ncclInt64, sizeof(int64_t)))) return s;
    h->cur_tot_all = mb ? static_cast<const int64_t*>(mb_gathered(h, MB_TOT)) : tot_all;
    return GPF_OK;
}
// the same over another view of the weights (a prioritised resample: alpha lw, then log_ws), into summary slot `slot`, with or
// without the CDF levels
gpf_status shard_summary_of(gpf_filter* h, const PrioView& pv, WSum* slot, bool want_cdf)
{
    struct Scope { gpf_filter* h; ~Scope() { h->sum_pv_set = false; h->sum_slot = nullptr; h->sum_no_cdf = false; } } scope{h};
    h->sum_pv = pv; h->sum_pv_set = true; h->sum_slot = slot; h->sum_no_cdf = !want_cdf;
    return shard_summary(h, 0);
}
// the gathered summaries on the host: global max, flags, S and sum q^2
gpf_status shard_scalars(gpf_filter* h, double& m, int& flags, uint64_t& S, uint64_t& Qhi, uint64_t& Qlo)
{
    const int G = h->comm_world;
    std::vector<double> mf(2 * (size_t)G); std::vector<int64_t> tot(5 * (size_t)G);
    if (h->mb_active) {                                          // wait for the peers' entries, then out of the mailbox into plain device memory
        GPF_LAUNCH(k_mbox_collect, dim3(1), dim3(BLOCK), 0, h->stream, mb_wait(h, MB_MF), reinterpret_cast<const uint64_t*>(h->cur_mf_all),
                   reinterpret_cast<uint64_t*>(h->sh_mf_all), 2 * G, mb_wait(h, MB_TOT), reinterpret_cast<const uint64_t*>(h->cur_tot_all),
                   reinterpret_cast<uint64_t*>(h->sh_tot_all), 5 * G);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipMemcpyAsync(mf.data(), h->mb_active ? h->sh_mf_all : h->cur_mf_all, mf.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(tot.data(), h->mb_active ? h->sh_tot_all : h->cur_tot_all, tot.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    m = -HUGE_VAL; flags = 0; S = 0;
    unsigned __int128 Q = 0;
    for (int g = 0; g < G; ++g) {
        m = std::max(m, mf[2 * g]); flags |= (int)mf[2 * g + 1]; S += (uint64_t)tot[5 * g];
        for (int k = 0; k < 4; ++k) Q += (unsigned __int128)(uint64_t)tot[5 * g + 1 + k] << (32 * k);
    }
    if (!(flags & FLAG_NAN) && m == -HUGE_VAL) flags |= FLAG_ALL_NEGINF;
    Qhi = (uint64_t)(Q >> 64); Qlo = (uint64_t)Q;
    return check_scan_timeout(h);
}
// ---- shard mailboxes: allocation, hipIpc exchange of the handles over the communicator that was just created, peer mapping.
// Every decision is taken from data all ranks hold identically (the all-gathered packets), so either every rank ends with the
// mailboxes up or every rank stays on the RCCL all-gathers.  Any failure is soft: the collectives remain.
void ring_teardown(gpf_filter* h)
{
    for (void* p : h->ring_opened) (void)hipIpcCloseMemHandle(p);
    h->ring_opened.clear();
    if (h->ring_peers) (void)hipFree(h->ring_peers);
    if (h->ring) (void)hipFree(h->ring);
    if (h->tr_dev) (void)hipFree(h->tr_dev);
    h->ring_peers = nullptr; h->ring = nullptr; h->tr_dev = nullptr; h->ring_active = false; h->pend_ring = false;
}
void mailbox_teardown(gpf_filter* h)
{
    ring_teardown(h);
    for (void* p : h->mb_opened) (void)hipIpcCloseMemHandle(p);
    h->mb_opened.clear();
    if (h->mb_peers) (void)hipFree(h->mb_peers);
    if (h->mbox) (void)hipFree(h->mbox);
    h->mb_peers = nullptr; h->mbox = nullptr; h->mb_active = false;
}
struct MboxPacket { hipIpcMemHandle_t handle; int64_t ok; int64_t pid; char bus[32]; };
// GPF_SHARD_SELFTEST: 0 = no self-test of mailboxes / windows at gpf_comm_create; fail_mailbox / fail_windows = rank 0 reports a failed test (tests of the fall-back)
int selftest_mode() { const char* e = getenv("GPF_SHARD_SELFTEST"); return !e ? 1 : (!strcmp(e, "0") ? 0 : (!strcmp(e, "fail_mailbox") ? 2 : (!strcmp(e, "fail_windows") ? 3 : 1))); }
// all-gather of `each` bytes per rank, host to host, over the handle's communicator (setup only)
gpf_status host_all_gather(gpf_filter* h, const void* src, void* dst_host, size_t each)
{
    const int G = h->comm_world;
    if (G == 1) { memcpy(dst_host, src, each); return GPF_OK; }
    char *dsrc = nullptr, *ddst = nullptr;
    struct Bufs { char*& a; char*& b; ~Bufs() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } bufs{dsrc, ddst};
    HIP_TRY(h, hipMalloc(&dsrc, each)); HIP_TRY(h, hipMalloc(&ddst, each * G));
    HIP_TRY(h, hipMemcpyAsync(dsrc, src, each, hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(h, g_rccl.AllGather(dsrc, ddst, each, ncclInt8, h->comm, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dst_host, ddst, each * G, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GPF_OK;
}
// ---- slot-addressed receive windows (gpf_k_common.hpp): allocated, exported and mapped exactly like the mailboxes, behind them (a rank without
// mailboxes has no windows); every decision again from all-gathered data, any failure soft -- then every rank keeps the grouped ncclSend / ncclRecv.
gpf_status ring_setup(gpf_filter* h)
{
    const char* mode = getenv("GPF_SHARD_EXCHANGE");              // "rccl": no windows at all (A/B measurements, fallback drills)
    if (!h->mb_active || (mode && !strcmp(mode, "rccl"))) return GPF_OK;
    if (mode && strcmp(mode, "p2p") && strcmp(mode, "p2p_all")) return fail(h, GPF_ERR_INVALID_ARGUMENT, "GPF_SHARD_EXCHANGE: p2p, p2p_all or rccl");
    const int G = h->comm_world, me = h->comm_rank;
    // one entry per local slot of the LARGEST shard (the first n_global % G shards hold one particle more), two parities
    const int64_t n_max = (h->cfg.n_global + G - 1) / G;
    h->ring_parity_words = n_max * (h->W + 2);
    const size_t bytes = (size_t)RING_PARITIES * (size_t)h->ring_parity_words * sizeof(uint64_t);
    int64_t ok = 1;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&h->ring), bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&h->ring, bytes) != hipSuccess) { (void)hipGetLastError(); h->ring = nullptr; ok = 0; }
    }
    if (ok && hipMalloc(&h->tr_dev, 2 * sizeof(int64_t)) != hipSuccess) { (void)hipGetLastError(); h->tr_dev = nullptr; ok = 0; }
    // (a failure of THIS rank alone must not keep it out of the all-gather its peers are about to enter: it only votes "no")
    if (ok && (hipMemsetAsync(h->ring, 0, bytes, h->stream) != hipSuccess || hipMemsetAsync(h->tr_dev, 0, 2 * sizeof(int64_t), h->stream) != hipSuccess ||
               hipStreamSynchronize(h->stream) != hipSuccess)) { (void)hipGetLastError(); ok = 0; }
    std::vector<MboxPacket> all((size_t)G);
    MboxPacket mine{};
    mine.pid = (int64_t)getpid();
    if (G > 1 && ok && hipIpcGetMemHandle(&mine.handle, h->ring) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
    mine.ok = ok;
    gpf_status s = host_all_gather(h, &mine, all.data(), sizeof(MboxPacket));
    if (s) { ring_teardown(h); return s; }
    for (int r = 0; r < G; ++r) if (!all[r].ok) { ring_teardown(h); return GPF_OK; }
    std::vector<uint64_t*> peers((size_t)G, nullptr);
    int64_t opened = 1;
    for (int r = 0; r < G && opened; ++r) {
        if (r == me) { peers[r] = h->ring; continue; }
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, all[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened = 0; break; }
        h->ring_opened.push_back(ptr);
        peers[r] = static_cast<uint64_t*>(ptr);
    }
    std::vector<int64_t> oks((size_t)G, 0);
    if ((s = host_all_gather(h, &opened, oks.data(), sizeof(int64_t)))) { ring_teardown(h); return s; }
    for (int r = 0; r < G; ++r) if (!oks[r]) { ring_teardown(h); return GPF_OK; }
    HIP_TRY(h, hipMalloc(&h->ring_peers, (size_t)G * sizeof(uint64_t*)));
    HIP_TRY(h, hipMemcpy(h->ring_peers, peers.data(), (size_t)G * sizeof(uint64_t*), hipMemcpyHostToDevice));
    h->ring_seq = 0;
    // Self-test, as for the mailboxes: in round d every rank stores one entry into the window of rank (me + d) % G and waits for the one rank (me - d) % G
    // stores into its own (k_ring_selftest: the stores, loads and seals of a slab exchange across the real mappings); the all-gather of the verdicts between
    // the rounds is their barrier.  One "no" and every rank keeps the grouped ncclSend / ncclRecv.
    if (G > 1 && selftest_mode() != 0) {
        int32_t* bad = nullptr;
        int64_t pass = hipMalloc(&bad, sizeof(int32_t)) == hipSuccess && hipMemsetAsync(bad, 0, sizeof(int32_t), h->stream) == hipSuccess ? 1 : 0;
        if (!pass) (void)hipGetLastError();
        bool all_pass = true;
        for (int d = 1; d < G; ++d) {
            h->ring_seq += 1;
            if (pass) {
                GPF_LAUNCH(k_ring_selftest, dim3(1), dim3(WAVE), 0, h->stream, h->ring_peers, h->ring, G, me, d, h->ring_parity_words, h->ring_seq, h->W, h->h_timeout, bad);
                int32_t hb = 0;
                if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&hb, bad, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                    hipStreamSynchronize(h->stream) != hipSuccess || hb != 0) { (void)hipGetLastError(); pass = 0; }
                if (__atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) != 0) { __atomic_store_n(h->h_timeout, 0, __ATOMIC_RELEASE); pass = 0; }
            }
            if (selftest_mode() == 3 && me == 0) pass = 0;         // (tests: "this rank's test failed")
            if ((s = host_all_gather(h, &pass, oks.data(), sizeof(int64_t)))) { if (bad) (void)hipFree(bad); ring_teardown(h); return s; }
            for (int r = 0; r < G; ++r) all_pass = all_pass && oks[r] != 0;
            if (!all_pass) break;                                 // (every rank sees the same verdicts: every rank stops here)
        }
        if (bad) (void)hipFree(bad);
        if (!all_pass) { ring_teardown(h); return GPF_OK; }
    }
    h->ring_active = true;
    h->exchange_mode = (mode && !strcmp(mode, "p2p_all")) ? GPF_SHARD_EXCHANGE_P2P_ALL : GPF_SHARD_EXCHANGE_P2P;
    return GPF_OK;
}
gpf_status mailbox_setup(gpf_filter* h)
{
    const char* mode = getenv("GPF_SHARD_SUMMARY");               // "rccl": keep the all-gathers (A/B measurements, fallback drills)
    if (mode && !strcmp(mode, "rccl")) return GPF_OK;
    const int G = h->comm_world, me = h->comm_rank;
    if (!h->comm) return GPF_OK;                                  // a single shard without communicator aliases its own summaries
    const size_t bytes = (size_t)MB_TOTAL_WORDS * sizeof(uint64_t);
    int64_t ok = 1;
    // uncached device memory (peers write it, system-scope loads read it); plain device memory serves as well
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&h->mbox), bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&h->mbox, bytes) != hipSuccess) { (void)hipGetLastError(); h->mbox = nullptr; ok = 0; }
    }
    if (h->mbox) { HIP_TRY(h, hipMemsetAsync(h->mbox, 0, bytes, h->stream)); HIP_TRY(h, hipStreamSynchronize(h->stream)); }
    std::vector<MboxPacket> all((size_t)G);
    MboxPacket mine{};
    mine.pid = (int64_t)getpid();
    if (hipDeviceGetPCIBusId(mine.bus, (int)sizeof(mine.bus), h->cfg.device) != hipSuccess) { (void)hipGetLastError(); mine.bus[0] = 0; }
    if (G > 1 && ok) {
        if (hipIpcGetMemHandle(&mine.handle, h->mbox) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
    }
    mine.ok = ok;
    auto gather = [&](const void* src, void* dst_host, size_t each) -> gpf_status { return host_all_gather(h, src, dst_host, each); };
    gpf_status s = gather(&mine, all.data(), sizeof(MboxPacket));
    if (s) { mailbox_teardown(h); return s; }                     // (a failed collective is not soft: the communicator is unusable)
    bool all_ok = true;
    for (int r = 0; r < G; ++r) {
        all_ok = all_ok && all[r].ok != 0;
        if (r != me && all[r].pid == mine.pid) all_ok = false;    // two shards in one process: hipIpc cannot map a handle of its own process
    }
    if (!all_ok) { mailbox_teardown(h); return GPF_OK; }
    std::vector<uint64_t*> peers((size_t)G, nullptr);
    int64_t opened = 1;
    for (int r = 0; r < G && opened; ++r) {
        if (r == me) { peers[r] = h->mbox; continue; }
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, all[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened = 0; break; }
        h->mb_opened.push_back(ptr);
        peers[r] = static_cast<uint64_t*>(ptr);
    }
    std::vector<int64_t> oks((size_t)G, 0);
    if ((s = gather(&opened, oks.data(), sizeof(int64_t)))) { mailbox_teardown(h); return s; }
    for (int r = 0; r < G; ++r) if (!oks[r]) { mailbox_teardown(h); return GPF_OK; }
    HIP_TRY(h, hipMalloc(&h->mb_peers, (size_t)G * sizeof(uint64_t*)));
    HIP_TRY(h, hipMemcpy(h->mb_peers, peers.data(), (size_t)G * sizeof(uint64_t*), hipMemcpyHostToDevice));
    for (int k = 0; k < MB_KINDS; ++k) h->mb_seq[k] = h->mb_cur[k] = 0;
    h->mb_active = true;
    // a device of its own for every rank?  (an unknown bus id counts as shared; ranks of other nodes cannot be told apart by bus id alone and may
    // read as "shared": the separate launch is always safe)
    bool own_device = true;
    for (int r = 0; r < G && own_device; ++r) {
        all[r].bus[sizeof(all[r].bus) - 1] = 0;
        if (!all[r].bus[0]) own_device = false;
        for (int q = 0; q < r && own_device; ++q) if (!strcmp(all[r].bus, all[q].bus)) own_device = false;
    }
    h->mb_fuse_default = own_device;
    // Self-test before anything relies on the mailboxes (they are the default transport of the small summaries, and what this code could be tried on while it
    // was written was several ranks on ONE device): a few dependent rounds of the calibration kernel -- every rank stores into every peer's mailbox and waits
    // for all G entries.  A rank whose wait gives up votes "no"; one "no" and every rank keeps the RCCL all-gathers.  GPF_SHARD_SELFTEST=0 skips it.
    if (G > 1 && selftest_mode() != 0) {
        int64_t pass = 1;
        GPF_LAUNCH(k_mbox_rounds, dim3(1), dim3(WAVE), 0, h->stream, h->mb_peers, h->mbox, G, me, h->mb_seq[MB_CAL], 4, h->h_timeout);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { (void)hipGetLastError(); pass = 0; }
        if (__atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) != 0) { __atomic_store_n(h->h_timeout, 0, __ATOMIC_RELEASE); pass = 0; }
        if (selftest_mode() == 2 && me == 0) pass = 0;            // (tests: "this rank's test failed")
        h->mb_seq[MB_CAL] += 4;
        if ((s = gather(&pass, oks.data(), sizeof(int64_t)))) { mailbox_teardown(h); return s; }
        for (int r = 0; r < G; ++r) if (!oks[r]) { mailbox_teardown(h); return GPF_OK; }
    }
    return GPF_OK;
}
} // namespace

extern "C" {

/* 1: the handle's sharded resamples exchange their summaries through the shard mailboxes (peer stores, no collective);
 * 0: through RCCL all-gathers (or there is nothing to exchange: one shard) */
gpf_status gpf_comm_summary_mode(gpf_handle h, int32_t* mailbox)
{
    if (!h || !mailbox) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *mailbox = h->mb_active ? 1 : 0;
    return GPF_OK;
}

/* exchange volume of this handle's gpf_shard_resample calls so far: out4 = {calls, entries sent to other ranks, entries received from other
 * ranks, bytes of one exchanged entry in the latest call}; reset = 1 clears the counters after reading */
gpf_status gpf_comm_traffic(gpf_handle h, int64_t* out4, int32_t reset)
{
    if (!h || !out4) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    if (h->tr_dev) {                                              // the window exchanges' counts never reach the host: their plan kernels kept them on the device
        HIP_TRY(h, hipSetDevice(h->cfg.device));
        int64_t dv[2] = {0, 0};
        HIP_TRY(h, hipMemcpyAsync(dv, h->tr_dev, sizeof(dv), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->tr_dev, 0, sizeof(dv), h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->tr_sent += dv[0]; h->tr_recv += dv[1];
    }
    out4[0] = h->tr_calls; out4[1] = h->tr_sent; out4[2] = h->tr_recv; out4[3] = h->tr_entry_bytes;
    if (reset) h->tr_calls = h->tr_sent = h->tr_recv = 0;
    return GPF_OK;
}

/* what the transports under a sharded resample cost on THIS machine (a scaling run prints it beside its step times):
 *   out4[0] us per grouped ncclSend / ncclRecv exchange of `entries` packed entries ((W + 1) doubles each) with EVERY peer, mean of `reps`
 *   out4[1] the per-link rate of that exchange in GB/s: bytes one rank put on ONE link / out4[0]
 *   out4[2] us per mailbox round: every rank stores its entry into every peer's mailbox and waits for all of theirs (`reps` dependent rounds in one launch)
 *   out4[3] us of the launch that ran the mailbox rounds with reps = 0 (the floor under out4[2] x reps: launch + event overhead)
 * Collective: every rank calls it with the same arguments.  0 where there is nothing to measure (one rank; no mailboxes). */
gpf_status gpf_comm_calibrate(gpf_handle h, int64_t entries, int32_t reps, double* out4)
{
    if (!h || !out4) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_calibrate needs gpf_comm_create first");
    if (entries < 1 || reps < 1 || reps > 10000) return fail(h, GPF_ERR_INVALID_ARGUMENT, "entries >= 1, 1 <= reps <= 10000");
    gpf_status s = shard_ready(h);
    if (s) return s;
    for (int k = 0; k < 4; ++k) out4[k] = 0.0;
    const int G = h->comm_world, me = h->comm_rank;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(h, hipEventCreate(&e0)); HIP_TRY(h, hipEventCreate(&e1));
    struct Ev { hipEvent_t a, b; ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evs{e0, e1};
    if (G > 1 && h->comm) {
        const int64_t E = h->W + 1;
        double *sb = nullptr, *rb = nullptr;
        const size_t bytes = (size_t)entries * E * G * sizeof(double);
        // (allocation failures are local: decide together before the first collective)
        int64_t ok = hipMalloc(&sb, bytes) == hipSuccess && hipMalloc(&rb, bytes) == hipSuccess ? 1 : 0;
        if (!ok) (void)hipGetLastError();
        std::vector<int64_t> oks((size_t)G, 0);
        if ((s = host_all_gather(h, &ok, oks.data(), sizeof(int64_t)))) { if (sb) (void)hipFree(sb); if (rb) (void)hipFree(rb); return s; }
        bool all_ok = true;
        for (int r = 0; r < G; ++r) all_ok = all_ok && oks[r] != 0;
        if (all_ok) {
            HIP_TRY(h, hipMemsetAsync(sb, 0, bytes, h->stream));
            ncclResult_t first = ncclSuccess;
            auto note = [&](ncclResult_t r) { if (r != ncclSuccess && first == ncclSuccess) first = r; };
            for (int it = 0; it < reps + 2; ++it) {               // (two untimed exchanges first: connection set-up)
                if (it == 2) HIP_TRY(h, hipEventRecord(e0, h->stream));
                note(g_rccl.GroupStart());
                for (int g = 0; g < G; ++g) {
                    if (g == me) continue;
                    note(g_rccl.Send(sb + (size_t)g * entries * E, (size_t)(entries * E), ncclDouble, g, h->comm, h->stream));
                    note(g_rccl.Recv(rb + (size_t)g * entries * E, (size_t)(entries * E), ncclDouble, g, h->comm, h->stream));
                }
                note(g_rccl.GroupEnd());
            }
            HIP_TRY(h, hipEventRecord(e1, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (first != ncclSuccess) { (void)hipFree(sb); (void)hipFree(rb); return fail(h, GPF_ERR_HIP, std::string("calibration exchange: ") + g_rccl.GetErrorString(first)); }
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
            out4[0] = 1e3 * (double)ms / reps;
            out4[1] = out4[0] > 0.0 ? (double)(entries * E * (int64_t)sizeof(double)) / (out4[0] * 1e-6) / 1e9 : 0.0;
        }
        if (sb) (void)hipFree(sb);
        if (rb) (void)hipFree(rb);
    }
    if (h->mb_active && h->h_timeout) {
        for (int pass = 0; pass < 2; ++pass) {
            const int n = pass == 0 ? reps : 0;
            HIP_TRY(h, hipEventRecord(e0, h->stream));
            GPF_LAUNCH(k_mbox_rounds, dim3(1), dim3(WAVE), 0, h->stream, h->mb_peers, h->mbox, G, me, h->mb_seq[MB_CAL], n, h->h_timeout);
            HIP_TRY(h, hipEventRecord(e1, h->stream));
            HIP_TRY(h, hipGetLastError());
            h->mb_seq[MB_CAL] += (uint64_t)n;
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
            if (pass == 0) out4[2] = 1e3 * (double)ms; else out4[3] = 1e3 * (double)ms;
        }
        out4[2] = (out4[2] - out4[3] > 0.0 ? out4[2] - out4[3] : 0.0) / reps;
        if ((s = check_scan_timeout(h))) return s;
    }
    return GPF_OK;
}

gpf_status gpf_comm_set_exchange(gpf_handle h, int32_t mode)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (mode != GPF_SHARD_EXCHANGE_RCCL && mode != GPF_SHARD_EXCHANGE_P2P && mode != GPF_SHARD_EXCHANGE_P2P_ALL)
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "exchange mode: GPF_SHARD_EXCHANGE_RCCL, GPF_SHARD_EXCHANGE_P2P or GPF_SHARD_EXCHANGE_P2P_ALL");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_set_exchange needs gpf_comm_create first");
    if (mode != GPF_SHARD_EXCHANGE_RCCL && !h->ring_active) return fail(h, GPF_ERR_STATE, "no receive windows on this communicator (hipIpc mapping not possible, or GPF_SHARD_EXCHANGE / GPF_SHARD_SUMMARY = rccl)");
    h->exchange_mode = mode;
    return GPF_OK;
}
gpf_status gpf_comm_exchange(gpf_handle h, int32_t* mode)
{
    if (!h || !mode) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_exchange needs gpf_comm_create first");
    *mode = h->exchange_mode;
    return GPF_OK;
}

gpf_status gpf_comm_set_plan(gpf_handle h, int32_t plan)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (plan != GPF_SHARD_PLAN_PUSH && plan != GPF_SHARD_PLAN_PULL) return fail(h, GPF_ERR_INVALID_ARGUMENT, "exchange plan: GPF_SHARD_PLAN_PUSH or GPF_SHARD_PLAN_PULL");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_set_plan needs gpf_comm_create first");
    h->shard_plan_kind = plan;
    return GPF_OK;
}
gpf_status gpf_comm_plan(gpf_handle h, int32_t* plan)
{
    if (!h || !plan) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_plan needs gpf_comm_create first");
    *plan = h->shard_plan_kind;
    return GPF_OK;
}

gpf_status gpf_comm_unique_id(void* id128)
{
    if (!id128) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null id");
    std::string err;
    if (!rccl_load(err)) return fail(nullptr, GPF_ERR_HIP, err);
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return fail(nullptr, GPF_ERR_HIP, "ncclGetUniqueId failed");
    memcpy(id128, &id, sizeof(id));
    return GPF_OK;
}

gpf_status gpf_comm_create(gpf_handle h, const void* id128, int32_t rank, int32_t world)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (world < 1 || world > MAX_SHARDS || rank < 0 || rank >= world) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad rank / world (<= 64 shards)");
    if (h->parent) return fail(h, GPF_ERR_STATE, "a sub-state view has no communicator");
    if (h->comm || h->sh_mf) return fail(h, GPF_ERR_STATE, "the handle already has a communicator");
    if (world > 1 && !id128) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null id");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->comm_rank = rank; h->comm_world = world;
    if (id128) {                                                 // (world == 1 without an id: no RCCL at all)
        std::string err;
        if (!rccl_load(err)) return fail(h, GPF_ERR_HIP, err);
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        NCCL_TRY(h, g_rccl.CommInitRank(&h->comm, world, id, rank));
    }
    gpf_status s = shard_scratch(h);
    if (s) return s;
    h->shard_plan_kind = GPF_SHARD_PLAN_PUSH;
    if (const char* e = getenv("GPF_SHARD_PLAN")) {
        if (!strcmp(e, "pull")) h->shard_plan_kind = GPF_SHARD_PLAN_PULL;
        else if (strcmp(e, "push")) return fail(h, GPF_ERR_INVALID_ARGUMENT, "GPF_SHARD_PLAN: push or pull");
    }
    h->exchange_mode = GPF_SHARD_EXCHANGE_RCCL;
    if ((s = mailbox_setup(h))) return s;
    return ring_setup(h);
}

gpf_status gpf_comm_destroy(gpf_handle h)
{
    if (!h) return GPF_OK;
    hipSetDevice(h->cfg.device);
    // a deferred commit (gpf_shard_resample leaves the new population in the exchange buffers: pend_packed / pend_mf / pend_tot
    // point into sh_recv / sh_send / sh_mf_all / sh_tot_all) is scattered into the filter's own rows BEFORE those buffers go
    gpf_status ms = GPF_OK;
    if (h->pending_packed && h->initialized && h->rows[0]) ms = materialize(h);
    h->pending_packed = false; h->pend_packed = nullptr; h->pend_mf = nullptr; h->pend_tot = nullptr; h->pend_G = 0;
    if (h->stream) hipStreamSynchronize(h->stream);
    mailbox_teardown(h);
    h->comm_poisoned = false;
    h->cur_mf_all = nullptr; h->cur_tot_all = h->cur_cr_all = nullptr;
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0;
    void* bufs[] = {h->sh_mf, h->sh_mf_all != h->sh_mf ? h->sh_mf_all : nullptr, h->sh_tot, h->sh_tot_all != h->sh_tot ? h->sh_tot_all : nullptr,
                    h->sh_cr, h->sh_cr_all != h->sh_cr ? h->sh_cr_all : nullptr, h->sh_send, h->sh_recv};
    for (void* b : bufs) if (b) (void)hipFree(b);
    h->sh_mf = h->sh_mf_all = nullptr; h->sh_tot = h->sh_tot_all = h->sh_cr = h->sh_cr_all = nullptr;
    h->sh_send = h->sh_recv = nullptr; h->sh_send_cap = h->sh_recv_cap = 0;
    return ms;
}

// Phase 3 of the PULL plan (gpf_k_shard.hpp, k_pull_scan): this shard's requests grouped by owner, the request matrix gathered on
// every rank (one all-gather of G counts + the host wait the split sizes need), the requests exchanged straight into the owners'
// staging lists, the exchange counters set from the matrix.  Leaves the handle where gpf_shard_push_count leaves it; counts[0..G) =
// entries to serve per shard, counts[G..2G) = entries to receive.  Buffers are allocated by pull_buffers before any collective.
static gpf_status pull_buffers(gpf_filter* h, int G)
{
    if (!h->pull_counts) {
        HIP_TRY(h, hipMalloc(&h->pull_counts, (size_t)MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->pull_pc, (size_t)MAX_SHARDS * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->pull_pc_all, (size_t)MAX_SHARDS * MAX_SHARDS * sizeof(int64_t)));
        HIP_TRY(h, hipHostMalloc(&h->h_pull_pc_all, (size_t)MAX_SHARDS * MAX_SHARDS * sizeof(int64_t)));
    }
    const int64_t want = (int64_t)G * std::max<int64_t>(h->n, 1);
    if (h->pull_req_cap < want) {
        if (h->pull_req) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->pull_req); h->pull_req = nullptr; h->pull_req_cap = 0; }
        HIP_TRY(h, hipMalloc(&h->pull_req, (size_t)want * sizeof(ulonglong2)));
        h->pull_req_cap = want;
    }
    return GPF_OK;
}
static gpf_status pull_requests(gpf_filter* h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int G, int me, const int64_t* bounds,
                                bool exchange, bool force_self, std::vector<int64_t>& counts)
{
    PushArgs a;
    gpf_status s = push_args(h, method, tot_all, cr_all, G, me, bounds, a);
    if (s) return s;
    const int64_t n = h->n;
    HIP_TRY(h, hipMemsetAsync(h->pull_counts, 0, (size_t)MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t), h->stream));
    const int64_t nch = (n + PUSH_CHUNK - 1) / PUSH_CHUNK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nch, (int64_t)h->n_cu * PUSH_SCAN_BLOCKS_PER_CU));
    s = timed(h, GPF_K_SEARCH, [&] {
        if (method == GPF_RESAMPLE_MULTINOMIAL) GPF_LAUNCH((k_pull_scan<0>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a, h->pull_req, n, h->pull_counts);
        else                                    GPF_LAUNCH((k_pull_scan<1>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a, h->pull_req, n, h->pull_counts);
    });
    if (s) return s;
    GPF_LAUNCH(k_pull_counts, dim3(1), dim3(WAVE), 0, h->stream, h->pull_counts, G, h->pull_pc);
    HIP_TRY(h, hipGetLastError());
    if ((s = shard_all_gather(h, h->pull_pc, h->pull_pc_all, (size_t)G, ncclInt64, sizeof(int64_t)))) return s;
    HIP_TRY(h, hipMemcpyAsync(h->h_pull_pc_all, h->pull_pc_all, (size_t)G * G * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    {
        const auto w0 = std::chrono::steady_clock::now();
        HIP_TRY(h, hipStreamSynchronize(h->stream));              // the host wait of this plan: the split sizes of BOTH exchanges
        if (h->phases.on) h->phases.host_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    }
    const int64_t* M = h->h_pull_pc_all;                          // M[requester][owner]
    int64_t asked = 0;
    for (int g = 0; g < G; ++g) { counts[g] = M[(size_t)g * G + me]; counts[G + g] = M[(size_t)me * G + g]; asked += counts[G + g]; }
    // (as in the push plan: from the first exchange on, a local failure is remembered and the rank still joins the exchanges)
    gpf_status late = GPF_OK; std::string late_msg;
    auto remember = [&](gpf_status st) { if (st && !late) { late = st; late_msg = h->err; } };
    if (asked != n) remember(fail(h, GPF_ERR_STATE, "request counts do not add up to the shard's slots"));
    ncclResult_t first = ncclSuccess; const char* where = "";
    auto note = [&](ncclResult_t r, const char* w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; where = w; } };
    if (exchange) note(g_rccl.GroupStart(), "ncclGroupStart");
    for (int g = 0; g < G; ++g) {
        ulonglong2* from = h->pull_req + (size_t)g * n;
        ulonglong2* to = h->push_stage + bounds[g];
        if (g == me && !force_self) continue;
        if (counts[G + g]) note(g_rccl.Send(from, (size_t)counts[G + g] * 2, ncclUint64, g, h->comm, h->stream), "ncclSend");
        if (counts[g] && counts[g] <= bounds[g + 1] - bounds[g]) note(g_rccl.Recv(to, (size_t)counts[g] * 2, ncclUint64, g, h->comm, h->stream), "ncclRecv");
    }
    if (exchange) note(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess) remember(fail(h, GPF_ERR_HIP, std::string(where) + ": " + g_rccl.GetErrorString(first)));
    if (!force_self && counts[me]) {                              // the shard's own requests never touch RCCL
        const hipError_t ce = hipMemcpyAsync(h->push_stage + bounds[me], h->pull_req + (size_t)me * n, (size_t)counts[me] * sizeof(ulonglong2), hipMemcpyDeviceToDevice, h->stream);
        if (ce != hipSuccess) remember(fail(h, GPF_ERR_HIP, std::string("self copy: ") + hipGetErrorString(ce)));
    }
    GPF_LAUNCH(k_pull_set_counts, dim3(1), dim3(WAVE), 0, h->stream, h->pull_pc_all, G, me, h->shard_counts);
    if (hipGetLastError() != hipSuccess) remember(fail(h, GPF_ERR_HIP, "k_pull_set_counts launch"));
    if (late) { h->err = late_msg; return late; }
    h->counts_published = false;
    h->push_counted = true;
    return GPF_OK;
}

// ---- stratified with sort_particles = true across shards: the replicated plan (gpf_k_shard.hpp AncPlan)
static AncPlan anc_plan(const gpf_filter* h, int G, int me)
{
    AncPlan p{};
    p.anc_g = h->planner ? h->planner->anc : nullptr;
    p.n_global = h->cfg.n_global; p.G = G; p.me = me;
    p.base = p.n_global / G; p.extra = p.n_global % G;
    p.lo = h->cfg.gid0; p.hi = h->cfg.gid0 + h->n;
    return p;
}
// everything of the plan that can fail for reasons of this rank alone: before the first collective of the call
static gpf_status shard_sorted_buffers(gpf_filter* h, int G)
{
    gpf_status s;
    if (!h->planner) {
        gpf_config c = h->cfg;
        c.n_particles = c.n_global; c.gid0 = 0; c.keep_prev = 0; c.stream = (void*)h->stream; c.params = h->args.P;
        gpf_handle p = nullptr;
        if ((s = gpf_create(&c, &p))) return fail(h, s, std::string("planner of the sorted sharded resample (n_global particles on every rank): ") + gpf_last_error(nullptr));
        h->planner = p;
        if (p->chain_counted && p->cfg.device < 16) {            // the planner runs on its owner's stream, never beside it: not one more filter for the gate of
            std::lock_guard<std::mutex> lk(g_chain[p->cfg.device].mu);   // chained kernels (gpf_host.hpp ChainGate)
            g_chain[p->cfg.device].live -= 1; p->chain_counted = false;
        }
    }
    if ((s = ensure_shard_counts(h))) return s;
    if (!h->anc_cursors) HIP_TRY(h, hipMalloc(&h->anc_cursors, MAX_SHARDS * sizeof(unsigned long long)));
    const int64_t N = h->cfg.n_global, per = N / G + (N % G ? 1 : 0);
    if (N % G && h->sorted_per != per) {
        for (void* q : {(void*)h->sorted_src, (void*)h->sorted_gath}) if (q) (void)hipFree(q);
        h->sorted_src = h->sorted_gath = nullptr; h->sorted_per = 0;
        HIP_TRY(h, hipMalloc(&h->sorted_src, (size_t)per * sizeof(double)));
        HIP_TRY(h, hipMemsetAsync(h->sorted_src, 0, (size_t)per * sizeof(double), h->stream));
        HIP_TRY(h, hipMalloc(&h->sorted_gath, (size_t)per * G * sizeof(double)));
        h->sorted_per = per;
    }
    return GPF_OK;
}
// phase 3 of that plan, first half (the one collective): all log-weights on every rank, dense in global order, in the planner's weight array
static gpf_status shard_sorted_gather(gpf_filter* h, int G, int me)
{
    gpf_status s;
    gpf_filter* p = h->planner;
    const int64_t N = h->cfg.n_global;
    if (N % G == 0) return shard_all_gather(h, h->lw, p->lw, (size_t)h->n, ncclDouble, sizeof(double));
    HIP_TRY(h, hipMemcpyAsync(h->sorted_src, h->lw, (size_t)h->n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if ((s = shard_all_gather(h, h->sorted_src, h->sorted_gath, (size_t)h->sorted_per, ncclDouble, sizeof(double)))) return s;
    GPF_LAUNCH(k_anc_compact, dim3(grid_for(h, N, 8)), dim3(BLOCK), 0, h->stream, anc_plan(h, G, me), h->sorted_gath, h->sorted_per, p->lw);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// ... second half: the unsharded sort + scan + search on them (the planner), the exchange counts
static gpf_status shard_sorted_plan(gpf_filter* h, int G, int me, bool own)
{
    gpf_status s;
    gpf_filter* p = h->planner;
    const int64_t N = h->cfg.n_global;
    // the planner IS the unsharded filter as far as its weights go: same seed, same epoch, same K = fix_K(n_global)
    p->epoch = h->epoch;
    p->pending_gather = false; p->pending_search = false;
    p->max_valid = false; p->raw_valid = false; p->raw_sum_valid = false;
    mutated(p);
    s = resample_impl(p, GPF_RESAMPLE_STRATIFIED, raw_view(p), 1, GPF_CHECK_FALSE, nullptr);   // (validity was decided on the gathered flags)
    p->pending_gather = false;                                    // (its ancestors are all this call wants: the planner has no rows worth gathering)
    if (s) return fail(h, s, "planner: " + p->err);
    HIP_TRY(h, hipMemsetAsync(h->shard_counts, 0, (size_t)2 * MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t), h->stream));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((N + ANC_CHUNK - 1) / ANC_CHUNK, (int64_t)h->n_cu * 4));
    s = timed(h, GPF_K_SEARCH, [&] { GPF_LAUNCH(k_anc_count, dim3(grid), dim3(ANC_BLOCK), 0, h->stream, anc_plan(h, G, me), own ? 1 : 0, h->anc, h->shard_counts); });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->counts_published = false;                                  // (gpf_shard_counts copies them from the device)
    h->push_counted = true;
    return GPF_OK;
}
static gpf_status shard_sorted_push(gpf_filter* h, int G, int me, bool own, int64_t capacity, double* packed_out)
{
    if (capacity <= 0 || !packed_out) return GPF_OK;
    HIP_TRY(h, hipMemsetAsync(h->anc_cursors, 0, MAX_SHARDS * sizeof(unsigned long long), h->stream));
    const int64_t N = h->cfg.n_global;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((N + ANC_CHUNK - 1) / ANC_CHUNK, (int64_t)h->n_cu * 4));
    const AncPlan ap = anc_plan(h, G, me);
    const double* rows = h->rows[h->cur];
    gpf_status s = timed(h, GPF_K_GATHER, [&] {
        switch (h->W) {
            case 2: GPF_LAUNCH((k_anc_pack<2>), dim3(grid), dim3(ANC_BLOCK), 0, h->stream, ap, own ? 1 : 0, rows, h->shard_counts, h->anc_cursors, capacity, packed_out); break;
            case 4: GPF_LAUNCH((k_anc_pack<4>), dim3(grid), dim3(ANC_BLOCK), 0, h->stream, ap, own ? 1 : 0, rows, h->shard_counts, h->anc_cursors, capacity, packed_out); break;
            case 8: GPF_LAUNCH((k_anc_pack<8>), dim3(grid), dim3(ANC_BLOCK), 0, h->stream, ap, own ? 1 : 0, rows, h->shard_counts, h->anc_cursors, capacity, packed_out); break;
        }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// phase 5 of a window exchange: the new population = the shard's own range through the ancestor array + the window entries of the other slots
static gpf_status shard_commit_ring(gpf_handle h, const double* mf_all, const int64_t* tot_all, int G, bool own_is_a_range)
{
    h->pend_own = true; h->pend_m = 0; h->pend_own_range = own_is_a_range;   // (else: the own hits are the slots with anc >= 0, k_search_own)
    h->pending_packed = true;
    h->pend_packed = nullptr; h->pend_mf = mf_all; h->pend_tot = tot_all; h->pend_G = G;
    h->pend_mailbox = h->mb_active && h->mb_engine;
    h->pend_ring = true; h->pend_ring_seq = h->ring_seq;
    h->epoch += 1;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->max_valid = false;
    h->residual_scanned = false;
    h->push_counted = false;
    mutated(h);
    return GPF_OK;
}

static gpf_status shard_resample_impl(gpf_handle h, int32_t method, double priority_alpha, int32_t check, int32_t* invalid, int32_t sort_particles = 0)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED && method != GPF_RESAMPLE_MULTINOMIAL_SORTED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");             // resample.jl:28
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_shard_resample needs gpf_comm_create first");
    // stratified over the particles in descending weight order (resample.jl:145,156-157): the replicated plan (gpf_k_shard.hpp AncPlan) -- ancestors in slot
    // order are then a permutation's worth of shards, the rows travel like the i.i.d. resamplers'
    const bool sorted = method == GPF_RESAMPLE_STRATIFIED && sort_particles;
    const bool ranged = (method == GPF_RESAMPLE_STRATIFIED && !sorted) || method == GPF_RESAMPLE_MULTINOMIAL_SORTED;   // ascending targets: every shard serves ONE slot range
    const int G = h->comm_world, me = h->comm_rank;
    // priority_fn = w -> alpha w (resample.jl:51-52): ancestors from the priorities' CDF, the log-ML update from the RAW weights
    // (:57), new log-weights log_ws + (log N - logsumexp(log_ws)) with log_ws = lw[a] - lp[a] (:198-200).  Across shards that is
    // three summary rounds instead of one (raw weights, priorities, log_ws) and one more double per exchanged entry (log_ws: the
    // receiver does not hold its ancestors' weights); the commit cannot be deferred (the weights need the third round).
    const bool prio = priority_alpha == priority_alpha;
    const int64_t n = h->n, E = h->W + 1 + (prio ? 1 : 0);
    if (sorted && prio) return fail(h, GPF_ERR_INVALID_ARGUMENT, "sort_particles = true with a priority_fn is not available across shards");
    EngineScope engine(h);                                        // the phases below push / wait through the shard mailboxes when they are up
    phase_mark(h, -1);
    if (h->phases.on) h->phases.resamples += 1;
    // shard bounds from the contiguous-range rule every rank applies to its own gpf_config (ranks ordered by gid0)
    std::vector<int64_t> bounds((size_t)G + 1);
    {
        const int64_t base = h->cfg.n_global / G, extra = h->cfg.n_global % G;
        for (int g = 0; g <= G; ++g) bounds[g] = (int64_t)g * base + std::min<int64_t>(g, extra);
        if (bounds[me] != h->cfg.gid0 || bounds[me + 1] != h->cfg.gid0 + n)
            return fail(h, GPF_ERR_STATE, "this shard's (gid0, n_particles) is not rank's contiguous share of n_global");
    }
    // Everything that can fail for reasons of THIS rank alone (allocations) happens before the first collective of the call: a
    // rank that returned early would leave its peers blocked in a collective it never joins.  The send buffer holds a balanced
    // exchange with slack -- or, when that is cheap against the HBM at hand (<= 1/16 of the free memory), one entry per GLOBAL slot,
    // the most any shard can ever serve, so that the overflow re-push below never has to allocate.
    auto ensure = [&](double*& buf, int64_t& cap, int64_t want) -> gpf_status {
        if (cap >= want) return GPF_OK;
        if (buf) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(buf); buf = nullptr; cap = 0; }
        HIP_TRY(h, hipMalloc(&buf, (size_t)want * (h->W + 2) * sizeof(double)));   // (room for the widest entry: a prioritised resample's [row | meta | log_ws])
        cap = want;
        return GPF_OK;
    };
    int64_t cap = std::min<int64_t>(h->cfg.n_global, 2 * n + 65536);
    if (h->sh_send_cap < h->cfg.n_global) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)h->cfg.n_global * (h->W + 2) * sizeof(double) <= free_b / 16) cap = h->cfg.n_global;
    } else cap = h->cfg.n_global;
    if (const char* e = getenv("GPF_PUSH_CAPACITY")) cap = atoll(e);                                  // tests: force the overflow path
    if ((s = ensure(h->sh_send, h->sh_send_cap, std::max<int64_t>(cap, 1)))) return s;
    static const bool force = getenv("GPF_SHARD_FORCE_COLLECTIVES") && !strcmp(getenv("GPF_SHARD_FORCE_COLLECTIVES"), "1");
    const bool exchange = G > 1 || (force && h->comm);            // one shard: what it "sends" is what it "receives"
    if (exchange && (s = ensure(h->sh_recv, h->sh_recv_cap, n))) return s;
    const bool pull = h->shard_plan_kind == GPF_SHARD_PLAN_PULL && !ranged && !sorted;
    if (pull && (s = pull_buffers(h, G))) return s;
    if (sorted && ((s = materialize(h)) || (s = shard_sorted_buffers(h, G)))) return s;
    // Own-direct (multinomial, push plan, no priorities): a slot of this shard whose target falls into this shard's own part of the CDF is
    // resolved in place -- its ancestor goes into h->anc and the next propagate gathers the row through it, as on an unsharded filter;
    // only the slots other shards serve travel as packed entries.  On one rank nothing is staged, packed, counted or waited for.
    static const bool own_off = getenv("GPF_SHARD_OWN") && !strcmp(getenv("GPF_SHARD_OWN"), "0");        // (A/B measurements; tests of the packed path)
    // (own hits sit in the int32 ancestor array as GLOBAL ids, -1 = "arrives packed": needs n_global < 2^31 -- gpf_create guarantees it)
    const bool own = !own_off && !prio && h->cfg.n_global < ((int64_t)1 << 31) && ((method == GPF_RESAMPLE_MULTINOMIAL && !pull && multi_logg(h->ntiles) >= 0 &&
                                            multi_lds_bytes(h->ntiles, multi_logg(h->ntiles)) + 4096 <= (size_t)160 * 1024) ||
                                           (method == GPF_RESAMPLE_RESIDUAL && !pull && search_lds_bytes(h->ntiles, 2) + 40 * 1024 <= (size_t)160 * 1024) ||
                                           ranged || sorted);
    struct OwnScope { gpf_filter* h; ~OwnScope() { h->own_direct = false; h->own_direct_range = false; h->ring_now = false; h->splan_ride = false; h->splan_done = false; } } own_scope{h};
    h->splan_ride = method == GPF_RESAMPLE_STRATIFIED && !prio && !sorted;
    h->own_direct = own; h->own_direct_range = own && ranged;
    // The window exchange (gpf_k_common.hpp RingOut / RingIn; gpf_comm_set_exchange): the resamplers with ascending targets exchange boundary slabs -- the
    // merge kernel stores them straight into the destination ranks' slot-addressed receive windows, the next propagate reads them there.  No split
    // sizes for the host to wait for, no ncclGroup, no send / receive buffer, no overflow; the call returns as soon as its kernels are enqueued.
    // (GPF_SHARD_EXCHANGE_P2P_ALL: the i.i.d. resamplers' rows too -- every entry names its slot, so the window takes them as it takes the slabs; that
    //  exchange is bandwidth-bound, (G-1)/G of all rows as scattered 8 (W + 2)-byte peer stores: opt-in until a multi-GPU run has timed it against RCCL)
    const bool p2p = own && h->ring_active && ((ranged && h->exchange_mode >= GPF_SHARD_EXCHANGE_P2P) || (!ranged && !pull && !sorted && h->exchange_mode == GPF_SHARD_EXCHANGE_P2P_ALL));
    h->ring_now = p2p;
    // sorted multinomial: the tile totals of ALL global slots (they depend on seed, epoch and N alone) -- the job rides in the weight scan below
    struct SpScope { gpf_filter* h; ~SpScope() { h->sp_job_set = false; } } sp_scope{h};
    if (method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        if ((s = materialize(h))) return s;
        if ((s = sorted_job_prepare(h, 0, h->cfg.n_global))) return s;
        if (!h->shard_plan) HIP_TRY(h, hipMalloc(&h->shard_plan, sizeof(ShardPlan)));
        if (!h->splan_F) {
            HIP_TRY(h, hipMalloc(&h->splan_F, (MAX_SHARDS + 1) * sizeof(int64_t)));
            HIP_TRY(h, hipMalloc(&h->splan_arrive, sizeof(unsigned int)));
            HIP_TRY(h, hipMemsetAsync(h->splan_arrive, 0, sizeof(unsigned int), h->stream));
        }
    }

    const double* raw_mf = nullptr; const int64_t* raw_tot = nullptr;
    struct PushScope { gpf_filter* h; ~PushScope() { h->push_extra = 0; } } push_scope{h};
    if (prio) {
        if ((s = materialize(h))) return s;
        h->want_offsets = false;
        s = shard_summary_of(h, raw_view(h), &h->sc->raw, false);  // logsumexp(log_weights), every shard (resample.jl:180)
        h->want_offsets = true;
        if (s) return s;
        raw_mf = h->cur_mf_all; raw_tot = h->cur_tot_all;
        h->push_extra = 1; h->push_pv = PrioView{h->lw, nullptr, priority_alpha, 1};
    }
    // An ESS read stands in front of this resample (README.md:68-70; gpf_shard_step_ess, or the getter) and its one-launch reduction (k_sum_shard) has
    // exchanged (max, flags) and {S, limbs} already -- the weights and the mailbox's rounds are still those: the (max, flags) round is not repeated, and a
    // RESIDUAL resample, which samples from copy counts and residual weights, never from the weight CDF, runs no weight scan at all (the unsharded
    // filter's direct form).  Every rank holds the same summary, so every rank takes the same branch.
    static const bool reuse_off = getenv("GPF_SHARD_REUSE_SUMMARY") && !strcmp(getenv("GPF_SHARD_REUSE_SUMMARY"), "0");    // (A/B measurements, tests of the plain path)
    const bool reuse = !reuse_off && !prio && gsum_valid(h);
    const bool resid_direct = reuse && method == GPF_RESAMPLE_RESIDUAL;
    const WSum gsum = h->gsum;
    h->gsum_ok = false;                                           // (whatever follows starts new rounds or changes the weights)
    if (!resid_direct) {
        h->want_offsets = method == GPF_RESAMPLE_MULTINOMIAL;     // (the offset levels serve k_push_multi only)
        s = prio ? shard_summary_of(h, h->push_pv, &h->sc->prio, true) : shard_summary(h, 0, reuse);   // phases 1, 2 (safe_softmax of the priorities, :54)
        h->want_offsets = true;
        if (s) return s;
    }
    if (check != GPF_CHECK_FALSE || invalid) {                    // safe_softmax validity (utils.jl:117-140): pinned flags, no stream sync
        // (the flags describe the GLOBAL weights: every rank takes the same branch here)
        int32_t flags = resid_direct ? gsum.flags : 0;
        if (!resid_direct && (s = gpf_shard_flags(h, &flags))) return s;
        if (invalid) *invalid = flags != 0;
        if (flags & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && flags) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    const int64_t* cr_all = nullptr;
    const int64_t* tot_all = h->cur_tot_all;
    if (method == GPF_RESAMPLE_RESIDUAL) {                        // phase 2b
        if (resid_direct) { if ((s = shard_residual_scan_direct(h, h->cur_mf_all, gsum.S, G, h->sh_cr))) return s; }
        else if ((s = gpf_shard_residual_scan(h, tot_all, G, h->sh_cr))) return s;
        if (!h->mb_active && (s = shard_all_gather(h, h->sh_cr, h->sh_cr_all, 2, ncclInt64, sizeof(int64_t)))) return s;
        cr_all = h->cur_cr_all = h->mb_active ? static_cast<const int64_t*>(mb_gathered(h, MB_CR)) : h->sh_cr_all;
    }
    phase_mark(h, GPF_PHASE_SUMMARIES);
    if (p2p) {
        h->ring_seq += 1;                                         // (the same on every rank: SPMD call order)
        if ((s = gpf_shard_push_count(h, method, tot_all, cr_all, G, me, bounds.data()))) return s;             // the plan: served range, own range (device only)
        phase_mark(h, GPF_PHASE_PLAN);
        // own slots in place, the others into their ranks' windows (one rank, i.i.d. targets: every slot is an own hit, nothing to look up for anybody else)
        if ((ranged || G > 1) && (s = gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), h->cfg.n_global, nullptr))) return s;
        phase_mark(h, GPF_PHASE_PACK);
        h->tr_calls += 1; h->tr_entry_bytes = (int64_t)(h->W + 2) * (int64_t)sizeof(double);
        return shard_commit_ring(h, h->cur_mf_all, tot_all, G, ranged);   // deferred: the next propagate reads window and own hits in ONE launch
    }
    std::vector<int64_t> counts(2 * (size_t)G);
    int64_t pushed_cap = std::min(cap, h->sh_send_cap);
    int64_t n_send = 0, n_recv = 0;
    if (pull) {
        // phase 3 of the pull plan: requests out, counts known on the host BEFORE pass 2 is enqueued (no speculative capacity)
        if ((s = pull_requests(h, method, tot_all, cr_all, G, me, bounds.data(), exchange, force && G == 1, counts))) return s;
        phase_mark(h, GPF_PHASE_PLAN);
        for (int g = 0; g < G; ++g) { n_send += counts[g]; n_recv += counts[G + g]; }
        pushed_cap = std::min(n_send, h->sh_send_cap);
        if ((s = gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), pushed_cap, h->sh_send))) return s;
        phase_mark(h, GPF_PHASE_PACK);
    } else {
        if ((s = sorted ? ((s = shard_sorted_gather(h, G, me)) ? s : shard_sorted_plan(h, G, me, own)) : gpf_shard_push_count(h, method, tot_all, cr_all, G, me, bounds.data()))) return s;   // phase 3
        phase_mark(h, GPF_PHASE_PLAN);
        if (own && G == 1) {
            // one shard, own-direct: every slot is an own hit -- nothing to exchange, so no split sizes to wait for (the host wait left a
            // gap in the queue on every resample); the i.i.d. methods have nothing to push either, stratified writes its ancestors in
            // place from the merge kernel
            counts[0] = ranged ? n : 0; counts[1] = n;
            if (ranged && (s = gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), pushed_cap, h->sh_send))) return s;
            phase_mark(h, GPF_PHASE_PACK);
        } else {
        // phase 4 is enqueued before the host learns the counts; the kernel stops at the capacity and the push is repeated if the
        // counts say it overflowed
        if ((s = sorted ? shard_sorted_push(h, G, me, own, pushed_cap, h->sh_send) : gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), pushed_cap, h->sh_send))) return s;
        phase_mark(h, GPF_PHASE_PACK);
        const auto w0 = std::chrono::steady_clock::now();
        if ((s = gpf_shard_counts(h, G, counts.data()))) return s;   // ONE host wait (the exchange's split sizes), behind phase 4
        if (h->phases.on) h->phases.host_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        }
        for (int g = 0; g < G; ++g) { n_send += counts[g]; n_recv += counts[G + g]; }
    }
    // From here on a local failure is REMEMBERED and the rank still joins the exchange with the counts its peers expect (they
    // worked out their receive counts themselves and will wait for exactly that many entries): the error is returned after the
    // group has closed.  A failed gpf_shard_resample leaves the communicator and the filter unusable on every rank that sees
    // one (the entries a failing rank sends are undefined): the host must tear the job down.
    h->tr_calls += 1; h->tr_entry_bytes = E * (int64_t)sizeof(double);
    for (int g = 0; g < G; ++g) if (g != me) { h->tr_sent += counts[g]; h->tr_recv += counts[(size_t)G + g]; }
    gpf_status late = GPF_OK;
    std::string late_msg;
    auto remember = [&](gpf_status st) { if (st && !late) { late = st; late_msg = h->err; } };
    if (n_recv != n) remember(fail(h, GPF_ERR_STATE, "exchange counts do not add up to the shard's slots"));
    if (n_send > h->sh_send_cap) {                                // skewed weights: this shard serves more than its buffer held
        gpf_status es = ensure(h->sh_send, h->sh_send_cap, n_send);
        if (es) {                                                 // cannot hold what the peers expect: nothing sane can be sent
            if (exchange) remember(fail(h, GPF_ERR_HIP, "send buffer for a skewed exchange could not be allocated; the communicator is poisoned (peers are waiting)"));
            return es;
        }
    }
    if (n_send > pushed_cap)
        remember(sorted ? shard_sorted_push(h, G, me, own, n_send, h->sh_send) : gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), n_send, h->sh_send));
    // the exchange: [row | slot | ancestor id], grouped point-to-point sends and receives (one pair per PEER; the shard's own
    // entries never touch RCCL: one device-to-device copy on the same stream)
    const double* commit_from = h->sh_send;
    if (exchange) {
        commit_from = h->sh_recv;
        ncclResult_t first = ncclSuccess;
        const char* where = "";
        auto note = [&](ncclResult_t r, const char* w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; where = w; } };
        const bool recv_fits = n_recv <= h->sh_recv_cap;
        int64_t so = 0, ro = 0, self_so = -1, self_ro = -1;
        note(g_rccl.GroupStart(), "ncclGroupStart");
        for (int g = 0; g < G; ++g) {
            if (g == me && own) { so += counts[g]; continue; }    // own hits never enter the exchange (their places in the send buffer stay unused; counts[G + me] of them sit in h->anc)
            if (g == me && !(force && G == 1)) { self_so = so; self_ro = ro; }
            else {
                if (counts[g]) note(g_rccl.Send(h->sh_send + so * E, (size_t)(counts[g] * E), ncclDouble, g, h->comm, h->stream), "ncclSend");
                if (counts[G + g] && recv_fits) note(g_rccl.Recv(h->sh_recv + ro * E, (size_t)(counts[G + g] * E), ncclDouble, g, h->comm, h->stream), "ncclRecv");
            }
            so += counts[g]; ro += counts[G + g];
        }
        note(g_rccl.GroupEnd(), "ncclGroupEnd");                  // ALWAYS closed: librccl is shared with the host (PyTorch); an open group
                                                                  // would swallow every later RCCL call of this thread
        if (first != ncclSuccess) remember(fail(h, GPF_ERR_HIP, std::string(where) + ": " + g_rccl.GetErrorString(first)));
        if (self_so >= 0 && counts[me] && recv_fits) {
            const hipError_t ce = hipMemcpyAsync(h->sh_recv + self_ro * E, h->sh_send + self_so * E, (size_t)counts[me] * E * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
            if (ce != hipSuccess) remember(fail(h, GPF_ERR_HIP, std::string("self copy: ") + hipGetErrorString(ce)));
        }
    }
    phase_mark(h, GPF_PHASE_EXCHANGE);
    if (late) { h->err = late_msg; h->comm_poisoned = true; return late; }
    if (!prio) return gpf_shard_commit(h, commit_from, own ? n - counts[(size_t)G + me] : n, h->cur_mf_all, tot_all, G);   // phase 5 (deferred)
    // phase 5 of a prioritised resample, at once: scatter rows / parents / log_ws, log-ML from the raw summary ...
    {
        const int grid = grid_for(h, n, 8);
        double* out = h->rows[1 - h->cur];
        const int mbx = (int)h->mb_active;
        switch (h->W) {
            case 2: GPF_LAUNCH((k_commit_packed_ws<2>), dim3(grid), dim3(BLOCK), 0, h->stream, commit_from, n, out, h->anc, h->lws, raw_mf, raw_tot, G, h->K, h->logN, h->sc, mbx); break;
            case 4: GPF_LAUNCH((k_commit_packed_ws<4>), dim3(grid), dim3(BLOCK), 0, h->stream, commit_from, n, out, h->anc, h->lws, raw_mf, raw_tot, G, h->K, h->logN, h->sc, mbx); break;
            case 8: GPF_LAUNCH((k_commit_packed_ws<8>), dim3(grid), dim3(BLOCK), 0, h->stream, commit_from, n, out, h->anc, h->lws, raw_mf, raw_tot, G, h->K, h->logN, h->sc, mbx); break;
        }
        HIP_TRY(h, hipGetLastError());
        h->cur ^= 1;
    }
    // ... then logsumexp(log_ws) over all shards and lw = log_ws + (log N - logsumexp) (resample.jl:200)
    h->want_offsets = false;
    s = shard_summary_of(h, PrioView{h->lws, nullptr, 0.0, 0}, &h->sc->post, false);
    h->want_offsets = true;
    if (s) return s;
    GPF_LAUNCH(k_shard_apply_post, dim3(grid_for(h, n, 8)), dim3(BLOCK), 0, h->stream, h->cur_mf_all, h->cur_tot_all, G, h->K, h->logN, h->lws, h->lw, n,
               mb_wait(h, MB_TOT));
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = false; h->residual_scanned = false; h->push_counted = false;
    mutated(h);
    return GPF_OK;
}


} // extern "C"
namespace {
gpf_status shard_global_summary_launch(gpf_filter* h, double thr, bool* done);
bool shard_sum_fits(const gpf_filter* h);
} // namespace
extern "C" {
// One iteration of the README loop (README.md:66-77) on a sharded filter, one call per rank (every rank calls it, like gpf_shard_resample):
//     if effective_sample_size(state) < ess_frac * N_global;  pf_resample!(state, method);  pf_rejuvenate!(state, ...);  end;  pf_update!(state, ...)
// The verdict needs the GLOBAL sums: the summary reduction (k_sum_reduce<SHARD>) exchanges the shard totals through the mailboxes, leaves the
// verdict on the device and the global summary in pinned memory; the propagate runs speculatively behind it (as gpf_step_ess).
// Every rank folds the same global integers: all ranks take the same branch.  Without mailboxes / with pending work: the plain sequence.
gpf_status gpf_shard_step_ess(gpf_handle h, const double* obs, int32_t n_obs, double ess_frac, int32_t method, int32_t check,
                              int32_t rejuvenate_method, int32_t n_iters, int32_t* resampled, int32_t* invalid)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_shard_step_ess needs gpf_comm_create first");
    if (!(ess_frac == ess_frac) || ess_frac < 0.0) return fail(h, GPF_ERR_INVALID_ARGUMENT, "ess_frac must be >= 0");
    if (rejuvenate_method >= 0 && rejuvenate_method != GPF_REJUVENATE_MOVE && rejuvenate_method != GPF_REJUVENATE_REWEIGHT)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Method not recognized.");                        // rejuvenate.jl:25
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED && method != GPF_RESAMPLE_MULTINOMIAL_SORTED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");             // resample.jl:28
    if (rejuvenate_method >= 0 && !h->cfg.keep_prev) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate needs keep_prev = 1 (x_{t-1} must travel with the particle)");
    if (resampled) *resampled = 0;
    if (invalid) *invalid = 0;
    static const bool spec_off = getenv("GPF_STEP_SPECULATE") && !strcmp(getenv("GPF_STEP_SPECULATE"), "0");
    // one shard without a communicator IS the unsharded filter (n_global == n): the unsharded call, sort_particles = false
    if (!spec_off && h->comm_world == 1 && !h->comm && !h->pending_packed)
        return gpf_step_ess(h, obs, n_obs, ess_frac, method, 0, check, rejuvenate_method, n_iters, resampled, invalid, nullptr);
    const double thr = ess_frac * (double)h->cfg.n_global;
    const bool fast = !spec_off && h->mb_active && shard_sum_fits(h) && !h->pending_packed && !h->pending_gather && !h->pending_fill && !h->pending_move &&
                      !h->hist_on && h->blk_obs_size == 0 && obs != nullptr && n_obs == model_obs_dim(h->cfg.model);
    // (the launch says "not available" before it has begun any mailbox round: the plain sequence is still open then)
    bool done = false;
    if (fast && (s = shard_global_summary_launch(h, thr, &done))) return s;
    if (!done) {
        double ess = 0.0;
        if ((s = gpf_shard_effective_sample_size(h, &ess))) return s;
        if (ess < thr) {
            if ((s = shard_resample_impl(h, method, std::nan(""), check, invalid))) return s;
            if (resampled) *resampled = 1;
            if (rejuvenate_method >= 0 && (s = gpf_rejuvenate(h, rejuvenate_method, n_iters, nullptr))) return s;
        }
        return gpf_update(h, obs, n_obs);
    }
    const ModelArgs old_args = h->args;                          // (a rejuvenation moves under the CURRENT step's observation)
    const int mcur0 = h->mcur;
    // (the mailbox rounds of this step have begun: a failure of THIS rank from here on leaves its peers a step ahead of it -- they will wait for its next
    //  round until the mailbox wait gives up; the handle remembers, and every later sharded call on it fails at once instead of joining out of step)
    auto undo = [&](gpf_status st) { h->args = old_args; h->mcur = mcur0; h->comm_poisoned = true; return st; };
    if ((s = set_obs(h, obs, n_obs))) return undo(s);
    GateIn gate{}; gate.flag = &h->sc->gate_go;
    if ((s = speculative_step(h, gate))) return undo(s);
    WSum w{};
    if ((s = read_published_summary(h, w))) return undo(s);
    h->gsum = w; h->gsum_ok = true;                              // (a resample behind this verdict reuses the exchanged summary: shard_resample_impl)
    uint64_t hi, lo;
    normalise_Q(w, hi, lo);
    const bool go = !w.flags && ess_from(w.S, hi, lo) < thr;     // (the device's verdict: the same integers through the same operations)
    if (!go) { speculative_step_done(h, true); return GPF_OK; }
    speculative_step_done(h, false);
    h->args = old_args;
    if ((s = shard_resample_impl(h, method, std::nan(""), check, invalid))) return s;
    if (resampled) *resampled = 1;
    if (rejuvenate_method >= 0 && (s = gpf_rejuvenate(h, rejuvenate_method, n_iters, nullptr))) return s;
    return gpf_update(h, obs, n_obs);
}

gpf_status gpf_phase_timing(gpf_handle h, int32_t enable)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& m : h->phases.marks) (void)hipEventDestroy(m.second);
    h->phases.marks.clear();
    h->phases.host_wait_us = 0.0; h->phases.resamples = 0;
    h->phases.on = enable != 0;
    return GPF_OK;
}
gpf_status gpf_phase_times(gpf_handle h, double* us6, int64_t* resamples)
{
    if (!h || !us6) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < GPF_PHASE_COUNT; ++k) us6[k] = 0.0;
    const auto& mk = h->phases.marks;
    for (size_t i = 1; i < mk.size(); ++i) {
        if (mk[i].first < 0 || mk[i].first >= GPF_PHASE_COUNT) continue;
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, mk[i - 1].second, mk[i].second));
        us6[mk[i].first] += 1e3 * (double)ms;
    }
    us6[GPF_PHASE_HOST_WAIT] = h->phases.host_wait_us;
    if (resamples) *resamples = h->phases.resamples;
    return GPF_OK;
}
gpf_status gpf_shard_resample(gpf_handle h, int32_t method, int32_t check, int32_t* invalid)
{
    return shard_resample_impl(h, method, std::nan(""), check, invalid);
}
// the phases of gpf_shard_resample_sorted for hosts that bring their own collectives (sharded.py's python engine): between the usual summary phases
// (gpf_shard_weight_max / _weight_scan) and gpf_shard_counts / the exchange / gpf_shard_commit
gpf_status gpf_shard_sorted_count(gpf_handle h, const double* lw_all, int32_t G, int32_t me)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    const int64_t N = h->cfg.n_global;
    if (!lw_all || G < 1 || G > MAX_SHARDS || me < 0 || me >= G) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if ((int64_t)me * (N / G) + std::min<int64_t>(me, N % G) != h->cfg.gid0 || N / G + (me < N % G ? 1 : 0) != h->n)
        return fail(h, GPF_ERR_STATE, "this shard's (gid0, n_particles) is not rank's contiguous share of n_global");
    if ((s = materialize(h)) || (s = shard_sorted_buffers(h, G))) return s;
    HIP_TRY(h, hipMemcpyAsync(h->planner->lw, lw_all, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return shard_sorted_plan(h, G, me, h->own_direct);
}
gpf_status gpf_shard_sorted_push(gpf_handle h, int32_t G, int32_t me, int64_t capacity, double* packed_out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!h->planner || !h->push_counted) return fail(h, GPF_ERR_STATE, "gpf_shard_sorted_push needs gpf_shard_sorted_count of the same resample first");
    if (G < 1 || G > MAX_SHARDS || me < 0 || me >= G || capacity < 0 || (capacity > 0 && !packed_out)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    return shard_sorted_push(h, G, me, h->own_direct, capacity, packed_out);
}
gpf_status gpf_shard_resample_sorted(gpf_handle h, int32_t check, int32_t* invalid)
{
    return shard_resample_impl(h, GPF_RESAMPLE_STRATIFIED, __builtin_nan(""), check, invalid, 1);
}
gpf_status gpf_shard_resample_tempered(gpf_handle h, int32_t method, double priority_alpha, int32_t check, int32_t* invalid)
{
    if (!(priority_alpha == priority_alpha)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "priority_alpha is NaN: use gpf_shard_resample for priority_fn = nothing");
    return shard_resample_impl(h, method, priority_alpha, check, invalid);
}


} // extern "C"
namespace {
// The GLOBAL weight summary {flags, S, sum q^2} on the host without a scan, a copy or a stream synchronisation (the getters of an
// ESS-triggered sharded filter run on every step: through shard_summary + shard_scalars one read cost ~85 us of kernels, copies and idle GPU
// on one rank).  One shard without a communicator IS the unsharded filter: its own reduction (k_sum_host).  With the mailboxes up: (max,
// flags) to the peers (k_pack_mflags), then ONE reduction launch whose workgroup 0 exchanges the shard totals through the mailboxes, folds
// the global summary and publishes it to pinned memory (k_sum_reduce<SHARD>); thr >= 0: it also leaves the verdict ESS < thr on the device.
// *done = false: not available (RCCL all-gathers carry the summaries, or the filter is too large): the caller takes shard_summary.
gpf_status shard_global_summary_launch(gpf_filter* h, double thr, bool* done);
gpf_status shard_global_summary(gpf_filter* h, double thr, WSum& w, bool* done)
{
    gpf_status s = shard_global_summary_launch(h, thr, done);
    if (s || !*done) return s;
    if ((s = read_published_summary(h, w))) return s;
    h->gsum = w; h->gsum_ok = true;
    return GPF_OK;
}
gpf_status shard_global_summary_launch(gpf_filter* h, double thr, bool* done)
{
    *done = false;
    gpf_status s;
    if (!h->mb_active || !h->h_timeout) return GPF_OK;
    EngineScope engine(h);
    if ((s = shard_scratch(h))) return s;
    const int r = (int)(h->sh_round++ % SH_RING);
    double* mf = h->sh_mf + 2 * r;
    ShardSum ss{};
    // maximum slots -> (max, flags) -> every peer's mailbox (MB_MF round): its own small launch.  GPF_SHARD_FUSE_MF=1: from inside the reduction's
    // launch (k_sum_shard's workgroup 0 pushes, every workgroup waits) -- 0.7 us faster on one rank, but then EVERY workgroup of a launch that can fill
    // the GPU waits for a push that only the peers' launches of the same kind make: ranks that share one GPU (the loopback tests: 2 - 3 processes on
    // one device, 256 workgroups of 1024 threads each) starve each other until the mailbox wait gives up.  Behind a separate launch the push needs
    // 256 free thread slots somewhere, which a waiting reduction always leaves.
    const bool fuse_mb = mailbox_fuse_mf(h);
    if (shard_sum_collect() || !fuse_mb) { if ((s = gpf_shard_weight_max(h, mf))) return s; }
    else {
        if ((s = shard_max_slots(h))) return s;
        ss.slots = h->mslots[h->mcur]; ss.mf_out = mf; ss.push_mf = mb_begin(h, MB_MF);
    }
    ss.mf_all = static_cast<const double*>(mb_gathered(h, MB_MF)); ss.np = h->comm_world; ss.wait_mf = mb_wait(h, MB_MF);
    ss.push_tot = mb_begin(h, MB_TOT); ss.wait_tot = mb_wait(h, MB_TOT); ss.tot_all = static_cast<const int64_t*>(mb_gathered(h, MB_TOT));
    ss.G = h->comm_world; ss.me = h->comm_rank; ss.thr = thr; ss.go = &h->sc->gate_go;
    bool ok = false;
    if ((s = shard_sum_launch(h, ss, &ok))) return s;
    if (!ok) return fail(h, GPF_ERR_STATE, "sharded summary: the filter is too large for the reduction's tagged partials");   // (the rounds are begun: no way back to the scan)
    h->cur_mf_all = ss.mf_all; h->cur_tot_all = ss.tot_all;
    h->raw_valid = false; h->raw_sum_valid = false;              // (sc->raw holds this shard's sums under the GLOBAL maximum: not the unsharded summary)
    h->gsum_ok = false; h->gsum_mut = h->mutations; h->gsum_mf_seq = h->mb_cur[MB_MF]; h->gsum_tot_seq = h->mb_cur[MB_TOT];   // (valid once the host has read it)
    *done = true;
    return GPF_OK;
}
bool shard_sum_fits(const gpf_filter* h)
{
    if (!shard_sum_collect()) return true;                       // k_sum_shard: 64-bit accumulators, any size
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->n_cu * 4));
    return (h->ntiles + grid - 1) / grid <= Q_TAG_MAX_TILES;
}
} // namespace
extern "C" {
gpf_status gpf_shard_effective_sample_size(gpf_handle h, double* out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "needs gpf_comm_create first");
    static const bool fast_off = getenv("GPF_SHARD_GETTERS") && !strcmp(getenv("GPF_SHARD_GETTERS"), "scan");      // (A/B measurements, tests of the scan + copy path)
    if (!fast_off && h->comm_world == 1 && !h->comm && !h->pending_packed) return gpf_effective_sample_size(h, out);   // one shard IS the unsharded filter
    if (!fast_off && h->mb_active && shard_sum_fits(h)) {
        if ((s = materialize(h))) return s;
        WSum w{}; bool done = false;
        if ((s = shard_global_summary(h, -1.0, w, &done))) return s;
        if (done) {
            uint64_t hi, lo;
            normalise_Q(w, hi, lo);
            *out = w.flags ? std::nan("") : ess_from(w.S, hi, lo);
            return GPF_OK;
        }
    }
    EngineScope engine(h);
    if ((s = shard_summary(h, 1))) return s;
    double m; int flags; uint64_t S, Qhi, Qlo;
    if ((s = shard_scalars(h, m, flags, S, Qhi, Qlo))) return s;
    *out = flags ? std::nan("") : ess_from(S, Qhi, Qlo);
    return GPF_OK;
}

gpf_status gpf_shard_log_ml_estimate(gpf_handle h, double* out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "needs gpf_comm_create first");
    static const bool fast_off = getenv("GPF_SHARD_GETTERS") && !strcmp(getenv("GPF_SHARD_GETTERS"), "scan");
    if (!fast_off && h->comm_world == 1 && !h->comm && !h->pending_packed) return gpf_log_ml_estimate(h, out);   // one shard IS the unsharded filter
    EngineScope engine(h);
    if ((s = shard_summary(h, 0))) return s;
    double m; int flags; uint64_t S, Qhi, Qlo;
    if ((s = shard_scalars(h, m, flags, S, Qhi, Qlo))) return s;
    double base;
    if ((s = gpf_shard_lml_est(h, &base))) return s;
    *out = base + lse_from(m, S, h->K, flags) - h->logN;
    return GPF_OK;
}

} // extern "C"

