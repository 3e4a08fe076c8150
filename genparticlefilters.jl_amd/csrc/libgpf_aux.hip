// libgpf_aux.hip -- what widened around the hot path: block-wise operations on many small filters, weighted statistics,
// sub-state views (src/view.jl), the resize family (src/resize.jl), the trajectory store.
#include "gpf_host.hpp"

using namespace gpf;
using namespace gpfh;

namespace gpfh {

// block-wise propagate / move (ModelArgs::blk_*): the default proposal only, no fused gather (a block resample gathers eagerly)
template <int M>
void launch_init_blk(gpf_filter* h, int grid)
{
    GPF_LAUNCH((k_init<M, 0, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                       h->cfg.gid0, h->n, h->W, h->rows[h->cur], h->lw, next_slots(h));
}
template <int M, bool KEEP>
void launch_step_blk(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, KEEP);
    GPF_LAUNCH((k_step<M, Wc, KEEP, false, 0, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                       h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, next_slots(h), PackedCommit{});
}
template <int M>
void launch_init_blk_strata(gpf_filter* h, int grid)
{
    if constexpr (!Model<M>::HAS_STRATA) { (void)h; (void)grid; return; }
    else GPF_LAUNCH((k_init<M, 2, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                    h->cfg.gid0, h->n, h->W, h->rows[h->cur], h->lw, next_slots(h));
}
template <int M, bool KEEP>
void launch_step_blk_strata(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, KEEP);
    if constexpr (!Model<M>::HAS_STRATA) { (void)h; (void)grid; return; }
    else GPF_LAUNCH((k_step<M, Wc, KEEP, false, 2, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                    h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, next_slots(h), PackedCommit{});
}
template <int M, bool KEEP>
void launch_step_blk_prop(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, KEEP);
    if constexpr (!Model<M>::HAS_PROPOSAL) { (void)h; (void)grid; return; }
    else GPF_LAUNCH((k_step<M, Wc, KEEP, false, 4, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                    h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, next_slots(h), PackedCommit{});
}
template <int M, bool RW>
void launch_move_blk(gpf_filter* h, int grid, int n_iters)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    GPF_LAUNCH((k_move<M, Wc, RW, false, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                       h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                       h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
}
// a block of <= 128 / <= 512 particles is the work of one wave (2 / 8 particles per lane, four blocks per workgroup), a larger one of a workgroup
template <int METHOD, int Wc, bool PRIO>
void launch_block_resample_w(gpf_filter* h, const BlockArgs& a)
{
    if (a.nb <= 2 * WAVE)      GPF_LAUNCH((k_block_resample<METHOD, Wc, WAVE, 2, PRIO>), dim3((unsigned)((a.nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, a);
    else if (a.nb <= 8 * WAVE) GPF_LAUNCH((k_block_resample<METHOD, Wc, WAVE, 8, PRIO>), dim3((unsigned)((a.nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, a);
    else                       GPF_LAUNCH((k_block_resample<METHOD, Wc, BLOCK, 8, PRIO>), dim3((unsigned)a.nblocks), dim3(BLOCK), 0, h->stream, a);
}
template <int METHOD>
void launch_block_resample(gpf_filter* h, const BlockArgs& a, bool prio)
{
    switch (h->W) {
        case 2: if (prio) launch_block_resample_w<METHOD, 2, true>(h, a); else launch_block_resample_w<METHOD, 2, false>(h, a); break;
        case 4: if (prio) launch_block_resample_w<METHOD, 4, true>(h, a); else launch_block_resample_w<METHOD, 4, false>(h, a); break;
        case 8: if (prio) launch_block_resample_w<METHOD, 8, true>(h, a); else launch_block_resample_w<METHOD, 8, false>(h, a); break;
    }
}

} // namespace gpfh

extern "C" {

// ------------------------------------------------------------------ block-wise resampling: many small filters in one launch (K11)
static gpf_status block_buffers(gpf_filter* h, int64_t nblocks)
{
    if (!h->blk_words) HIP_TRY(h, hipMalloc(&h->blk_words, 2 * sizeof(int32_t)));
    if (h->blk_cap < nblocks) {
        if (h->blk_mask) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->blk_mask); (void)hipFree(h->blk_stats); h->blk_mask = nullptr; h->blk_stats = nullptr; h->blk_cap = 0; }
        h->blk_last = 0;                                         // the new mask is uninitialised: no block resample to refer to
        HIP_TRY(h, hipMalloc(&h->blk_mask, (size_t)nblocks * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&h->blk_stats, (size_t)nblocks * 2 * sizeof(double)));
        h->blk_cap = nblocks;
    }
    return GPF_OK;
}
static gpf_status block_checks(gpf_handle h, int64_t block_size, const char* who)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->parent) return fail(h, GPF_ERR_STATE, std::string(who) + " on a sub-state view: call it on the filter");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, std::string(who) + " on a shard of a sharded filter");
    if (block_size < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "block_size < 1");
    return GPF_OK;
}
// Blocks of more than BLK_MAX = 2048 particles do not fit the one-workgroup-per-block kernels (gpf_k_block.hpp keeps a block's weights, CDF
// and order in LDS).  Their loop over sub-states (for b in blocks; pf_resample!(state[b], ...); end -- test/resample.jl:130-162 has no
// size limit) runs on the host over view handles of the blocks, with the full-size kernels: the same results as the views give, the
// same single epoch for all blocks, no size cliff.  At these sizes a block fills the chip by itself.
static gpf_status big_block_views(gpf_filter* h, int64_t block_size)
{
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if (h->blk_views_size == block_size && h->blk_views_gen == h->generation && (int64_t)h->blk_views.size() == nblocks) return GPF_OK;
    for (gpf_filter* v : h->blk_views) gpf_destroy(v);
    h->blk_views.clear();
    for (int64_t b = 0; b < nblocks; ++b) {
        gpf_handle v = nullptr;
        const int64_t start = b * block_size, cnt = std::min(block_size, h->n - start);
        gpf_status s = gpf_view_create(h, start, cnt, &v);
        if (s) return s;
        h->blk_views.push_back(v);
    }
    h->blk_views_size = block_size; h->blk_views_gen = h->generation;
    return GPF_OK;
}
static gpf_status resample_big_blocks(gpf_handle h, int32_t method, int64_t block_size, double priority_alpha, int32_t sort_particles,
                                      double ess_frac, int32_t check, int32_t* invalid, int64_t* n_resampled)
{
    gpf_status s = big_block_views(h, block_size);
    if (s) return s;
    const int64_t nblocks = (int64_t)h->blk_views.size();
    if ((s = block_buffers(h, nblocks))) return s;
    const uint32_t E = h->epoch;                                 // every block resamples under the call's ONE epoch (like the batched kernel)
    std::vector<int32_t> words((size_t)nblocks, 0);
    bool any_invalid = false, any_nan = false, any_neginf_err = false;
    int64_t count = 0;
    const bool gate = ess_frac == ess_frac && ess_frac >= 0.0;
    // (every view reads its validity flags, also under check = false: a NaN block must be left as it stands, as the batched kernel leaves
    //  it -- which costs one pinned-memory wait per block; at > 2048 particles per block the kernels of the block dominate)
    gpf_status hard = GPF_OK;                                    // a failure other than invalid weights: the loop stops, the bookkeeping below still runs
    for (int64_t b = 0; b < nblocks; ++b) {
        gpf_filter* v = h->blk_views[(size_t)b];
        h->epoch = E;
        if (gate) {
            double ess = 0.0;
            if ((s = gpf_effective_sample_size(v, &ess))) { h->err = v->err; hard = s; break; }
            if (!(ess < ess_frac * (double)v->n)) continue;      // (an invalid block: ESS NaN -- it does not resample, nothing is reported)
        }
        int32_t inv = 0;
        v->last_flags = 0;
        s = gpf_resample(v, method, priority_alpha, sort_particles, check == GPF_CHECK_TRUE ? GPF_CHECK_TRUE : GPF_CHECK_WARN, &inv);
        if (s == GPF_ERR_INVALID_WEIGHTS) {                      // the block is left as it stands; the others go on
            any_invalid = true;
            const bool nan_block = (v->last_flags & (FLAG_NAN | FLAG_POSINF)) != 0;   // (the view's own flags, not its error text)
            if (nan_block) any_nan = true; else any_neginf_err = true;
            words[(size_t)b] = (nan_block ? FLAG_NAN : FLAG_ALL_NEGINF) << 8;      // (the word layout of the batched kernel: flags << 8 | resampled)
            continue;
        }
        if (s) { h->err = v->err; hard = s; break; }
        if (inv) { any_invalid = true; words[(size_t)b] |= FLAG_ALL_NEGINF << 8; }
        words[(size_t)b] |= 1;
        ++count;
    }
    // (also on the error path: the blocks before the failing one HAVE resampled under epoch E -- a later call must not reuse their streams,
    //  the mask must name them and the cached summaries are stale)
    h->epoch = E + 1;
    HIP_TRY(h, hipMemcpyAsync(h->blk_mask, words.data(), (size_t)nblocks * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                // (the host vector goes out of scope)
    h->blk_last = nblocks;
    h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; h->max_valid = false;
    mutated(h);
    if (invalid) *invalid = any_invalid ? 1 : 0;
    if (n_resampled) *n_resampled = count;
    if (hard) return hard;
    if (check != GPF_CHECK_FALSE || invalid || n_resampled) {
        if (any_nan) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && (any_neginf_err || any_invalid)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    return GPF_OK;
}
gpf_status gpf_resample_blocks(gpf_handle h, int32_t method, int64_t block_size, double priority_alpha, int32_t sort_particles,
                               double ess_frac, int32_t check, int32_t* invalid, int64_t* n_resampled)
{
    gpf_status s = block_checks(h, block_size, "gpf_resample_blocks");
    if (s) return s;
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");          // resample.jl:28
    if (h->hist_on) return fail(h, GPF_ERR_STATE, "gpf_resample_blocks on a filter with a trajectory store");
    if (h->W != 2 && h->W != 4 && h->W != 8) return fail(h, GPF_ERR_STATE, "row width");
    if ((s = materialize(h))) return s;
    if (block_size > BLK_MAX) return resample_big_blocks(h, method, block_size, priority_alpha, sort_particles, ess_frac, check, invalid, n_resampled);
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if ((s = block_buffers(h, nblocks))) return s;
    BlockArgs a{};
    a.rows_in = h->rows[h->cur]; a.rows_out = h->rows[1 - h->cur]; a.lw = h->lw; a.anc = h->anc;
    a.n = h->n; a.nb = block_size; a.nblocks = nblocks; a.gid0 = h->cfg.gid0; a.seed = h->cfg.seed; a.epoch = h->epoch;
    a.sorted = method == GPF_RESAMPLE_STRATIFIED && sort_particles ? 1 : 0;
    const bool prio = priority_alpha == priority_alpha;
    a.alpha = prio ? priority_alpha : 1.0;
    a.ess_frac = ess_frac == ess_frac ? ess_frac : -1.0;
    a.check_true = check == GPF_CHECK_TRUE ? 1 : 0;
    a.resampled = h->blk_mask;
    s = timed(h, GPF_K_SEARCH, [&] {
        if (method == GPF_RESAMPLE_MULTINOMIAL)   launch_block_resample<0>(h, a, prio);
        else if (method == GPF_RESAMPLE_RESIDUAL) launch_block_resample<1>(h, a, prio);
        else                                      launch_block_resample<2>(h, a, prio);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    h->blk_last = nblocks;
    h->pending_gather = false; h->pending_fill = false;
    h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; h->max_valid = false;
    h->epoch += 1;
    mutated(h);
    if (check != GPF_CHECK_FALSE || invalid || n_resampled) {
        int32_t words[2] = {0, 0};
        GPF_LAUNCH(k_block_summary, dim3(1), dim3(BLOCK), 0, h->stream, h->blk_mask, nblocks, h->blk_words);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(words, h->blk_words, sizeof(words), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (invalid) *invalid = words[0] != 0;
        if (n_resampled) *n_resampled = (int64_t)(uint32_t)words[1];
        if (words[0] & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && words[0]) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    return GPF_OK;
}
gpf_status gpf_block_resampled(gpf_handle h, int32_t* out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if (!h->blk_mask || h->blk_last < 1) return fail(h, GPF_ERR_STATE, "gpf_block_resampled needs gpf_resample_blocks first");
    HIP_TRY(h, hipMemcpyAsync(out, h->blk_mask, (size_t)h->blk_last * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->blk_last; ++i) out[i] &= 1;       // (the words also carry the blocks' validity flags)
    return GPF_OK;
}
gpf_status gpf_block_stats(gpf_handle h, int64_t block_size, double* ess_out, double* lml_out)
{
    gpf_status s = block_checks(h, block_size, "gpf_block_stats");
    if (s) return s;
    if ((s = materialize(h))) return s;
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if (block_size > BLK_MAX) {                                  // the loop over sub-states (big_block_views)
        if ((s = big_block_views(h, block_size))) return s;
        for (int64_t b = 0; b < nblocks; ++b) {
            gpf_filter* v = h->blk_views[(size_t)b];
            if (ess_out && (s = gpf_effective_sample_size(v, ess_out + b))) { h->err = v->err; return s; }
            if (lml_out && (s = gpf_log_ml_estimate(v, lml_out + b))) { h->err = v->err; return s; }
        }
        return GPF_OK;
    }
    if ((s = block_buffers(h, nblocks))) return s;
    if (block_size <= 2 * WAVE)      GPF_LAUNCH((k_block_stats<WAVE, 2>), dim3((unsigned)((nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, h->lw, h->n, block_size, nblocks, &h->sc->lml_est, h->blk_stats, h->blk_stats + nblocks);
    else if (block_size <= 8 * WAVE) GPF_LAUNCH((k_block_stats<WAVE, 8>), dim3((unsigned)((nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, h->lw, h->n, block_size, nblocks, &h->sc->lml_est, h->blk_stats, h->blk_stats + nblocks);
    else                             GPF_LAUNCH((k_block_stats<BLOCK, 8>), dim3((unsigned)nblocks), dim3(BLOCK), 0, h->stream, h->lw, h->n, block_size, nblocks, &h->sc->lml_est, h->blk_stats, h->blk_stats + nblocks);
    HIP_TRY(h, hipGetLastError());
    if (ess_out) HIP_TRY(h, hipMemcpyAsync(ess_out, h->blk_stats, (size_t)nblocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (lml_out) HIP_TRY(h, hipMemcpyAsync(lml_out, h->blk_stats + nblocks, (size_t)nblocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GPF_OK;
}

// the blocks' observation vectors -> device ([n_blocks][MAX_OBS], zero-padded), ModelArgs::blk_* set
static gpf_status set_block_obs(gpf_filter* h, const double* obs, int32_t n_obs, int64_t block_size)
{
    if (!obs || n_obs != model_obs_dim(h->cfg.model))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model takes " + std::to_string(model_obs_dim(h->cfg.model)) + " observation values per step and block");
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if (h->blk_obs_cap < nblocks) {
        if (h->blk_obs) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            (void)hipFree(h->blk_obs); h->blk_obs = nullptr; h->blk_obs_cap = 0;
            for (int k = 0; k < gpf_filter::BLK_STAGE; ++k) { (void)hipHostFree(h->h_blk_obs[k]); h->h_blk_obs[k] = nullptr; }
        }
        HIP_TRY(h, hipMalloc(&h->blk_obs, (size_t)nblocks * MAX_OBS * sizeof(double)));
        for (int k = 0; k < gpf_filter::BLK_STAGE; ++k) HIP_TRY(h, hipHostMalloc(&h->h_blk_obs[k], (size_t)nblocks * MAX_OBS * sizeof(double)));
        if (!h->h_blk_done) {
            HIP_TRY(h, hipHostMalloc(&h->h_blk_done, sizeof(int64_t))); *h->h_blk_done = 0;
            HIP_TRY(h, hipMalloc(&h->blk_stage_counter, sizeof(unsigned int)));
            HIP_TRY(h, hipMemsetAsync(h->blk_stage_counter, 0, sizeof(unsigned int), h->stream));
        }
        h->blk_obs_cap = nblocks;
    }
    // the staging buffers are used in turn: wait only until the copy that last read THIS buffer (four calls ago) has finished -- its
    // kernel publishes a ticket to pinned memory -- not for the stream
    const int k = (int)(h->blk_stage_next % gpf_filter::BLK_STAGE);
    if (h->blk_stage_next >= gpf_filter::BLK_STAGE) {
        const int64_t need = h->blk_stage_next - gpf_filter::BLK_STAGE + 1;
        uint64_t spins = 0;
        while (__atomic_load_n(h->h_blk_done, __ATOMIC_ACQUIRE) < need) {
            cpu_relax();
            if ((++spins & 0x3fff) != 0) continue;
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipErrorNotReady) continue;
            if (__atomic_load_n(h->h_blk_done, __ATOMIC_ACQUIRE) >= need) break;
            return fail(h, GPF_ERR_HIP, q == hipSuccess ? "observation staging: the stream drained without the copy's ticket" : hipGetErrorString(q));
        }
    }
    h->blk_stage_next += 1;
    double* const stage = h->h_blk_obs[k];
    for (int64_t b = 0; b < nblocks; ++b)
        for (int i = 0; i < MAX_OBS; ++i) stage[b * MAX_OBS + i] = i < n_obs ? obs[b * n_obs + i] : 0.0;
    const int64_t n_words = nblocks * MAX_OBS;
    GPF_LAUNCH(k_stage_obs, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (n_words + BLOCK - 1) / BLOCK))), dim3(BLOCK), 0, h->stream,
               stage, h->blk_obs, n_words, h->blk_stage_counter, h->h_blk_done, h->blk_stage_next);
    HIP_TRY(h, hipGetLastError());
    h->args.blk_obs = h->blk_obs; h->args.blk_mask = nullptr; h->args.blk_size = (int32_t)block_size;
    h->blk_obs_size = block_size;
    return GPF_OK;
}
// block_size comes back clamped to the particle count ("one block" may be asked for as any size >= n, 2^32 included): the kernels divide by it as a
// 32-bit number (ModelArgs::blk_size), and n < 2^31
static gpf_status block_step_checks(gpf_handle h, int64_t& block_size, const char* who, const double* obs = nullptr, int32_t n_obs = 0, bool with_obs = false)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (h->parent) return fail(h, GPF_ERR_STATE, std::string(who) + " on a sub-state view: call it on the filter");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, std::string(who) + " on a shard of a sharded filter");
    if (h->hist_on) return fail(h, GPF_ERR_STATE, std::string(who) + " on a filter with a trajectory store");
    if (block_size < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "block_size < 1");      // (the per-block steps index observations by i / block_size: any size)
    block_size = std::min<int64_t>(block_size, std::max<int64_t>(h->n, 1));
    // (callers that change the handle's arguments before set_block_obs -- the strata -- validate the observations first, so that a bad call changes nothing)
    if (with_obs && (!obs || n_obs != model_obs_dim(h->cfg.model)))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model takes " + std::to_string(model_obs_dim(h->cfg.model)) + " observation values per step and block");
    return GPF_OK;
}
gpf_status gpf_initialize_blocks(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_initialize_blocks");
    if (s) return s;
    h->generation += 1;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int grid = step_grid(h);
    s = timed(h, GPF_K_STEP, [&] { DISPATCH_MODEL(h, (launch_init_blk<MM>(h, grid))); });
    if (s) return s;
    h->pending_gather = false; h->pending_fill = false; h->pending_packed = false; h->pending_search = false; h->pending_move = false;
    h->max_valid = true;
    GPF_LAUNCH(k_iota, dim3(grid), dim3(BLOCK), 0, h->stream, h->anc, h->n);            // parents = 1:N (initialize.jl:43)
    HIP_TRY(h, hipMemsetAsync(&h->sc->lml_est, 0, sizeof(double), h->stream));
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->initialized = true; h->has_prev = false; h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;                                             // only_resampled refers to a gpf_resample_blocks of the CURRENT step
    mutated(h);
    return GPF_OK;
}
gpf_status gpf_update_blocks(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_update_blocks");
    if (s) return s;
    if ((s = check_ready(h))) return s;
    if ((s = materialize(h))) return s;                          // (no fused gather in the block-wise step)
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    s = timed(h, GPF_K_STEP, [&] {
        if (keep) { DISPATCH_MODEL(h, (launch_step_blk<MM, true>(h, grid))); }
        else      { DISPATCH_MODEL(h, (launch_step_blk<MM, false>(h, grid))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->max_valid = true;
    h->cur ^= 1;
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;                                             // (as in gpf_initialize_blocks)
    mutated(h);
    return GPF_OK;
}
// for b in blocks: pf_initialize(model, args, observations[b], strata, n_b) / pf_update!(state[b], ..., observations[b], strata) -- stratified
// initialisation / update (src/initialize.jl:92-109, src/update.jl:193-210) of every block by itself, one launch: the stratum of a particle
// follows from its index INSIDE its block and the block's own size (stratified_map!, src/utils.jl:29-55, on the sub-state), the same strata for all blocks
static bool has_strata(gpf_filter* h) { bool v = false; DISPATCH_MODEL(h, (v = Model<MM>::HAS_STRATA)); return v; }
gpf_status gpf_initialize_blocks_strata(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size, const double* values, int32_t n_strata, int32_t interleaved)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_initialize_blocks_strata", obs, n_obs, true);
    if (s) return s;
    if (!has_strata(h)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no discrete latent to stratify over");
    if ((s = set_strata(h, values, n_strata, interleaved))) return s;
    h->generation += 1;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int grid = step_grid(h);
    s = timed(h, GPF_K_STEP, [&] { DISPATCH_MODEL(h, (launch_init_blk_strata<MM>(h, grid))); });
    if (s) return s;
    h->pending_gather = false; h->pending_fill = false; h->pending_packed = false; h->pending_search = false; h->pending_move = false;
    h->max_valid = true;
    GPF_LAUNCH(k_iota, dim3(grid), dim3(BLOCK), 0, h->stream, h->anc, h->n);            // parents = 1:N (initialize.jl:43)
    HIP_TRY(h, hipMemsetAsync(&h->sc->lml_est, 0, sizeof(double), h->stream));
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->initialized = true; h->has_prev = false; h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;
    mutated(h);
    return GPF_OK;
}
gpf_status gpf_update_blocks_strata(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size, const double* values, int32_t n_strata, int32_t interleaved)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_update_blocks_strata", obs, n_obs, true);
    if (s) return s;
    if ((s = check_ready(h))) return s;
    if (!has_strata(h)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no discrete latent to stratify over");
    if ((s = set_strata(h, values, n_strata, interleaved))) return s;
    if ((s = materialize(h))) return s;
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    s = timed(h, GPF_K_STEP, [&] {
        if (keep) { DISPATCH_MODEL(h, (launch_step_blk_strata<MM, true>(h, grid))); }
        else      { DISPATCH_MODEL(h, (launch_step_blk_strata<MM, false>(h, grid))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->max_valid = true;
    h->cur ^= 1;
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;
    mutated(h);
    return GPF_OK;
}
// for b in blocks: pf_update!(state[b], new_args, argdiffs, observations[b][, proposal, proposal_args]) -- the per-view updates with DIFFERENT
// proposals per view (test/update.jl:179-189) in one launch: use_proposal[b] != 0 -> block b is extended with the model's native proposal
// (src/update.jl:79-96), else with the default one (src/update.jl:12-25).  One epoch for all blocks, like gpf_update_blocks.
gpf_status gpf_update_blocks_proposal(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size, const int32_t* use_proposal, int32_t proposal)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_update_blocks_proposal");
    if (s) return s;
    if ((s = check_ready(h))) return s;
    if (!use_proposal) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null use_proposal");
    const bool ok = (proposal == GPF_PROPOSAL_LOCALLY_OPTIMAL && h->cfg.model != MODEL_LINE) || (proposal == GPF_PROPOSAL_LINE_FIXED && h->cfg.model == MODEL_LINE);
    bool has = false;
    DISPATCH_MODEL(h, (has = Model<MM>::HAS_PROPOSAL));
    if (!ok || !has) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no such native proposal");
    if ((s = materialize(h))) return s;                          // (no fused gather in the block-wise step)
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if ((s = block_buffers(h, nblocks))) return s;               // (blk_mask doubles as the flag array: no block resample refers to it after this call)
    HIP_TRY(h, hipMemcpyAsync(h->blk_mask, use_proposal, (size_t)nblocks * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                 // (the caller's array may go away)
    h->args.blk_prop = h->blk_mask;
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    s = timed(h, GPF_K_STEP, [&] {
        if (keep) { DISPATCH_MODEL(h, (launch_step_blk_prop<MM, true>(h, grid))); }
        else      { DISPATCH_MODEL(h, (launch_step_blk_prop<MM, false>(h, grid))); }
    });
    h->args.blk_prop = nullptr;
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->max_valid = true;
    h->cur ^= 1;
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;
    mutated(h);
    return GPF_OK;
}
gpf_status gpf_rejuvenate_blocks(gpf_handle h, int32_t method, int32_t n_iters, int32_t only_resampled, uint64_t* n_accepted)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->blk_obs_size < 1) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate_blocks needs gpf_initialize_blocks / gpf_update_blocks first (per-block observations)");
    if ((s = block_step_checks(h, h->blk_obs_size, "gpf_rejuvenate_blocks"))) return s;
    if (method != GPF_REJUVENATE_MOVE && method != GPF_REJUVENATE_REWEIGHT) return fail(h, GPF_ERR_UNKNOWN_METHOD, "Method not recognized.");   // rejuvenate.jl:25
    if (!h->cfg.keep_prev) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate needs keep_prev = 1 (x_{t-1} must travel with the particle)");
    if (n_iters < 0) return fail(h, GPF_ERR_INVALID_ARGUMENT, "n_iters < 0");
    if (only_resampled) {
        const int64_t nblocks = (h->n + h->blk_obs_size - 1) / h->blk_obs_size;
        if (!h->blk_mask || h->blk_last != nblocks) return fail(h, GPF_ERR_STATE, "only_resampled needs a gpf_resample_blocks with the same block size first");
    }
    if ((s = materialize(h))) return s;
    h->args.blk_mask = only_resampled ? h->blk_mask : nullptr;
    const int grid = move_grid(h);
    s = timed(h, GPF_K_MOVE, [&] {
        if (method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_blk<MM, true>(h, grid, n_iters))); }
        else                                   { DISPATCH_MODEL(h, (launch_move_blk<MM, false>(h, grid, n_iters))); }
    });
    h->args.blk_mask = nullptr;
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    h->epoch += 1;
    if (method == GPF_REJUVENATE_REWEIGHT) { h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = true; }
    mutated(h);
    if (n_accepted) {
        // (move-reweight: every particle of a participating block moves; the per-workgroup counts cover both cases)
        if (method == GPF_REJUVENATE_REWEIGHT && !only_resampled) *n_accepted = (uint64_t)h->n * (uint64_t)n_iters;
        else if (method == GPF_REJUVENATE_REWEIGHT) {
            int64_t nres = 0;
            GPF_LAUNCH(k_block_summary, dim3(1), dim3(BLOCK), 0, h->stream, h->blk_mask, h->blk_last, h->blk_words);
            int32_t words[2] = {0, 0};
            HIP_TRY(h, hipMemcpyAsync(words, h->blk_words, sizeof(words), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            nres = (int64_t)(uint32_t)words[1];
            // (all blocks have block_size particles except possibly the last)
            const int64_t bs = h->blk_obs_size, last = h->n - (h->blk_last - 1) * bs;
            int32_t last_word = 0;
            HIP_TRY(h, hipMemcpy(&last_word, h->blk_mask + (h->blk_last - 1), sizeof(int32_t), hipMemcpyDeviceToHost));
            *n_accepted = (uint64_t)((nres - (last_word & 1)) * bs + (last_word & 1) * last) * (uint64_t)n_iters;
        } else {
            GPF_LAUNCH(k_sum_accepts, dim3(1), dim3(BLOCK), 0, h->stream, h->acc_part, grid, reinterpret_cast<unsigned long long*>(&h->sc->n_accept));
            HIP_TRY(h, hipGetLastError());
            if ((s = fetch_scalars(h))) return s;
            *n_accepted = h->h_sc->n_accept;
        }
    }
    return GPF_OK;
}

} // extern "C"

namespace gpfh {
// sum_i w_i f(values[i * stride + col]) by the binary tree of DESIGN.md §3.5 (one workgroup per 2048 terms, then the same tree over
// the partials) into *out_dev (device)
gpf_status weighted_tree_sum(gpf_filter* h, const double* values, int stride, int col, int pw, const double* center, double match, double* out_dev)
{
    const int64_t nb = (h->n + TREE_CHUNK - 1) / TREE_CHUNK;
    const int64_t need = nb + (nb + TREE_CHUNK - 1) / TREE_CHUNK + 1;
    if (h->tree_cap < need) {
        if (h->tree_buf) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->tree_buf); h->tree_buf = nullptr; h->tree_cap = 0; }
        HIP_TRY(h, hipMalloc(&h->tree_buf, (size_t)need * sizeof(double)));
        h->tree_cap = need;
    }
    double *in = h->tree_buf, *out = h->tree_buf + nb;
    GPF_LAUNCH(k_wsum_tree, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, h->lw, &h->sc->raw, h->K, values, stride, col, h->n, pw, center, match, in);
    for (int64_t np = nb; np > 1;) {
        const int64_t g = (np + TREE_CHUNK - 1) / TREE_CHUNK;
        GPF_LAUNCH(k_tree_partials, dim3((unsigned)g), dim3(BLOCK), 0, h->stream, in, np, out);
        np = g; std::swap(in, out);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_dev, in, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return GPF_OK;
}

} // namespace gpfh

extern "C" {

static gpf_status wstat(gpf_handle h, int32_t column, double* out, bool variance)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || column < 0 || column >= h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad column/output");
    if ((s = ensure_raw(h))) return s;
    if ((s = weighted_tree_sum(h, h->rows[h->cur], h->W, column, 1, nullptr, 0.0, h->dscal))) return s;
    if (variance && (s = weighted_tree_sum(h, h->rows[h->cur], h->W, column, 2, h->dscal, 0.0, h->dscal + 1))) return s;
    double tmp[2];
    if ((s = copy_out(h, h->dscal, tmp, sizeof(tmp)))) return s;
    *out = variance ? tmp[1] : tmp[0];
    return GPF_OK;
}
gpf_status gpf_mean(gpf_handle h, int32_t column, double* out) { return wstat(h, column, out, false); }
gpf_status gpf_var(gpf_handle h, int32_t column, double* out) { return wstat(h, column, out, true); }

// =================================================================================== sub-state views (src/view.jl)
gpf_status gpf_view_create(gpf_handle parent, int64_t start, int64_t count, gpf_handle* out)
{
    return gpf_view_create_strided(parent, start, 1, count, out);
}

static gpf_status view_create_impl(gpf_handle parent, int64_t start, int64_t step, int64_t count, const int64_t* index, gpf_handle* out);
gpf_status gpf_view_create_strided(gpf_handle parent, int64_t start, int64_t step, int64_t count, gpf_handle* out)
{
    if (!parent || !out) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (step < 1 || step >= ((int64_t)1 << 31)) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view step must be >= 1");
    return view_create_impl(parent, start, step, count, nullptr, out);
}
// state[idxs] / view(state, idxs) for any vector of DISTINCT indices (src/view.jl:35-48): the strided view's compact-copy mechanism with
// an index array.  index: HOST, 0-based, count entries.
gpf_status gpf_view_create_indexed(gpf_handle parent, const int64_t* index, int64_t count, gpf_handle* out)
{
    if (!parent || !out || !index) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (count < 1) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "empty index vector");
    std::vector<int64_t> sorted(index, index + count);
    std::sort(sorted.begin(), sorted.end());
    if (sorted.front() < 0 || sorted.back() >= parent->n) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view index out of bounds");
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
        return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view indices must be distinct (a particle written through two slots of a view has no defined value)");
    return view_create_impl(parent, index[0], 0, count, index, out);
}
static gpf_status view_create_impl(gpf_handle parent, int64_t start, int64_t step, int64_t count, const int64_t* index, gpf_handle* out)
{
    if (parent->parent) return fail(parent, GPF_ERR_STATE, "views of views are not supported");
    // the trajectory store keeps ONE ancestor map and one set of columns per time step for the whole filter: a sub-state that
    // resamples or advances only its own particles would leave it describing something else -- refuse instead of going stale
    if (parent->hist_on) return fail(parent, GPF_ERR_STATE, "a filter with a trajectory store has no sub-state views");
    if (!index && (start < 0 || count < 1 || start + (count - 1) * step >= parent->n)) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view range out of bounds");
    gpf_filter* v = new gpf_filter();
    v->cfg = parent->cfg;
    v->cfg.n_particles = count; v->cfg.n_global = count;          // a sub-state normalises over its own particles
    v->cfg.gid0 = parent->cfg.gid0 + start;                        // ... but RNG counters keep the global particle id
    v->args = parent->args;
    v->d = parent->d; v->W = parent->W; v->n = count; v->n_cu = parent->n_cu;
    v->stream = parent->stream; v->own_stream = false;
    v->parent = parent; v->view_start = start; v->view_step = step; v->parent_generation = parent->generation;
    v->args.gstride = (int32_t)(step ? step : 1);                  // per-particle RNG counters stay the source's particle ids (index views: ModelArgs::gid_map)
    auto body = [&]() -> gpf_status {
        HIP_TRY(v, hipSetDevice(v->cfg.device));
        // scratch of its own (weight levels, descriptors, partials, scalars); rows / lw / anc alias the parent
        v->ntiles = (v->n + TILE - 1) / TILE;
        v->K = fix_K(count);
        v->logN = log_((double)count);
        const size_t n = (size_t)count;
        if (index) {                                             // the particles' indices in the parent; their ids relative to the first
            std::vector<int32_t> ix((size_t)count), rel((size_t)count);
            for (int64_t i = 0; i < count; ++i) { ix[i] = (int32_t)index[i]; rel[i] = (int32_t)(index[i] - index[0]); }
            HIP_TRY(v, hipMalloc(&v->vidx, n * sizeof(int32_t)));
            HIP_TRY(v, hipMalloc(&v->vgid, n * sizeof(int32_t)));
            HIP_TRY(v, hipMemcpy(v->vidx, ix.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(v, hipMemcpy(v->vgid, rel.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            v->args.gid_map = v->vgid;
        }
        if (step != 1) {
            HIP_TRY(v, hipMalloc(&v->vrows[0], n * (size_t)v->W * sizeof(double)));
            HIP_TRY(v, hipMalloc(&v->vrows[1], n * (size_t)v->W * sizeof(double)));
            HIP_TRY(v, hipMalloc(&v->vlw, n * sizeof(double)));
            HIP_TRY(v, hipMalloc(&v->vanc, n * sizeof(int32_t)));
        }
        HIP_TRY(v, hipMalloc(&v->lws, n * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->lp, n * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->dtmp, n * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->cdf[0], (size_t)v->ntiles * TILE * sizeof(uint64_t)));
        HIP_TRY(v, hipMalloc(&v->t16[0], (size_t)v->ntiles * (TILE / 16) * sizeof(uint64_t)));
        HIP_TRY(v, hipMalloc(&v->t256[0], t256_bytes(v->ntiles)));
        const size_t db = (((size_t)2 * v->ntiles * sizeof(uint64_t)) + 15) & ~(size_t)15;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) {
                HIP_TRY(v, hipMalloc(&v->desc[i][j], db));
                HIP_TRY(v, hipMemsetAsync(v->desc[i][j], 0, db, v->stream));
            }
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(v, hipMalloc(&v->mslots[b], (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long)));
            HIP_TRY(v, hipMemsetAsync(v->mslots[b], 0, (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long), v->stream));
        }
        HIP_TRY(v, hipMalloc(&v->blockQ, (size_t)4 * 8 * v->n_cu * sizeof(uint64_t) + 64));
        HIP_TRY(v, hipMemsetAsync(v->blockQ, 0, (size_t)4 * 8 * v->n_cu * sizeof(uint64_t) + 64, v->stream));
        HIP_TRY(v, hipMalloc(&v->partial, MAX_PARTIALS * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->acc_part, MAX_PARTIALS * sizeof(unsigned long long)));
        HIP_TRY(v, hipMalloc(&v->dscal, 4 * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->sc, sizeof(Scalars)));
        HIP_TRY(v, hipHostMalloc(&v->h_sc, sizeof(Scalars)));
        HIP_TRY(v, hipHostMalloc(&v->h_timeout, sizeof(int32_t)));
        *v->h_timeout = 0;
        v->scan_blocks_per_cu = parent->scan_blocks_per_cu; v->wscan_blocks_per_cu = parent->wscan_blocks_per_cu;
        HIP_TRY(v, hipMemsetAsync(v->sc, 0, sizeof(Scalars), v->stream));
        return GPF_OK;
    };
    gpf_status st = body();
    if (st != GPF_OK) { parent->err = v->err; gpf_destroy(v); return st; }
    parent->views.push_back(v);
    *out = v;
    return GPF_OK;
}

// =================================================================================== resize family (src/resize.jl)
static gpf_status resize_ready(gpf_handle h)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "resizing a sharded filter is not supported");
    if (h->hist_on) return fail(h, GPF_ERR_STATE, "resizing a filter with a trajectory store is not supported");
    if (h->parent) return fail(h, GPF_ERR_STATE, "a sub-state view cannot be resized");
    h->generation += 1;                  // views of this filter become stale
    return materialize(h);
}
// after the particle count changed: unsharded bookkeeping
static void set_count(gpf_filter* h, int64_t n_new)
{
    h->n = n_new; h->cfg.n_particles = n_new; h->cfg.n_global = n_new; h->cfg.gid0 = 0;
    if (h->blk_obs_size != 0) h->blk_obs_size = -1;              // per-block observations do not survive a change of the particle count
    h->blk_last = 0;
    h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; h->max_valid = false; h->pending_gather = false; h->pending_fill = false;
    h->pending_packed = false; h->pending_search = false;
}

gpf_status gpf_n_particles(gpf_handle h, int64_t* out)
{
    if (!h || !out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = h->n;
    return GPF_OK;
}

// pf_optimal_resize! (resize.jl:149-200): keep every particle with c w_i >= 1, resample the rest by systematic
// sampling, in exact fixed point.  n_new <= n_old.
static gpf_status resize_optimal(gpf_handle h, int64_t n_new, int32_t check, int32_t* invalid)
{
    const int64_t n_old = h->n;
    if (n_new < 1 || n_new > n_old) return fail(h, GPF_ERR_INVALID_ARGUMENT, "optimal resize: need 1 <= n_particles <= current count");   // resize.jl:185
    gpf_status s;
    // sort(weights) (resize.jl:204), descending; safe_softmax + logsumexp (resize.jl:152,190) over that order
    if ((s = ensure_residual_buffers(h))) return s;
    const PrioView pv = raw_view(h);
    if ((s = ensure_max(h, pv, true))) return s;
    if ((s = sort_desc(h, pv, n_old))) return s;
    WSum* ws = &h->sc->raw;
    h->raw_valid = false; h->raw_sum_valid = false;
    if ((s = summarize(h, pv, ws, true, h->order, true, false, false, true))) return s;
    HIP_TRY(h, hipMemsetAsync(&h->sc->opt_d, 0xff, sizeof(long long), h->stream));
    GPF_LAUNCH(k_opt_threshold, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, h->cdf[0], ws, n_new, n_old, h->sc);
    GPF_LAUNCH(k_opt_params, dim3(1), dim3(1), 0, h->stream, h->cdf[0], ws, n_new, h->sc);
    // keep flags -> compaction offsets (channel 1); weights of the others -> their CDF (channel 2)
    InOptimal ik{h->lw, ws, h->sc, h->K, 0}, iw{h->lw, ws, h->sc, h->K, 1};
    if ((s = scan_launch_optimal(h, 1, ik, &h->sc->Ctot))) return s;
    if ((s = scan_launch_optimal(h, 2, iw, &h->sc->Rs))) return s;
    if ((s = fetch_scalars(h))) return s;
    const WSum& w = h->h_sc->raw;
    const int64_t n_keep = (int64_t)h->h_sc->Ctot, n_res = n_new - n_keep;
    bool inv = w.flags != 0;
    if ((w.flags & (FLAG_NAN | FLAG_POSINF)) || (check == GPF_CHECK_TRUE && inv)) {
        if (invalid) *invalid = 1;
        return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");                              // resize.jl:153
    }
    if (n_res > 0 && h->h_sc->Rs == 0) {
        // every particle that is not kept has weight 0 at this resolution: uniform among them (safe_softmax, resize.jl:166-168)
        inv = true;
        if (check == GPF_CHECK_TRUE) { if (invalid) *invalid = 1; return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights."); }
        InOptimal iu{h->lw, ws, h->sc, h->K, 2};
        if ((s = scan_launch_optimal(h, 2, iu, &h->sc->Rs))) return s;
    }
    if (invalid) *invalid = inv ? 1 : 0;
    const CdfLevels lv = levels(h, 2);
    const uint64_t* keepcdf = h->cdf[1];
    const int64_t ntiles_old = h->ntiles;
    const int K = h->K;
    Bufs old = take_particle_buffers(h);
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    GPF_LAUNCH(k_opt_keep_scatter, dim3(grid_for(h, n_old, 8)), dim3(BLOCK), 0, h->stream, keepcdf, n_old, h->anc);
    if (n_res > 0) {
        SearchArgs sa{};
        sa.w = lv; sa.c = lv; sa.ntiles = ntiles_old; sa.order = nullptr; sa.sc = h->sc; sa.ws = ws; sa.raw = ws;
        sa.n = n_res; sa.n_cells = n_old; sa.n_global = n_res; sa.gid0 = 0; sa.seed = h->cfg.seed; sa.epoch = h->epoch;
        sa.K = K; sa.logN = 0.0; sa.update_lml = 0; sa.anc = h->anc + n_keep;
        const size_t lds = search_lds_bytes(ntiles_old, 1);
        const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((n_res + 2 * SBLOCK - 1) / (2 * SBLOCK), h->n_cu));
        launch_search_plain(h, 3, gsr, lds, sa);
    }
    // new_traces .= view(traces, parents); log_weights (resize.jl:189-197)
    launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], h->lw, n_new);
    const double ratio = log_((double)n_new) - log_((double)n_old);
    GPF_LAUNCH(k_opt_weights, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, h->lw, n_new, h->sc, ws, K, ratio);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    return GPF_OK;
}

gpf_status gpf_resize(gpf_handle h, int64_t n_new, int32_t method, double priority_alpha, int32_t check, int32_t* invalid)
{
    gpf_status s = resize_ready(h);
    if (s) return s;
    if (method == GPF_RESAMPLE_OPTIMAL) return resize_optimal(h, n_new, check, invalid);          // resize.jl:22-23
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");             // resize.jl:26
    if (n_new < 1 || n_new >= ((int64_t)1 << 31)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad n_particles");
    const int64_t n_old = h->n;
    PrioView pv = raw_view(h);
    if (priority_alpha == priority_alpha) { pv.alpha = priority_alpha; pv.mode = 1; }
    // fixed-point scale for max(n_old, n_new): both N_old 2^K and n_new 2^K must stay below 2^62
    h->K = fix_K(std::max(n_old, n_new));
    h->raw_valid = false; h->raw_sum_valid = false;
    WSum* ws = &h->sc->raw;
    if ((s = summarize(h, raw_view(h), &h->sc->raw, true, nullptr, true))) return s;          // logsumexp(log_weights), resize.jl:58
    if (pv.mode != 0) { ws = &h->sc->prio; if ((s = summarize(h, pv, ws, true, nullptr, false))) return s; }
    if (check == GPF_CHECK_TRUE || invalid) {
        if ((s = fetch_scalars(h))) return s;
        const WSum& w = pv.mode == 0 ? h->h_sc->raw : h->h_sc->prio;
        if (invalid) *invalid = w.flags != 0;
        if ((w.flags & (FLAG_NAN | FLAG_POSINF)) || (check == GPF_CHECK_TRUE && w.flags)) {
            h->K = fix_K(n_old);
            return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");                          // resize.jl:56,97
        }
    }
    SearchArgs sa{};
    sa.w = levels(h, 0); sa.c = levels(h, 0); sa.ntiles = h->ntiles; sa.order = nullptr; sa.sc = h->sc; sa.ws = ws;
    sa.raw = &h->sc->raw; sa.n = n_new; sa.n_cells = n_old; sa.n_global = n_new; sa.gid0 = 0; sa.seed = h->cfg.seed;
    sa.epoch = h->epoch; sa.K = h->K; sa.logN = log_((double)n_old); sa.update_lml = 1;
    if (method == GPF_RESAMPLE_RESIDUAL) {
        if ((s = residual_scans(h, ws, n_new))) return s;                                        // floor(n_particles * w), resize.jl:106
        sa.w = levels(h, 2); sa.c = levels(h, 1);
    }
    const int64_t ntiles_old = h->ntiles;
    const int Kp = h->K;
    Bufs old = take_particle_buffers(h);                                                          // resize!(...), resize.jl:60-61
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    sa.anc = h->anc;
    const int64_t nt = method == GPF_RESAMPLE_RESIDUAL ? 2 : 1;
    const size_t lds = search_lds_bytes(ntiles_old, (int)nt);
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((n_new + 2 * SBLOCK - 1) / (2 * SBLOCK), h->n_cu));
    if (method == GPF_RESAMPLE_RESIDUAL) launch_search_plain(h, 1, gsr, lds, sa);
    else                                 launch_multinomial_search(h, sa);
    // new_traces .= view(traces, parents) + update_weights!(state, n_particles, log_priorities)   resize.jl:64-66,424-438
    launch_gather_ex(h, h->anc, old.rows[old.cur], h->rows[0], pv, pv.mode == 0 ? h->lw : h->lws, n_new);
    if (pv.mode != 0) {
        PrioView post{h->lws, nullptr, 0.0, 0};
        if ((s = summarize(h, post, &h->sc->post, false, nullptr, false))) { free_bufs(old); return s; }
        GPF_LAUNCH(k_apply_post, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, h->sc, h->K, h->logN, h->lws, h->lw, n_new);
        h->max_valid = false;
    }
    (void)Kp;
    HIP_TRY(h, hipStreamSynchronize(h->stream));                 // the old buffers are read by the kernels above
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    return GPF_OK;
}

gpf_status gpf_replicate(gpf_handle h, int32_t n_replicates, int32_t interleaved)
{
    gpf_status s = resize_ready(h);
    if (s) return s;
    if (n_replicates < 1 || h->n * (int64_t)n_replicates >= ((int64_t)1 << 31)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad n_replicates");
    const int64_t n_old = h->n, n_new = n_old * n_replicates;
    Bufs old = take_particle_buffers(h);
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    GPF_LAUNCH(k_replicate_anc, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, n_new, n_old, (int)n_replicates,
                       (int)(interleaved != 0), 0, h->anc);
    launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], h->lw, n_new);     // resize.jl:240-242
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

gpf_status gpf_dereplicate(gpf_handle h, int32_t n_replicates, int32_t interleaved, int32_t sample)
{
    gpf_status s = resize_ready(h);
    if (s) return s;
    if (n_replicates < 1 || h->n % n_replicates != 0)
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "the number of particles must be a multiple of n_replicates");   // resize.jl:270
    const int64_t n_old = h->n, n_new = n_old / n_replicates;
    Bufs old = take_particle_buffers(h);
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    const int grid = grid_for(h, n_new, 8);
    if (sample) {                                                                                // resize.jl:281-293
        GPF_LAUNCH(k_dereplicate_sample, dim3(grid), dim3(BLOCK), 0, h->stream, old.lw, n_new, n_old, (int)n_replicates,
                           (int)(interleaved != 0), h->cfg.seed, h->epoch, fix_K(n_replicates), log_((double)n_replicates), h->anc, h->lw);
        launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], nullptr, n_new);
        h->epoch += 1;
    } else {                                                                                     // :keepfirst, resize.jl:274-279
        GPF_LAUNCH(k_replicate_anc, dim3(grid), dim3(BLOCK), 0, h->stream, n_new, n_old, (int)n_replicates,
                           (int)(interleaved != 0), 1, h->anc);
        launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], h->lw, n_new);
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// =================================================================================== trajectory store
gpf_status gpf_history_enable(gpf_handle h, int32_t max_steps)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (max_steps < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "max_steps < 1");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "the trajectory store is not available for sharded filters");
    if (h->initialized) return fail(h, GPF_ERR_STATE, "enable the trajectory store before gpf_initialize");
    h->hist_on = true; h->hist_cap = max_steps;
    if (h->hist_dev_maps) (void)hipFree(h->hist_dev_maps);
    HIP_TRY(h, hipMalloc(&h->hist_dev_maps, (size_t)max_steps * sizeof(int32_t*)));
    return GPF_OK;
}

gpf_status gpf_history_steps(gpf_handle h, int32_t* n_steps)
{
    if (!h || !n_steps) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *n_steps = h->hist_on ? (int32_t)h->hist_x.size() : 0;
    return GPF_OK;
}

// column `column` of time step `step` (1-based, like the t of the Julia address t => :name) into h->dtmp
static gpf_status history_values(gpf_handle h, int32_t step, int32_t column)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!h->hist_on) return fail(h, GPF_ERR_STATE, "trajectory store not enabled (gpf_history_enable)");
    const int T = (int)h->hist_x.size();
    if (step < 1 || step > T || column < 0 || column >= h->d) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad step/column");
    if ((s = hist_snapshot(h))) return s;                         // the current step, in its current order
    // maps of steps T, T-1, ..., step+1 (0-based indices T-1 ... step), applied in that order
    std::vector<const int32_t*> maps;
    for (int q = T - 1; q >= step; --q) maps.push_back(h->hist_map[q]);
    if (!maps.empty())
        HIP_TRY(h, hipMemcpyAsync(h->hist_dev_maps, maps.data(), maps.size() * sizeof(int32_t*), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                   // `maps` is a host temporary
    GPF_LAUNCH(k_hist_column, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->hist_dev_maps, (int)maps.size(),
                       h->hist_x[step - 1], h->d, (int)column, h->n, h->dtmp);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

gpf_status gpf_history_column(gpf_handle h, int32_t step, int32_t column, double* out, int64_t n)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    gpf_status s = history_values(h, step, column);
    if (s) return s;
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(double));
}

static gpf_status history_stat(gpf_handle h, int32_t step, int32_t column, double* out, bool variance)
{
    if (!h || !out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    gpf_status s = history_values(h, step, column);
    if (s) return s;
    if ((s = ensure_raw(h))) return s;
    if ((s = weighted_tree_sum(h, h->dtmp, 1, 0, 1, nullptr, 0.0, h->dscal))) return s;
    if (variance && (s = weighted_tree_sum(h, h->dtmp, 1, 0, 2, h->dscal, 0.0, h->dscal + 1))) return s;
    double tmp[2];
    if ((s = copy_out(h, h->dscal, tmp, sizeof(tmp)))) return s;
    *out = variance ? tmp[1] : tmp[0];
    return GPF_OK;
}
// proportionmap(state, addr)[value] (statistics.jl:91-101): normalised weight of the particles whose column equals `value`;
// step = 0: the current step's column, step >= 1: a past choice along the ancestry (trajectory store)
gpf_status gpf_proportion(gpf_handle h, int32_t step, int32_t column, double value, double* out)
{
    if (!h || !out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    gpf_status s;
    if (step > 0) { if ((s = history_values(h, step, column))) return s; }
    else {
        if ((s = check_ready(h))) return s;
        if (column < 0 || column >= h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad column");
        if ((s = materialize(h))) return s;
        GPF_LAUNCH(k_extract_column, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->rows[h->cur], h->W, column, h->n, h->dtmp);
    }
    if ((s = ensure_raw(h))) return s;
    if ((s = weighted_tree_sum(h, h->dtmp, 1, 0, 3, nullptr, value, h->dscal))) return s;
    return copy_out(h, h->dscal, out, sizeof(double));
}
// ---- checkpoint / resume: everything a filter needs to continue bit for bit -- the population, its log-weights and parents, the log-ML
// estimate, the RNG epoch, the latest observation and strata -- as ONE host blob; loads into a handle created with the same gpf_config
namespace {
struct CkptHeader {
    uint64_t magic; int32_t version, model, keep_prev, W, n_params, has_prev, n_strata, interleaved;
    int64_t n, n_global, gid0; uint64_t seed; uint32_t epoch, pad;
    double lml_est, logK;
    double params[MAX_PARAMS], obs[MAX_OBS], strata[MAX_STRATA], q[4];
};
constexpr uint64_t CKPT_MAGIC = 0x4750465f434b5054ull;          // "GPF_CKPT"
int64_t ckpt_bytes(const gpf_filter* h) { return (int64_t)sizeof(CkptHeader) + h->n * h->W * 8 + h->n * 8 + ((h->n * 4 + 7) & ~(int64_t)7); }
gpf_status ckpt_ready(gpf_filter* h)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (h->parent) return fail(h, GPF_ERR_STATE, "checkpoints are taken of whole filters, not of sub-state views");
    return GPF_OK;
}
} // namespace
gpf_status gpf_checkpoint_size(gpf_handle h, int64_t* bytes)
{
    gpf_status s = ckpt_ready(h);
    if (s) return s;
    if (!bytes) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null output");
    *bytes = ckpt_bytes(h);
    return GPF_OK;
}
gpf_status gpf_checkpoint_save(gpf_handle h, void* out, int64_t bytes)
{
    gpf_status s = ckpt_ready(h);
    if (s) return s;
    if (!out || bytes != ckpt_bytes(h)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "the buffer must hold exactly gpf_checkpoint_size bytes");
    // whatever is still deferred (lazy move, lazy search, un-gathered or un-scattered resample) becomes state first
    if ((s = check_ready(h)) || (s = finish_search(h)) || (s = materialize(h))) return s;
    CkptHeader hd{};
    hd.magic = CKPT_MAGIC; hd.version = 1; hd.model = h->cfg.model; hd.keep_prev = h->cfg.keep_prev; hd.W = h->W; hd.n_params = h->cfg.n_params;
    hd.has_prev = h->has_prev ? 1 : 0; hd.n_strata = h->args.n_strata; hd.interleaved = h->args.interleaved;
    hd.n = h->n; hd.n_global = h->cfg.n_global; hd.gid0 = h->cfg.gid0; hd.seed = h->cfg.seed; hd.epoch = h->epoch;
    hd.logK = h->args.logK;
    for (int i = 0; i < MAX_PARAMS; ++i) hd.params[i] = h->args.P[i];
    for (int i = 0; i < MAX_OBS; ++i) hd.obs[i] = h->args.obs[i];
    for (int i = 0; i < MAX_STRATA; ++i) hd.strata[i] = h->args.strata[i];
    for (int i = 0; i < 4; ++i) hd.q[i] = h->args.q[i];
    char* o = static_cast<char*>(out) + sizeof(CkptHeader);
    const size_t rb = (size_t)h->n * h->W * 8, wb = (size_t)h->n * 8, ab = (size_t)h->n * 4;
    HIP_TRY(h, hipMemcpyAsync(&hd.lml_est, reinterpret_cast<const char*>(h->sc) + offsetof(Scalars, lml_est), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(o, h->rows[h->cur], rb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(o + rb, h->lw, wb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(o + rb + wb, h->anc, ab, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if ((s = check_scan_timeout(h))) return s;
    memcpy(out, &hd, sizeof(hd));
    return GPF_OK;
}
gpf_status gpf_checkpoint_load(gpf_handle h, const void* in, int64_t bytes)
{
    gpf_status s = ckpt_ready(h);
    if (s) return s;
    if (!in || bytes < (int64_t)sizeof(CkptHeader)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "not a checkpoint");
    CkptHeader hd;
    memcpy(&hd, in, sizeof(hd));
    if (hd.magic != CKPT_MAGIC || hd.version != 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "not a checkpoint of this library version");
    bool same = hd.model == h->cfg.model && hd.keep_prev == h->cfg.keep_prev && hd.W == h->W && hd.n == h->n && hd.n_global == h->cfg.n_global &&
                hd.gid0 == h->cfg.gid0 && hd.seed == h->cfg.seed && hd.n_params == h->cfg.n_params;
    for (int i = 0; same && i < hd.n_params; ++i) same = memcmp(&hd.params[i], &h->args.P[i], sizeof(double)) == 0;
    if (!same) return fail(h, GPF_ERR_INVALID_ARGUMENT, "the checkpoint was taken of a filter with another gpf_config (model, parameters, particle counts, gid0, seed, keep_prev)");
    if (bytes != ckpt_bytes(h)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "truncated checkpoint");
    if (h->hist_on) return fail(h, GPF_ERR_STATE, "a filter with a trajectory store cannot load a checkpoint (the store is not part of it)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (gpf_filter* v : h->blk_views) gpf_destroy(v);
    h->blk_views.clear();
    // the state is replaced: nothing deferred survives
    h->pending_move = false; h->pending_search = false; h->pending_gather = false; h->pending_fill = false;
    h->pending_packed = false; h->pend_ring = false;
    const char* o = static_cast<const char*>(in) + sizeof(CkptHeader);
    const size_t rb = (size_t)h->n * h->W * 8, wb = (size_t)h->n * 8, ab = (size_t)h->n * 4;
    HIP_TRY(h, hipMemcpyAsync(h->rows[h->cur], o, rb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->lw, o + rb, wb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->anc, o + rb + wb, ab, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(reinterpret_cast<char*>(h->sc) + offsetof(Scalars, lml_est), &hd.lml_est, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->epoch = hd.epoch; h->has_prev = hd.has_prev != 0; h->initialized = true;
    h->args.n_strata = hd.n_strata; h->args.interleaved = hd.interleaved; h->args.logK = hd.logK;
    for (int i = 0; i < MAX_OBS; ++i) h->args.obs[i] = hd.obs[i];
    for (int i = 0; i < MAX_STRATA; ++i) h->args.strata[i] = hd.strata[i];
    for (int i = 0; i < 4; ++i) h->args.q[i] = hd.q[i];
    h->blk_obs_size = 0;
    h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = false; h->raw_has_q = false; h->raw_q_folded = false;
    h->residual_scanned = false; h->push_counted = false; h->gsum_ok = false; h->ch0_offsets = false;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_history_mean(gpf_handle h, int32_t step, int32_t column, double* out) { return history_stat(h, step, column, out, false); }
gpf_status gpf_history_var(gpf_handle h, int32_t step, int32_t column, double* out) { return history_stat(h, step, column, out, true); }


} // extern "C"
