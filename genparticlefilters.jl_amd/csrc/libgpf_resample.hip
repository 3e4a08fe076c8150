// libgpf_resample.hip -- weight summaries (maximum, fixed-point scan, reductions), sorting, the ancestor searches and pf_resample!
// (src/resample.jl:19-218), the getters that need a summary (ESS, log-ML estimate, normalised weights), sample_unweighted_traces.
#include "gpf_host.hpp"

using namespace gpf;
using namespace gpfh;

namespace gpfh {

// one scan launch on descriptor channel `ch` (0 weights, 1 residual counts, 2 residual weights)
template <class In, int FIXQ>
gpf_status scan_launch(gpf_filter* h, int ch, const In& in, int np, WSum* slot, bool want_cdf, uint64_t* total_out,
                       const double* mf_all = nullptr, ScanExtras ex = ScanExtras{nullptr, nullptr, 0, 0})
{
    uint64_t* dc = h->desc[ch][h->dcur[ch]];
    uint64_t* dn = h->desc[ch][1 - h->dcur[ch]];
    int gs = std::is_same<In, InFixQ>::value ? wscan_grid(h) : scan_grid(h);
    // a sorted multinomial resample is waiting for its tile totals: they ride in this launch as extra workgroups behind the scan's own
    if (h->sp_job_set && std::is_same<In, InFixQ>::value) { ex.sp = h->sp_job; ex.sp.blocks = (int)((h->sp_job.ntl + SCAN_BLOCK - 1) / SCAN_BLOCK); h->sp_job_set = false; }
    const int g_launch = gs + ex.sp.blocks;
    const bool offsets = ch == 0 && h->want_offsets && want_cdf;
    const ScanOut so = scan_out(want_cdf ? h->cdf[ch] : nullptr, h->t16[ch], h->t256[ch], h->ntiles, offsets);
    if (ch == 0 && want_cdf) h->ch0_offsets = offsets && so.off16 != nullptr;
    const ChainScope chain(h);                                   // (several filters on this device: their chained kernels take turns, gpf_host.hpp)
    gpf_status s = timed(h, GPF_K_SCAN, [&] {
        GPF_LAUNCH((k_scan<In, FIXQ>), dim3(g_launch), dim3(SCAN_BLOCK), 0, h->stream, in, h->n, h->ntiles, mf_all, h->mslots[h->mcur], np, slot,
                           so, dc, dn, total_out, h->blockQ, h->h_timeout, ex);
    });
    if (s) return s;
    h->table[ch] = dc + h->ntiles;
    h->dcur[ch] ^= 1;
    return GPF_OK;
}

gpf_status scan_launch_shard(gpf_filter* h, int mode, const InFixQ& in, int np, WSum* slot, bool want_cdf, uint64_t* total_out, const double* mf_all, const ScanExtras& ex)
{
    return mode == 4 ? scan_launch<InFixQ, 4>(h, 0, in, np, slot, want_cdf, total_out, mf_all, ex) : scan_launch<InFixQ, 3>(h, 0, in, np, slot, want_cdf, total_out, mf_all, ex);
}
gpf_status scan_launch_optimal(gpf_filter* h, int ch, const InOptimal& in, uint64_t* total_out)
{
    return scan_launch<InOptimal, 0>(h, ch, in, 0, nullptr, true, total_out);
}

// what gpf_create asks of this unit's kernels: how many scan workgroups a CU keeps resident (the scans' inter-workgroup protocol relies on
// it) and the dynamic-LDS ceilings of the search kernels
gpf_status resample_device_setup(gpf_filter* h)
{
        {   // resident scan workgroups per CU: the smallest answer over the scan kernels, never more than 2 (what the tile
            // schedule was tuned for), one fewer than the API says when it says more (the API can over-count by one)
            int nb = 2;
            const void* scans[] = {reinterpret_cast<const void*>(&k_scan<InFixQ, 1>), reinterpret_cast<const void*>(&k_scan<InFixQ, 2>),
                                   reinterpret_cast<const void*>(&k_scan<InFixQ, 3>), reinterpret_cast<const void*>(&k_scan<InFixQ, 4>),
                                   reinterpret_cast<const void*>(&k_scan<InOptimal, 0>), reinterpret_cast<const void*>(&k_scan_residual2<false>),
                                   reinterpret_cast<const void*>(&k_scan_residual2<true>)};
            for (const void* f : scans) {
                int q = 0;
                HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, f, SCAN_BLOCK, 0));
                if (q < 1) return fail(h, GPF_ERR_HIP, "a scan kernel cannot be resident on this device");
                nb = std::min(nb, q > 2 ? q - 1 : q);
            }
            h->scan_blocks_per_cu = std::max(1, std::min(nb, 2));
            int wb = 64;
            for (int k = 0; k < 4; ++k) {
                int q = 0;
                HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, scans[k], SCAN_BLOCK, 0));
                wb = std::min(wb, q > 2 ? q - 1 : q);
            }
            // (measured at 2 x 10^6 particles, 977 tiles: 2 per CU = two rounds 23.2 us, 4 per CU = one round 21.3 us; GPF_WSCAN_BLOCKS for A/B)
            static const int wscan_max = getenv("GPF_WSCAN_BLOCKS") ? atoi(getenv("GPF_WSCAN_BLOCKS")) : 4;
            h->wscan_blocks_per_cu = std::max(1, std::min(std::min(wb, wscan_max), 8));
        }
        // k_search keeps up to LDS_TILE_TABLE top-level entries (64 KiB) + 32 KiB of cooperation strips in LDS
        const int max_dyn = (int)((lds_pad(LDS_TILE_TABLE) + 4) * sizeof(uint64_t));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<0>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<1>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<3>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_multi<0>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_multi<1>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_multi_s<2>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET));
    return GPF_OK;
}

// the maximum slots (MaxSlots) describe pv: the slots left by the kernel that produced the log-weights when pv is the raw weights and
// they are current (use_producer_max), else one k_max_partial pass over pv
gpf_status ensure_max(gpf_filter* h, const PrioView& pv, bool use_producer_max)
{
    if (use_producer_max && h->max_valid) return GPF_OK;
    const int gp = (int)std::min<int64_t>(MAX_PARTIALS, (h->n + BLOCK - 1) / BLOCK);
    gpf_status s = timed(h, GPF_K_MAX, [&] {
        GPF_LAUNCH(k_max_partial, dim3(gp), dim3(BLOCK), 0, h->stream, pv, h->n, next_slots(h));
    });
    if (s) return s;
    h->max_valid = use_producer_max;           // (otherwise the slots describe pv, which may not be the raw log-weights)
    return GPF_OK;
}
// pv: the weights to summarise; max_ready: ensure_max(pv) has run already (the sorted resample needs the maximum for its sort keys)
gpf_status summarize(gpf_filter* h, const PrioView& pv, WSum* slot, bool want_cdf, const int32_t* order, bool use_producer_max,
                     bool want_q, bool publish_flags, bool max_ready)
{
    ScanExtras ex{nullptr, nullptr, 0, h->cfg.n_global};
    if (publish_flags) {
        if (!h->h_flags) { HIP_TRY(h, hipHostMalloc(&h->h_flags, 2 * sizeof(int64_t))); h->h_flags[0] = h->h_flags[1] = 0; }
        h->flag_ticket += 1;
        ex.host_flags = h->h_flags; ex.ticket = h->flag_ticket;
    }
    const int np = 0;               // (only the sharded scans fold gathered pairs)
    gpf_status s;
    h->q_published = false;
    static const bool q_publish_off = getenv("GPF_ESS_PUBLISH") && !strcmp(getenv("GPF_ESS_PUBLISH"), "kernel");   // (A/B: the separate publish launch)
    // (tag << 48 | limb sum: only while a workgroup folds <= Q_TAG_MAX_TILES tiles -- beyond, e.g. N > 2^26 at 4 x 256 workgroups, the
    // untagged partials + k_publish_scalars)
    const int64_t tiles_per_wg = (h->ntiles + wscan_grid(h) - 1) / wscan_grid(h);
    if (want_q && slot == &h->sc->raw && !q_publish_off && tiles_per_wg <= Q_TAG_MAX_TILES) {
        // the ESS getter's scan: the workgroup of its last tile folds sum q^2 and publishes {flags, S, limbs} to pinned memory itself
        if (!h->h_qpub) { HIP_TRY(h, hipHostMalloc(&h->h_qpub, 8 * sizeof(int64_t))); for (int i = 0; i < 8; ++i) h->h_qpub[i] = 0; }
        h->q_ticket += 1;
        ex.q_host = h->h_qpub; ex.q_ticket = h->q_ticket;
        h->q_published = true;
    }
    if (!max_ready && (s = ensure_max(h, pv, use_producer_max))) return s;
    InFixQ in{pv, order, order ? h->keys : nullptr, h->K, 0.0, 0};     // (after sort_desc the sorted keys are in h->keys)
    if (want_q) s = scan_launch<InFixQ, 2>(h, 0, in, np, slot, want_cdf, &slot->S, nullptr, ex);
    else        s = scan_launch<InFixQ, 1>(h, 0, in, np, slot, want_cdf, &slot->S, nullptr, ex);
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// want_q: also accumulate sum q^2 (only the ESS needs it)
gpf_status ensure_raw(gpf_filter* h, bool want_q)
{
    gpf_status s = materialize(h);
    if (s) return s;
    if (h->raw_valid && (!want_q || h->raw_has_q)) return GPF_OK;
    h->want_offsets = h->offsets_hint;                           // a resample that follows may reuse this CDF
    s = summarize(h, raw_view(h), &h->sc->raw, true, nullptr, true, want_q);
    h->want_offsets = true;
    if (s) return s;
    h->raw_valid = true;
    h->raw_has_q = want_q;
    h->raw_q_folded = h->q_published;                            // (a publishing scan folds the limbs itself; its partials are tagged words)
    return GPF_OK;
}

gpf_status wait_ticket(gpf_filter* h, volatile int64_t* tk, int64_t want, const char* what);
gpf_status check_scan_timeout(gpf_filter* h);
// {flags, S, limbs} as published by the ESS scan or by k_sum_reduce (8 words, unordered on their way to pinned memory: re-read until the
// check word -- ticket ^ payload -- agrees)
gpf_status read_published_summary(gpf_filter* h, WSum& w)
{
    gpf_status s;
    if ((s = wait_ticket(h, h->h_qpub + 6, h->q_ticket, "weight summary"))) return s;
    if ((s = check_scan_timeout(h))) return s;
    for (uint64_t spins = 0;; ++spins) {
        int64_t v[8];
        for (int k = 0; k < 8; ++k) v[k] = __atomic_load_n(h->h_qpub + k, __ATOMIC_ACQUIRE);
        uint64_t chk = (uint64_t)v[6];
        for (int k = 0; k < 6; ++k) chk ^= (uint64_t)v[k];
        if (v[6] == h->q_ticket && chk == (uint64_t)v[7]) {
            w.flags = (int32_t)v[0]; w.S = (uint64_t)v[1];
            for (int k = 0; k < 4; ++k) w.Ql[k] = (uint64_t)v[2 + k];
            return GPF_OK;
        }
        cpu_relax();
        if (spins > (1ull << 26)) return fail(h, GPF_ERR_HIP, "weight summary: the published words never became consistent");
    }
}
// The summary of the raw log-weights WITHOUT the CDF (the ESS and log-ML getters): reuses a valid scan, else ONE reduction launch
// (k_sum_reduce) instead of the scan -- no inter-workgroup chain, no 10 MB of CDF and levels.  false in *done: the filter is too large
// for the tagged partials (a workgroup would fold more than Q_TAG_MAX_TILES tiles): the caller takes the scan.
gpf_status ensure_raw_summary(gpf_filter* h, bool want_q, bool* done)
{
    *done = false;
    gpf_status s = materialize(h);
    if (s) return s;
    static const bool off = getenv("GPF_SUM_REDUCE") && !strcmp(getenv("GPF_SUM_REDUCE"), "0");          // (A/B: always the scan)
    if (off || (h->raw_valid && (!want_q || h->raw_has_q))) return GPF_OK;                              // (a scan's summary is there: use it)
    if (h->raw_sum_valid) { *done = true; return GPF_OK; }
    // GPF_SUM_REDUCE=device: the reduction whose workgroup 0 folds the partials on the device (k_sum_reduce) instead of the host (k_sum_host)
    static const bool device_fold = getenv("GPF_SUM_REDUCE") && !strcmp(getenv("GPF_SUM_REDUCE"), "device");
    const int hgrid = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + SH_TILE - 1) / SH_TILE, (int64_t)h->n_cu));
    if (!device_fold && (h->n + hgrid - 1) / hgrid <= (int64_t)Q_TAG_MAX_TILES * TILE) {
        (void)hgrid;
        // every workgroup's partial sums go straight to pinned memory; this thread adds them up
        if ((s = sum_host_launch(h, nullptr))) return s;
        if ((s = sum_host_fold(h, nullptr))) return s;
        *done = true;
        return GPF_OK;
    }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->n_cu * 4));
    if ((h->ntiles + grid - 1) / grid > Q_TAG_MAX_TILES) return GPF_OK;
    if (!h->sum_part) {
        HIP_TRY(h, hipMalloc(&h->sum_part, (size_t)6 * 4 * h->n_cu * sizeof(uint64_t)));
        HIP_TRY(h, hipMemsetAsync(h->sum_part, 0, (size_t)6 * 4 * h->n_cu * sizeof(uint64_t), h->stream));
    }
    if (!h->h_qpub) { HIP_TRY(h, hipHostMalloc(&h->h_qpub, 8 * sizeof(int64_t))); for (int i = 0; i < 8; ++i) h->h_qpub[i] = 0; }
    if ((s = ensure_max(h, raw_view(h), true))) return s;
    h->q_ticket += 1;
    InFixQ in{raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
    s = timed(h, GPF_K_SCAN, [&] {
        GPF_LAUNCH(k_sum_reduce<false>, dim3(grid), dim3(SCAN_BLOCK), 0, h->stream, in, h->n, h->ntiles, h->mslots[h->mcur], &h->sc->raw, h->sum_part, h->h_qpub,
                   h->q_ticket, h->h_timeout, ShardSum{});
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    WSum w{};
    if ((s = read_published_summary(h, w))) return s;
    h->sum_cache = w;                                            // (m is not published by this kernel: the log-ML getter reads the device block)
    h->sum_on_host = false;
    h->raw_sum_valid = true;
    *done = true;
    return GPF_OK;
}

// k_sum_host: the raw weights' {maximum, flags, S, sum q^2} with the HOST as the folder.  sum_host_launch enqueues it, sum_host_fold waits for
// this launch's lines and leaves the summary in h->sum_cache (sum_on_host, raw_sum_valid).  thr != nullptr (gpf_step_ess): the gated
// form -- the last workgroup also folds on the device and leaves the verdict ESS < *thr in sc->gate_go for the launch behind it; the fold
// then returns that verdict too (after cross-checking it against the host's own).
bool sum_host_ok(const gpf_filter* h)
{
    static const bool off = getenv("GPF_SUM_REDUCE") && (!strcmp(getenv("GPF_SUM_REDUCE"), "0") || !strcmp(getenv("GPF_SUM_REDUCE"), "device"));
    const int hgrid = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + SH_TILE - 1) / SH_TILE, (int64_t)h->n_cu));
    return !off && (h->n + hgrid - 1) / hgrid <= (int64_t)Q_TAG_MAX_TILES * TILE && true;
}
// two sets of accumulator lines, used in turn: every gated reduction (k_sum_host<GATE>, k_sum_shard) clears the other set; behind them the
// arrival counter of k_sum_shard
gpf_status ensure_gate_buffers(gpf_filter* h)
{
    if (h->gate_part) return GPF_OK;
    HIP_TRY(h, hipMalloc(&h->gate_part, (size_t)(2 * GATE_WORDS + 8) * sizeof(uint64_t)));
    HIP_TRY(h, hipMemsetAsync(h->gate_part, 0, (size_t)(2 * GATE_WORDS + 8) * sizeof(uint64_t), h->stream));
    HIP_TRY(h, hipHostMalloc(&h->h_gate, sizeof(int64_t)));
    *h->h_gate = 0;
    h->gate_cur = 0;
    return GPF_OK;
}
gpf_status sum_host_launch(gpf_filter* h, const double* thr)
{
    gpf_status s;
    const int hgrid = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + SH_TILE - 1) / SH_TILE, (int64_t)h->n_cu));
    if (!h->h_spart) {
        HIP_TRY(h, hipHostMalloc(&h->h_spart, (size_t)8 * h->n_cu * sizeof(int64_t)));
        memset(h->h_spart, 0, (size_t)8 * h->n_cu * sizeof(int64_t));
    }
    if (thr && (s = ensure_gate_buffers(h))) return s;
    if (thr) h->gate_cur ^= 1;
    if ((s = ensure_max(h, raw_view(h), true))) return s;
    h->q_ticket += 1;
    InFixQ in{raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
    s = timed(h, GPF_K_SCAN, [&] {
        if (thr) GPF_LAUNCH(k_sum_host<true>, dim3(hgrid), dim3(SH_BLOCK), 0, h->stream, in, h->n, h->mslots[h->mcur], h->h_spart, h->q_ticket, SumGate{h->gate_part + h->gate_cur * GATE_WORDS, h->gate_part + (1 - h->gate_cur) * GATE_WORDS});
        else     GPF_LAUNCH(k_sum_host<false>, dim3(hgrid), dim3(SH_BLOCK), 0, h->stream, in, h->n, h->mslots[h->mcur], h->h_spart, h->q_ticket, SumGate{nullptr, nullptr});
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
gpf_status sum_host_fold(gpf_filter* h, const double* thr, int* go_out)
{
    const int hgrid = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + SH_TILE - 1) / SH_TILE, (int64_t)h->n_cu));
    const uint64_t tag = (uint64_t)((h->q_ticket & 0x7fff) + 1);
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t flags_m[3] = {0, 0, 0};
    for (int b = 0; b < hgrid; ++b) {
        volatile int64_t* line = h->h_spart + (size_t)b * 8;
        uint64_t v[8];
        for (int k = 0; k < 8; ++k) {
            uint64_t spins = 0;
            while (((v[k] = (uint64_t)__atomic_load_n(line + k, __ATOMIC_ACQUIRE)) >> 48) != tag) {
                cpu_relax();
                if ((++spins & 0x3fff) != 0) continue;
                const hipError_t q = hipStreamQuery(h->stream);
                if (q == hipErrorNotReady) continue;
                if (((uint64_t)__atomic_load_n(line + k, __ATOMIC_ACQUIRE) >> 48) == tag) continue;
                return fail(h, GPF_ERR_HIP, q == hipSuccess ? "weight summary: the stream drained without the partial sums being published" : hipGetErrorString(q));
            }
            v[k] &= 0xffffffffffffull;
        }
        // the consumed line goes back to zero (tag 0 is never valid): the 15-bit tag alone cannot tell this launch's words from those of a
        // launch 32768 tickets earlier (q_ticket is shared with the ESS scan and k_sum_reduce; the grid changes with a resize)
        for (int k = 0; k < 8; ++k) __atomic_store_n(line + k, (int64_t)0, __ATOMIC_RELAXED);
        t[0] += v[0]; t[1] += v[1] & 0xffffffffffull;             // (S can be 2^62 itself: the high part takes 32 bits; the flags sit above bit 40)
        for (int k = 2; k < 6; ++k) t[k] += v[k];
        if (b == 0) { flags_m[0] = v[1] >> 40; flags_m[1] = v[6]; flags_m[2] = v[7]; }
    }
    WSum w{};
    w.flags = (int32_t)flags_m[0];
    w.S = t[0] + (t[1] << 31);
    for (int k = 0; k < 4; ++k) w.Ql[k] = t[2 + k];
    const uint64_t mb = flags_m[1] | (flags_m[2] << 32);
    memcpy(&w.m, &mb, sizeof(double));
    h->sum_cache = w;                                        // (sc->raw on the device is NOT updated: the getters read this copy)
    h->sum_on_host = true;
    h->raw_sum_valid = true;
    if (thr) {
        uint64_t hi, lo;
        normalise_Q(w, hi, lo);
        if (go_out) *go_out = !w.flags && ess_from(w.S, hi, lo) < *thr ? 1 : 0;
    }
    return GPF_OK;
}
// the verdict the speculative propagate acted on (gate_verdict: ticket << 1 | go in pinned memory, published by its first workgroup) against
// the host's: the same sums through the same operations -- a difference means the two sides of the call went different ways
// ticket: the ticket of the reduction launch whose verdict is checked, captured when it was launched (the handle's live q_ticket may have moved on: a
// resample, an update or a getter between the launch and this check may summarise again)
gpf_status sum_gate_check(gpf_filter* h, int host_go, int64_t ticket)
{
    {
        gpf_status s;
        uint64_t spins = 0;
        int64_t gv;
        while (((gv = __atomic_load_n(h->h_gate, __ATOMIC_ACQUIRE)) >> 1) != ticket) {
            cpu_relax();
            if ((++spins & 0x3fff) != 0) continue;
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipErrorNotReady) continue;
            if ((__atomic_load_n(h->h_gate, __ATOMIC_ACQUIRE) >> 1) == ticket) continue;
            return fail(h, GPF_ERR_HIP, q == hipSuccess ? "ESS gate: the stream drained without the verdict being published" : hipGetErrorString(q));
        }
        if ((s = check_scan_timeout(h))) return s;
        if (host_go != (int)(gv & 1)) return fail(h, GPF_ERR_HIP, "ESS gate: the device's verdict differs from the host's (state may be inconsistent)");
    }
    return GPF_OK;
}

// the sharded getters' reduction (k_sum_reduce<SHARD>): launched on behalf of libgpf_shard.hip, which fills the mailbox rounds; false in *ok:
// the filter is too large for the tagged partials (the caller takes the scan + copies)
bool shard_sum_collect() { static const bool collect = getenv("GPF_SHARD_SUM") && !strcmp(getenv("GPF_SHARD_SUM"), "collect"); return collect; }
gpf_status shard_sum_launch(gpf_filter* h, const ShardSum& ss, bool* ok)
{
    *ok = false;
    // GPF_SHARD_SUM=collect: the first form (k_sum_reduce<SHARD>: 256-thread workgroups, workgroup 0 collects tagged partials) -- A/B, tests
    if (!shard_sum_collect()) {
        gpf_status s0 = ensure_gate_buffers(h);
        if (s0) return s0;
        if (!h->h_qpub) { HIP_TRY(h, hipHostMalloc(&h->h_qpub, 8 * sizeof(int64_t))); for (int i = 0; i < 8; ++i) h->h_qpub[i] = 0; }
        h->gate_cur ^= 1;
        h->q_ticket += 1;
        const int hgrid = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + SH_TILE - 1) / SH_TILE, (int64_t)h->n_cu));
        InFixQ in0{raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
        const ShardAcc sa{h->gate_part + h->gate_cur * GATE_WORDS, h->gate_part + (1 - h->gate_cur) * GATE_WORDS, reinterpret_cast<unsigned int*>(h->gate_part + 2 * GATE_WORDS)};
        s0 = timed(h, GPF_K_SCAN, [&] { GPF_LAUNCH(k_sum_shard, dim3(hgrid), dim3(SH_BLOCK), 0, h->stream, in0, h->n, &h->sc->raw, h->h_qpub, h->q_ticket, sa, ss); });
        if (s0) return s0;
        HIP_TRY(h, hipGetLastError());
        *ok = true;
        return GPF_OK;
    }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->n_cu * 4));
    if ((h->ntiles + grid - 1) / grid > Q_TAG_MAX_TILES) return GPF_OK;
    if (!h->sum_part) {
        HIP_TRY(h, hipMalloc(&h->sum_part, (size_t)6 * 4 * h->n_cu * sizeof(uint64_t)));
        HIP_TRY(h, hipMemsetAsync(h->sum_part, 0, (size_t)6 * 4 * h->n_cu * sizeof(uint64_t), h->stream));
    }
    if (!h->h_qpub) { HIP_TRY(h, hipHostMalloc(&h->h_qpub, 8 * sizeof(int64_t))); for (int i = 0; i < 8; ++i) h->h_qpub[i] = 0; }
    h->q_ticket += 1;
    InFixQ in{raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
    gpf_status s = timed(h, GPF_K_SCAN, [&] {
        GPF_LAUNCH(k_sum_reduce<true>, dim3(grid), dim3(SCAN_BLOCK), 0, h->stream, in, h->n, h->ntiles, h->mslots[h->mcur], &h->sc->raw, h->sum_part, h->h_qpub,
                   h->q_ticket, h->h_timeout, ss);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    *ok = true;
    return GPF_OK;
}

// Poll a pinned ticket that a kernel on h->stream publishes.  A failed kernel never writes it: any stream status other than
// "not ready" is terminal (re-read once, then report), so a faulting kernel cannot hang the host -- or, in a multi-rank job,
// its peers in the next collective.
gpf_status wait_ticket(gpf_filter* h, volatile int64_t* tk, int64_t want, const char* what)
{
    uint64_t spins = 0;
    while (__atomic_load_n(tk, __ATOMIC_ACQUIRE) != want) {
        cpu_relax();
        if ((++spins & 0x3fff) != 0) continue;
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipErrorNotReady) continue;
        if (__atomic_load_n(tk, __ATOMIC_ACQUIRE) == want) break;
        if (q == hipSuccess) return fail(h, GPF_ERR_HIP, std::string(what) + ": the stream drained without the ticket being published");
        return fail(h, GPF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
    }
    return GPF_OK;
}
// a scan whose bounded inter-workgroup wait gave up leaves garbage prefixes behind: fail loudly at the next host touch point
gpf_status check_scan_timeout(gpf_filter* h)
{
    if (h->h_timeout && __atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) == 2)
        return fail(h, GPF_ERR_HIP, "sharded resample: a peer's summary did not arrive in its mailbox in time (a rank is down or far behind); results are invalid");
    if (h->h_timeout && __atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) == 3)
        return fail(h, GPF_ERR_HIP, "sharded resample: a peer's rows did not arrive in the receive window in time (a rank is down or far behind); results are invalid");
    if (h->h_timeout && __atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) != 0)
        return fail(h, GPF_ERR_HIP, "scan kernel: bounded inter-workgroup wait timed out (workgroups not co-resident?); results are invalid");
    return GPF_OK;
}

gpf_status fetch_scalars(gpf_filter* h, bool fold_raw_q)
{
    if (!h->h_sc_ticket) { HIP_TRY(h, hipHostMalloc(&h->h_sc_ticket, sizeof(long long))); *h->h_sc_ticket = 0; }
    h->sc_ticket += 1;
    GPF_LAUNCH(k_publish_scalars, dim3(1), dim3(WAVE), 0, h->stream, h->sc, h->h_sc, h->h_sc_ticket, h->sc_ticket,
               fold_raw_q ? h->blockQ : nullptr, fold_raw_q ? wscan_grid(h) : 0);
    HIP_TRY(h, hipGetLastError());
    { gpf_status w = wait_ticket(h, reinterpret_cast<volatile int64_t*>(h->h_sc_ticket), (int64_t)h->sc_ticket, "scalar block"); if (w) return w; }
    return check_scan_timeout(h);
}

gpf_status ensure_sort_buffers(gpf_filter* h)
{
    if (h->order) return GPF_OK;
    const size_t n = (size_t)h->n;
    HIP_TRY(h, hipMalloc(&h->order, n * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&h->idx_in, n * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&h->keys, n * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->keys_out, n * sizeof(uint64_t)));
    // TWO workspaces (histograms, tickets, descriptor planes), used in turn: every sort clears the other one for the next sort
    h->sort_tmp_bytes = (sort_ws_bytes(h->n) + 15) & ~(size_t)15;
    HIP_TRY(h, hipMalloc(&h->sort_tmp, 2 * h->sort_tmp_bytes));
    HIP_TRY(h, hipMemsetAsync(h->sort_tmp, 0, 2 * h->sort_tmp_bytes, h->stream));
    h->sort_ws_cur = 0;
    return GPF_OK;
}

// order = sortperm(log_priorities, rev=true) (resample.jl:156-157) into h->order, the sorted keys into h->keys (gpf_k_sort.hpp K10):
// keys + digit histograms in one pass, then one onesweep kernel per 8-bit digit.
//   coarse = true : three passes over the 24-bit coarse key (sort_coarse: distance from the maximum, which the maximum slots must hold
//                   -- ensure_max); the caller finishes the runs of equal coarse keys (k_sort_finish).  keys -> keys_out -> keys ->
//                   keys_out, payload index -> idx_in -> order -> idx_in: the finish brings both back to h->keys / h->order.
//   coarse = false: all eight passes over the 64-bit key; the eighth leaves keys / permutation in h->keys / h->order.
// K10d in its wide form (512 buckets, 16 384 fine bins): above BK_NARROW_N particles, or everywhere with GPF_SORT=wide (tests at small n)
bool sort_buckets_wide(int64_t n) { static const bool force = getenv("GPF_SORT") && strstr(getenv("GPF_SORT"), "wide"); return force || n > BK_NARROW_N; }
gpf_status sort_passes(gpf_filter* h, const PrioView& pv, int64_t n, bool coarse, uint32_t** ws_used = nullptr, bool buckets = false)
{
    gpf_status s = ensure_sort_buffers(h);
    if (s) return s;
    static_assert(SORT_BINS == BLOCK, "one thread per digit bin");
    // this sort's workspace starts zeroed (by the previous sort, or by the allocation); the key pass zeroes the other one
    char* ws = static_cast<char*>(h->sort_tmp) + (size_t)h->sort_ws_cur * h->sort_tmp_bytes;
    char* other = static_cast<char*>(h->sort_tmp) + (size_t)(1 - h->sort_ws_cur) * h->sort_tmp_bytes;
    h->sort_ws_cur ^= 1;
    if (ws_used) *ws_used = reinterpret_cast<uint32_t*>(ws);
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws);
    uint32_t* const ticket_words = hist + SORT_PASSES * SORT_BINS;
    double* m_ptr = reinterpret_cast<double*>(ticket_words + SORT_M_WORD);
    uint32_t* fine = reinterpret_cast<uint32_t*>(ws + sort_ws_fine_offset());
    uint32_t* bbase = fine + SORT_FINE_MAX;
    uint64_t* desc = reinterpret_cast<uint64_t*>(ws + sort_ws_desc_offset());
    const int64_t nt = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t* const ticket = ticket_words;
    const int64_t clear16 = (int64_t)(h->sort_tmp_bytes / 16);
    // (ONE workgroup per CU: every workgroup ends with up to 256 global atomic adds per sorted digit into the same counters; with 2 / 4
    //  workgroups per CU a four-digit kernel took 16.0 / 24.2 us against 13.3)
    const unsigned long long* slots = h->mslots[h->mcur];
    const ChainScope chain(h);                                   // (k_sort_pass chains its workgroups: gpf_host.hpp ChainGate)
    if (buckets) {
        // K10d: keys + fine-bin histogram, ONE partition pass (keys -> keys_out, payload index -> idx_in); the caller runs k_sort_buckets
        const int64_t kf_grid = std::max<int64_t>(1, std::min<int64_t>((n + 4 * KF_BLOCK - 1) / (4 * KF_BLOCK), h->n_cu));
        if (sort_buckets_wide(n)) {
            GPF_LAUNCH(k_sort_keys_fine<SORT_FINE_BITS_WIDE>, dim3((unsigned)kf_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, fine, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
            GPF_LAUNCH((k_sort_pass<2, SORT_BINS_WIDE, SORT_FINE_BITS_WIDE>), dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, h->keys, nullptr, h->keys_out, h->idx_in, n, 0, hist, ticket, desc, h->h_timeout, m_ptr, fine, bbase);
        } else {
            GPF_LAUNCH(k_sort_keys_fine<SORT_FINE_BITS>, dim3((unsigned)kf_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, fine, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
            GPF_LAUNCH(k_sort_pass<2>, dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, h->keys, nullptr, h->keys_out, h->idx_in, n, 0, hist, ticket, desc, h->h_timeout, m_ptr, fine, bbase);
        }
        HIP_TRY(h, hipGetLastError());
        return GPF_OK;
    }
    const int64_t kh_grid = std::max<int64_t>(1, std::min<int64_t>((n + 4 * KF_BLOCK - 1) / (4 * KF_BLOCK), h->n_cu));
    if (coarse) GPF_LAUNCH((k_sort_keys_hist<0, true>), dim3((unsigned)kh_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, hist, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
    else        GPF_LAUNCH((k_sort_keys_hist<0, false>), dim3((unsigned)kh_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, hist, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
    for (int p = 0; p < (coarse ? 3 : SORT_PASSES); ++p) {
        const uint64_t* kin = (p & 1) ? h->keys_out : h->keys;
        uint64_t* kout = (p & 1) ? h->keys : h->keys_out;
        const int32_t* vin = p == 0 ? nullptr : ((p & 1) ? h->idx_in : h->order);
        int32_t* vout = (p & 1) ? h->order : h->idx_in;
        if (coarse) GPF_LAUNCH(k_sort_pass<1>, dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, kin, vin, kout, vout, n, p, hist, ticket, desc, h->h_timeout, m_ptr, nullptr, nullptr);
        else        GPF_LAUNCH(k_sort_pass<0>, dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, kin, vin, kout, vout, n, p, hist, ticket, desc, h->h_timeout, m_ptr, nullptr, nullptr);
    }
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// three coarse passes + k_sort_finish; all eight passes when a run of equal coarse keys was too long for the finish (the host learns
// it from pinned memory), or with GPF_SORT=radix8 in the environment (A/B measurements; GPF_SORT=fallback: always both, for the tests).
// The maximum slots must describe pv (ensure_max).
//   sort_desc_begin enqueues the sort; *pending = the finish's verdict is still out: the caller may enqueue the work that consumes
//   the order behind it and asks sort_desc_flagged AFTERWARDS (no host wait between the sort and its consumers) -- when that says
//   "flagged" the order was wrong: sort_passes(..., false) and the consumers again.
// GPF_SORT: radix8 = always the eight passes; fallback = the fast path AND the eight passes (tests); coarse3 = the three coarse passes +
// k_sort_finish also where the bucket sort (K10d, n <= BK_MAX_N) would run (A/B measurements, tests of that path at small n)
int sort_mode() { static const int mode = [] { const char* e = getenv("GPF_SORT"); return e && strstr(e, "radix8") ? 1 : (e && strstr(e, "fallback") ? 2 : 0); }(); return mode; }
bool sort_buckets_ok(int64_t n) { static const bool off = getenv("GPF_SORT") && strstr(getenv("GPF_SORT"), "coarse3"); return !off && n <= BK_MAX_N; }
gpf_status sort_desc_begin(gpf_filter* h, const PrioView& pv, int64_t n, bool* pending)
{
    *pending = false;
    if (sort_mode() == 1) return sort_passes(h, pv, n, false);
    const bool buckets = sort_buckets_ok(n);
    uint32_t* ws = nullptr;
    gpf_status s = sort_passes(h, pv, n, true, &ws, buckets);    // (either form leaves keys / payload in h->keys_out / h->idx_in)
    if (s) return s;
    if (!h->h_sort_flag) { HIP_TRY(h, hipHostMalloc(&h->h_sort_flag, 2 * sizeof(int64_t))); h->h_sort_flag[0] = h->h_sort_flag[1] = 0; }
    uint32_t* done = ws + SORT_PASSES * SORT_BINS + 64;                                  // (behind this sort's zeroed tickets)
    const double* m_ptr = reinterpret_cast<const double*>(ws + SORT_PASSES * SORT_BINS + SORT_M_WORD);
    h->sort_ticket += 1;
    if (buckets) {
        const uint32_t* bbase = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ws) + sort_ws_fine_offset()) + SORT_FINE_MAX;
        // every bucket is ordered in place (the dead keys' bucket is left as the partition wrote it): the partition's output buffers
        // become the sorted keys / the permutation
        GPF_LAUNCH(k_sort_buckets, dim3(sort_buckets_wide(n) ? SORT_BINS_WIDE : SORT_BINS), dim3(BK_BLOCK), 0, h->stream, h->keys_out, h->idx_in, n, bbase,
                   done, h->h_sort_flag, h->sort_ticket, m_ptr);
        std::swap(h->keys, h->keys_out);
        std::swap(h->order, h->idx_in);
    } else {
        GPF_LAUNCH(k_sort_finish, dim3((unsigned)((n + FIN_TILE - 1) / FIN_TILE)), dim3(FIN_BLOCK), 0, h->stream, h->keys_out, h->idx_in, h->keys, h->order, n,
                   done, h->h_sort_flag, h->sort_ticket, m_ptr);
    }
    HIP_TRY(h, hipGetLastError());
    *pending = true;
    return GPF_OK;
}
gpf_status sort_desc_flagged(gpf_filter* h, bool* flagged)
{
    // (the finish publishes ticket << 1 | verdict as one word)
    volatile int64_t* tk = h->h_sort_flag + 1;
    uint64_t spins = 0;
    int64_t v;
    while (((v = __atomic_load_n(tk, __ATOMIC_ACQUIRE)) >> 1) != h->sort_ticket) {
        cpu_relax();
        if ((++spins & 0x3fff) != 0) continue;
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipErrorNotReady) continue;
        if ((__atomic_load_n(tk, __ATOMIC_ACQUIRE) >> 1) == h->sort_ticket) continue;
        return fail(h, GPF_ERR_HIP, q == hipSuccess ? "sort finish: the stream drained without the ticket being published" : hipGetErrorString(q));
    }
    *flagged = (v & 1) != 0 || sort_mode() == 2;
    return GPF_OK;
}
gpf_status sort_desc(gpf_filter* h, const PrioView& pv, int64_t n)
{
    bool pending = false, flagged = false;
    gpf_status s = sort_desc_begin(h, pv, n, &pending);
    if (s || !pending) return s;
    if ((s = sort_desc_flagged(h, &flagged))) return s;
    return flagged ? sort_passes(h, pv, n, false) : GPF_OK;
}

gpf_status ensure_residual_buffers(gpf_filter* h)
{
    for (int i = 1; i < 3; ++i) {
        if (h->cdf[i]) continue;
        HIP_TRY(h, hipMalloc(&h->cdf[i], (size_t)h->ntiles * TILE * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&h->t16[i], (size_t)h->ntiles * (TILE / 16) * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&h->t256[i], t256_bytes(h->ntiles)));
    }
    return GPF_OK;
}

CdfLevels levels(const gpf_filter* h, int ch)
{
    int logg = ch == 0 && h->ch0_offsets ? multi_logg(h->ntiles) : -1;
    const int sample = ch == 0 && h->ch0_offsets && logg < 0 ? multi_sample(h->ntiles) : 0;
    if (sample > 0) logg = 0;
    return CdfLevels{h->cdf[ch], h->t16[ch], h->t256[ch], h->table[ch], k32_of(h->t256[ch], h->ntiles),
                     logg >= 0 ? off16_of(h->t256[ch], h->ntiles) : nullptr, logg >= 0 ? coarse_of(h->t256[ch], h->ntiles) : nullptr, logg,
                     sample > 0 ? k32s_of(h->t256[ch], h->ntiles) : nullptr, sample};
}

// residual: copy-count and residual-weight CDFs from the weight CDF (resample.jl:99,109); ws->S must be the GLOBAL sum
// head_anc: the plain resample hands over its ancestor array -- the scan writes the deterministic head into it (k_scan_residual2), the
// search then covers the tail only (SearchArgs::head_done) and the copy-count CDF is not stored
gpf_status residual_scans(gpf_filter* h, const WSum* ws, int64_t n_slots_global, int32_t* head_anc, const ResidDirect* direct)
{
    gpf_status s = ensure_residual_buffers(h);
    if (s) return s;
    // both prefix sums in one pass (one read of the weight CDF, one division per element)
    Scan2Chan ch[2];
    for (int c = 0; c < 2; ++c) {
        const int id = 1 + c;
        uint64_t* dc = h->desc[id][h->dcur[id]];
        ch[c].out = scan_out(h->cdf[id], h->t16[id], h->t256[id], h->ntiles, false);
        ch[c].dcur = dc; ch[c].dnext = h->desc[id][1 - h->dcur[id]];
        ch[c].total_out = c == 0 ? &h->sc->Ctot : &h->sc->Rs;
        h->table[id] = dc + h->ntiles;
        h->dcur[id] ^= 1;
    }
    const int gs = scan_grid(h);
    const ChainScope chain(h);
    s = timed(h, GPF_K_SCAN, [&] {
        if (direct) GPF_LAUNCH(k_scan_residual2<true>, dim3(gs), dim3(SCAN_BLOCK), 0, h->stream, h->cdf[0], ws, n_slots_global, h->n, h->ntiles, ch[0], ch[1], h->h_timeout, head_anc, &h->sc->giants, h->epoch & 0xffffffu, *direct);
        else        GPF_LAUNCH(k_scan_residual2<false>, dim3(gs), dim3(SCAN_BLOCK), 0, h->stream, h->cdf[0], ws, n_slots_global, h->n, h->ntiles, ch[0], ch[1], h->h_timeout, head_anc, &h->sc->giants, h->epoch & 0xffffffu, ResidDirect{});
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// ancestors of i.i.d. targets: k_search_multi (4-byte keys of every 32 / 64 cells in LDS) while the key table fits, else k_search<0>
void launch_multinomial_search(gpf_filter* h, const SearchArgs& sa)
{
    const int logg = sa.w.off16 && sa.w.sample == 0 ? sa.w.logg : -1;   // the offset levels exist for channel 0 only
    if (sa.w.off16 && sa.w.sample > 0) {                         // 2.5 M .. 5 M particles: sampled key table
        const int gss = (int)std::max<int64_t>(1, std::min<int64_t>((sa.n + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu));
        const size_t lds = multi_lds_bytes(sa.ntiles, sa.w.sample);
        static_assert(MULTI_SAMPLE_MAX == 2, "one instantiation");
        GPF_LAUNCH((k_search_multi_s<2>), dim3(gss), dim3(SBLOCK), lds, h->stream, sa);
        return;
    }
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((sa.n + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu));   // (k_search_multi strides by its own slots per lane)
    if (logg == 0)      GPF_LAUNCH((k_search_multi<0>), dim3(gsr), dim3(SBLOCK), multi_lds_bytes(sa.ntiles, 0), h->stream, sa);
    else if (logg == 1) GPF_LAUNCH((k_search_multi<1>), dim3(gsr), dim3(SBLOCK), multi_lds_bytes(sa.ntiles, 1), h->stream, sa);
    else                GPF_LAUNCH((k_search<0>), dim3(gsr), dim3(SBLOCK), search_lds_bytes(sa.ntiles, 1), h->stream, sa);
}

void launch_search_plain(gpf_filter* h, int which, int grid, size_t lds, const SearchArgs& sa)
{
    if (which == 1) GPF_LAUNCH((k_search<1>), dim3(grid), dim3(SBLOCK), lds, h->stream, sa);
    else            GPF_LAUNCH((k_search<3>), dim3(grid), dim3(SBLOCK), lds, h->stream, sa);
}
void launch_search_strat(gpf_filter* h, const SearchArgs& sa, int64_t n_slots, bool sorted_uniforms)
{
    if (sorted_uniforms) GPF_LAUNCH((k_search_strat<true>), dim3((unsigned)((n_slots + MJB - 1) / MJB)), dim3(MBLOCK), 0, h->stream, sa);
    else                 GPF_LAUNCH((k_search_strat<false>), dim3((unsigned)((n_slots + MJB_STRAT - 1) / MJB_STRAT)), dim3(MBLOCK), 0, h->stream, sa);
}
// GPF_RESAMPLE_MULTINOMIAL_SORTED: buffers for the tile totals of n_slots slots whose first slot has the RNG id gid0, and the job that draws them
// (left pending: the next weight scan of the call carries it as extra workgroups -- scan_launch --, else sorted_gammas_finish launches it)
gpf_status sorted_job_prepare(gpf_filter* h, int64_t gid0, int64_t n_slots)
{
    const int64_t ntl = (n_slots + SP_TILE - 1) / SP_TILE;
    if (h->sp_cap < ntl + 1) {
        if (h->sp_g) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->sp_g); (void)hipFree(h->sp_vlo); h->sp_g = h->sp_vlo = nullptr; h->sp_cap = 0; }
        HIP_TRY(h, hipMalloc(&h->sp_g, (size_t)(ntl + 1) * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&h->sp_vlo, (size_t)(ntl + 1) * sizeof(uint64_t)));
        h->sp_cap = ntl + 1;
    }
    h->sp_job = SortedGammaJob{h->cfg.seed, h->sp_g, gid0, n_slots, ntl, h->epoch, gamma_E(ntl), 0};
    h->sp_job_set = true;
    return GPF_OK;
}
// the pending tile totals now, if no weight scan has carried them; with_tiles: also where every tile starts (k_sorted_tiles)
gpf_status sorted_gammas_finish(gpf_filter* h, bool with_tiles)
{
    const int64_t ntl = h->sp_job.ntl;
    if (h->sp_job_set) {
        h->sp_job_set = false;
        GPF_LAUNCH(k_sorted_gammas, dim3((unsigned)((ntl + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, h->sp_job);
    }
    if (with_tiles) GPF_LAUNCH(k_sorted_tiles, dim3(1), dim3(STILES_BLOCK), 0, h->stream, h->sp_g, ntl, h->sp_vlo);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// the ancestors of a pending multinomial resample are wanted as an array after all (getters, views, rejuvenation, a second resample,
// an update that is not the plain propagate): the stand-alone search, as the resample itself would have run it
gpf_status finish_search(gpf_filter* h)
{
    if (!h->pending_search) return GPF_OK;
    h->pending_search = false;
    gpf_status s = timed(h, GPF_K_SEARCH, [&] { launch_multinomial_search(h, h->pend_sa); });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// which models / sizes k_step_search covers: the key-table regime of the search (up to 2.5 M particles)
bool lazy_search_ok(const gpf_filter* h)
{
    const bool off = !h->lazy_search;                            // off by default: no faster than the two kernels (gpf_k_fused.hpp), gpf_set_lazy_search
    const int logg = multi_logg(h->ntiles);
    // (the ancestor ring sits behind the key table: both must fit the CU's LDS with the kernel's static words)
    return !off && !h->parent && !h->hist_on && h->cfg.n_global == h->n && logg >= 0 &&
           multi_lds_bytes(h->ntiles, logg) + 16 + FUSED_LDS_EXTRA + 1024 <= (size_t)160 * 1024;
}

gpf_status resample_impl(gpf_filter* h, int method, PrioView pv, int sort_particles, int check, int32_t* invalid, bool local)
{
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED && method != GPF_RESAMPLE_MULTINOMIAL_SORTED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");          // resample.jl:28
    if (h->cfg.n_global != h->n)
        return fail(h, GPF_ERR_STATE, "sharded filters resample through the shard-level API (sharded.py)");
    const bool sorted = method == GPF_RESAMPLE_STRATIFIED && sort_particles;
    const bool need_sync = check == GPF_CHECK_TRUE || invalid != nullptr;
    // only the multinomial search reads the offset levels: the scans of this call write them for it alone
    const bool need_off = method == GPF_RESAMPLE_MULTINOMIAL && (multi_logg(h->ntiles) >= 0 || multi_sample(h->ntiles) > 0);
    struct OffScope { gpf_filter* h; ~OffScope() { h->want_offsets = true; } } off_scope{h};
    h->want_offsets = need_off;
    h->offsets_hint = need_off;
    gpf_status s;
    if ((s = check_scan_timeout(h))) return s;                   // an earlier scan gave up: do not build on its CDF
    if ((s = materialize(h))) return s;                          // two resamples in a row: finish the first one
    // sortperm(log_priorities, rev=true)  (resample.jl:156-157)
    // with priorities the log-ML update needs the summary of the RAW weights (cdf[0] is overwritten later; only S, m matter).  First:
    // it may recompute the maximum slots for the raw weights, and from here on they must describe the priorities (sort keys, scan)
    if (pv.mode != 0 && (s = ensure_raw(h))) return s;
    bool sort_pending = false;                                   // the sort's verdict (k_sort_finish) is asked for after the search is enqueued
    if (sorted) {
        if ((s = ensure_max(h, pv, pv.mode == 0))) return s;     // (the coarse sort keys are distances from the maximum)
        if ((s = sort_desc_begin(h, pv, h->n, &sort_pending))) return s;
    }
    if (method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        // the gamma total of every tile of SP_TILE slots: one lane per tile, as extra workgroups of the weight scan below;
        // beyond SP_DIRECT_TILES tiles one more small launch turns them into the tiles' starting points, else the merge kernel does
        // that for its own tile
        if ((s = sorted_job_prepare(h, h->cfg.gid0, h->n))) return s;
    }
    struct SpScope { gpf_filter* h; ~SpScope() { h->sp_job_set = false; } } sp_scope{h};
    // safe_softmax(log_priorities) (resample.jl:54) and logsumexp(log_weights) (resample.jl:180)
    WSum* ws;
    bool published = false;                                      // the scan of THIS call publishes the flags to pinned memory
    // :residual right after an ESS / log-ML read (README.md:68-70): the summary is with the host (k_sum_host) and the residual scan needs
    // nothing else of the weight scan -- it converts the weights itself (k_scan_residual2<DIRECT>) and the weight scan is not run
    static const bool no_direct = getenv("GPF_RESIDUAL_DIRECT") && !strcmp(getenv("GPF_RESIDUAL_DIRECT"), "0");
    ResidDirect rdirect{};
    const bool resid_direct = !no_direct && method == GPF_RESAMPLE_RESIDUAL && pv.mode == 0 && !h->raw_valid && h->raw_sum_valid && h->sum_on_host;
    int direct_flags = 0;
    if (resid_direct) {
        rdirect = ResidDirect{h->lw, h->sum_cache.m, h->sum_cache.flags, h->K, h->sum_cache.S, &h->sc->raw, nullptr, 0, 0};
        direct_flags = h->sum_cache.flags;
        ws = &h->sc->raw;
    } else
    if (pv.mode == 0) {
        ws = &h->sc->raw;
        if (!h->raw_valid || sorted || (need_off && !h->ch0_offsets)) {
            if ((s = summarize(h, pv, ws, true, sorted ? h->order : nullptr, true, false, need_sync, sorted))) return s;
            published = need_sync;
        }
    } else {
        h->want_offsets = need_off;
        ws = &h->sc->prio;
        if ((s = summarize(h, pv, ws, true, sorted ? h->order : nullptr, false, false, need_sync, sorted))) return s;
        published = need_sync;
    }
    h->raw_valid = false; h->raw_sum_valid = false;                                        // cdf[0] no longer the plain raw CDF / lw about to change
    h->raw_q_folded = false;
    if (need_sync) {
        int flags;
        if (published) {
            // safe_softmax's flags are known when the scan STARTS (it folds the per-block maxima first): poll the ticket; the
            // scan keeps running and the search below is enqueued behind it without a gap
            if ((s = wait_ticket(h, h->h_flags + 1, h->flag_ticket, "weight scan flags"))) return s;
            flags = (int)h->h_flags[0];
        } else if (resid_direct) flags = direct_flags;           // (known since the getter)
        else {
            if ((s = fetch_scalars(h))) return s;
            flags = (pv.mode == 0 ? h->h_sc->raw : h->h_sc->prio).flags;
        }
        const bool inv = flags != 0;
        h->last_flags = flags;
        if (invalid) *invalid = inv ? 1 : 0;
        if (flags & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && inv) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    // ancestors (+ update_lml_est!, resample.jl:57,178-182, inside the search kernel)
    SearchArgs sa{};
    sa.w = levels(h, 0); sa.c = levels(h, 0); sa.ntiles = h->ntiles;
    sa.order = sorted ? h->order : nullptr; sa.sc = h->sc; sa.ws = ws; sa.raw = &h->sc->raw; sa.n = h->n; sa.n_cells = h->n;
    sa.n_global = h->cfg.n_global; sa.gid0 = h->cfg.gid0; sa.seed = h->cfg.seed; sa.epoch = h->epoch;
    sa.K = h->K; sa.logN = h->logN; sa.anc = h->anc; sa.invN = 1.0 / (double)h->cfg.n_global;
    sa.update_lml = local ? 2 : (h->parent ? 0 : 1);             // sub-states do not track the estimate (resample.jl:185-187)

    if (method == GPF_RESAMPLE_RESIDUAL) {
        static const bool head_in_search = getenv("GPF_RESIDUAL_HEAD") && !strcmp(getenv("GPF_RESIDUAL_HEAD"), "search");   // (A/B measurements)
        if ((s = residual_scans(h, ws, h->cfg.n_global, head_in_search ? nullptr : h->anc, resid_direct ? &rdirect : nullptr))) return s;
        sa.w = levels(h, 2); sa.c = levels(h, 1);
        sa.head_done = head_in_search ? 0 : 1;
    }
    if (method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        const int64_t ntl = (h->n + SP_TILE - 1) / SP_TILE;
        // (the tile totals in a launch of their own if no weight scan ran in this call -- the CDF of an earlier getter is reused)
        if ((s = sorted_gammas_finish(h, ntl > SP_DIRECT_TILES))) return s;
        sa.sp_g = h->sp_g; sa.sp_vlo = ntl > SP_DIRECT_TILES ? h->sp_vlo : nullptr;
    }
    const int64_t nt = method == GPF_RESAMPLE_RESIDUAL && !sa.head_done ? 2 : 1;   // (top tables the search keeps in LDS: its shape depends on their number)
    const size_t lds = search_lds_bytes(h->ntiles, (int)nt);
    // every block first copies the top level of the CDF into LDS: keep the grid small (persistent blocks)
    // one 1024-thread workgroup per CU, two slots per lane and iteration
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu * SEARCH_BLOCKS_PER_CU));
    auto search = [&]() {
        return timed(h, GPF_K_SEARCH, [&] {
            switch (method) {
                case GPF_RESAMPLE_MULTINOMIAL: launch_multinomial_search(h, sa); break;
                case GPF_RESAMPLE_RESIDUAL:    GPF_LAUNCH((k_search<1>), dim3(gsr), dim3(SBLOCK), lds, h->stream, sa); break;
                case GPF_RESAMPLE_MULTINOMIAL_SORTED:   // sorted uniforms: the same streaming merge (the spacing sums were enqueued above)
                    GPF_LAUNCH((k_search_strat<true>), dim3((unsigned)((h->n + MJB - 1) / MJB)), dim3(MBLOCK), 0, h->stream, sa); break;
                default:                       // monotone targets: a streaming merge, MJB_STRAT slots per workgroup
                    GPF_LAUNCH((k_search_strat<false>), dim3((unsigned)((h->n + MJB_STRAT - 1) / MJB_STRAT)), dim3(MBLOCK), 0, h->stream, sa); break;
            }
        });
    };
    // lazy search: a plain multinomial resample of a whole filter leaves its search to the pf_update! that follows (k_step_search)
    const bool lazy = method == GPF_RESAMPLE_MULTINOMIAL && pv.mode == 0 && !local && lazy_search_ok(h) && sa.w.off16 && sa.w.sample == 0;
    if (lazy) { h->pend_sa = sa; h->pending_search = true; }
    else if ((s = search())) return s;
    if (sort_pending) {
        // the scan and the search above ran behind the sort without a host wait; if the finish met a run it could not order (equal
        // or nearly equal priorities) they worked on a wrong order: eight passes over the full key, then both again.  The weight
        // sums are order-independent (integers): the log-ML update of the first search stands.
        bool flagged = false;
        if ((s = sort_desc_flagged(h, &flagged))) return s;
        if (flagged) {
            if ((s = sort_passes(h, pv, h->n, false))) return s;
            h->want_offsets = need_off;
            if ((s = summarize(h, pv, ws, true, h->order, pv.mode == 0, false, false, true))) return s;
            sa.w = levels(h, 0); sa.c = levels(h, 0); sa.update_lml = 0;
            if ((s = search())) return s;
        }
    }
    if ((s = hist_on_resample(h))) return s;
    if (h->parent) {
        // sub-state (resample.jl:205-218): eager gather; weights keep the block's total mass
        s = timed(h, GPF_K_GATHER, [&] { launch_gather(h, pv, pv.mode == 0 ? h->lw : h->lws); });
        if (s) return s;
        h->cur ^= 1;
        if (pv.mode == 0) {
            GPF_LAUNCH(k_view_fill_weights, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->lw, h->n, &h->sc->raw, h->K, h->logN);
        } else {
            PrioView post{h->lws, nullptr, 0.0, 0};
            if ((s = summarize(h, post, &h->sc->post, false, nullptr, false))) return s;
            GPF_LAUNCH(k_view_apply_post, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->sc, h->K, h->lws, h->lw, h->n);
        }
        h->max_valid = false;
        HIP_TRY(h, hipGetLastError());
        h->epoch += 1;
        mutated(h);
        return view_exit(h);
    }
    if (pv.mode == 0) {
        // new_traces .= view(traces, parents) is deferred: the next pf_update! reads rows through anc (fused
        // gather), any other consumer calls materialize().  Log-weights are 0 (resample.jl:195).
        h->pending_gather = true;
        h->pending_fill = local;
        h->max_valid = false;
    } else {
        // gather + update_weights! with priorities (resample.jl:60,198-200), update_refs! (utils.jl:10-15)
        s = timed(h, GPF_K_GATHER, [&] { launch_gather(h, pv, h->lws); });
        if (s) return s;
        h->cur ^= 1;
        PrioView post{h->lws, nullptr, 0.0, 0};
        if ((s = summarize(h, post, &h->sc->post, false, nullptr, false))) return s;
        GPF_LAUNCH(k_apply_post, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->sc, h->K, h->logN, h->lws, h->lw, h->n);
        h->max_valid = false;
    }
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    mutated(h);
    return GPF_OK;
}


} // namespace gpfh

extern "C" {

gpf_status gpf_resample(gpf_handle h, int32_t method, double priority_alpha, int32_t sort_particles, int32_t check,
                        int32_t* invalid)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    PrioView pv = raw_view(h);
    if (priority_alpha == priority_alpha) { pv.alpha = priority_alpha; pv.mode = 1; }
    return resample_impl(h, method, pv, sort_particles, check, invalid);
}

// pf_resample!(state[1:n], method) on a whole filter or shard (src/resample.jl:185-187,205-218) without the view's copies: it
// normalises over its OWN n particles (strata, fixed-point scale and log n of n, not of n_global), leaves log_ml_est alone and
// every particle keeps the log-weight logsumexp - log n.  The gather stays deferred like gpf_resample's.
gpf_status gpf_resample_local(gpf_handle h, int32_t method, int32_t sort_particles, int32_t check, int32_t* invalid)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->parent) return fail(h, GPF_ERR_STATE, "gpf_resample_local on a sub-state view: resample the view itself");
    if ((s = materialize(h))) return s;
    struct Scope {                                               // the filter as its own population for the duration of the call
        gpf_filter* h; int K; double logN; int64_t ng;
        explicit Scope(gpf_filter* f) : h(f), K(f->K), logN(f->logN), ng(f->cfg.n_global)
        { h->K = fix_K(h->n); h->logN = log_((double)h->n); h->cfg.n_global = h->n; h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; }
        ~Scope() { h->K = K; h->logN = logN; h->cfg.n_global = ng; h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; }
    } scope(h);
    return resample_impl(h, method, raw_view(h), sort_particles, check, invalid, true);
}

gpf_status gpf_resample_with_priorities(gpf_handle h, int32_t method, const double* log_priorities, int32_t sort_particles,
                                        int32_t check, int32_t* invalid)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!log_priorities) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null log_priorities");
    HIP_TRY(h, hipMemcpyAsync(h->lp, log_priorities, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    PrioView pv{h->lw, h->lp, 0.0, 2};
    return resample_impl(h, method, pv, sort_particles, check, invalid);
}

gpf_status gpf_effective_sample_size(gpf_handle h, double* out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    WSum w{};
    bool reduced = false;
    if ((s = ensure_raw_summary(h, true, &reduced))) return s;
    if (reduced) w = h->sum_cache;                                // S and sum q^2 from the reduction: no CDF was written
    else {
        h->q_published = false;
        if ((s = ensure_raw(h, true))) return s;
        if (h->q_published) {
            // the scan of this call publishes {flags, S, limbs of sum q^2} itself: wait for its ticket, no publish launch
            h->q_published = false;
            if ((s = read_published_summary(h, w))) return s;
        } else {
            const bool fold = !h->raw_q_folded;                      // the scan blocks' limb partials of sum q^2: folded by the publish kernel
            if ((s = fetch_scalars(h, fold))) return s;
            h->raw_q_folded = true;
            w = h->h_sc->raw;
        }
    }
    if (w.flags) { *out = std::nan(""); return GPF_OK; }
    uint64_t hi, lo;
    normalise_Q(w, hi, lo);
    *out = ess_from(w.S, hi, lo);
    return GPF_OK;
}

gpf_status gpf_log_ml_estimate(gpf_handle h, double* out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    bool reduced = false;
    if ((s = ensure_raw_summary(h, false, &reduced))) return s;  // (S and the maximum are all it needs: no CDF)
    if (!reduced && (s = ensure_raw(h))) return s;
    if ((s = fetch_scalars(h))) return s;
    const WSum& w = reduced && h->sum_on_host ? h->sum_cache : h->h_sc->raw;   // (k_sum_host leaves {m, flags, S} with the host, not in the device block)
    double base = h->h_sc->lml_est;
    if (h->parent) {                                             // source.log_ml_est (utils.jl:174-178)
        if ((s = fetch_scalars(h->parent))) { h->err = h->parent->err; return s; }
        base = h->parent->h_sc->lml_est;
    }
    *out = base + lse_from(w.m, w.S, h->K, w.flags) - h->logN;
    return GPF_OK;
}

static gpf_status norm_weights(gpf_handle h, double* out, int64_t n, int want_log)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if ((s = ensure_raw(h))) return s;
    GPF_LAUNCH(k_norm_weights, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->lw, &h->sc->raw, h->K, h->n, want_log,
                       h->dtmp);
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(double));
}
gpf_status gpf_get_log_norm_weights(gpf_handle h, double* out, int64_t n) { return norm_weights(h, out, n, 1); }
gpf_status gpf_get_norm_weights(gpf_handle h, double* out, int64_t n) { return norm_weights(h, out, n, 0); }

// Gen.sample_unweighted_traces(state, n_samples) (reference src/utils.jl:7,189-194): n i.i.d. draws from the normalised
// weights, WITHOUT touching the filter (no log-ML update, weights unchanged).  Same CDF + search kernels as a resample.
gpf_status gpf_sample_unweighted(gpf_handle h, int64_t n_samples, double* rows_out, int64_t* idx_out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (n_samples < 1 || n_samples >= ((int64_t)1 << 31) || !rows_out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "not available on shards");
    if ((s = ensure_raw(h))) return s;                            // CDF of state.log_weights in cdf[0]
    if ((s = fetch_scalars(h))) return s;
    if (h->h_sc->raw.flags & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
    int32_t* anc = nullptr; double* rows = nullptr; int64_t* idx64 = nullptr;
    HIP_TRY(h, hipMalloc(&anc, (size_t)n_samples * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&rows, (size_t)n_samples * h->W * sizeof(double)));
    SearchArgs sa{};
    sa.w = levels(h, 0); sa.c = levels(h, 0); sa.ntiles = h->ntiles; sa.order = nullptr; sa.sc = h->sc; sa.ws = &h->sc->raw;
    sa.raw = &h->sc->raw; sa.n = n_samples; sa.n_cells = h->n; sa.n_global = n_samples; sa.gid0 = h->cfg.gid0; sa.seed = h->cfg.seed;
    sa.epoch = h->epoch; sa.K = h->K; sa.logN = h->logN; sa.update_lml = 0; sa.anc = anc;
    launch_multinomial_search(h, sa);
    launch_gather_rows_lw(h, anc, h->rows[h->cur], h->lw, rows, nullptr, n_samples);
    HIP_TRY(h, hipMemcpyAsync(rows_out, rows, (size_t)n_samples * h->W * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (idx_out) {
        HIP_TRY(h, hipMalloc(&idx64, (size_t)n_samples * sizeof(int64_t)));
        GPF_LAUNCH(k_parents, dim3(grid_for(h, n_samples, 8)), dim3(BLOCK), 0, h->stream, anc, n_samples, idx64);
        HIP_TRY(h, hipMemcpyAsync(idx_out, idx64, (size_t)n_samples * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    (void)hipFree(anc); (void)hipFree(rows); if (idx64) (void)hipFree(idx64);
    h->epoch += 1;
    if (h->parent) h->parent->epoch = h->epoch;
    return GPF_OK;
}


} // extern "C"
